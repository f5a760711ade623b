"""Stage-1 -> fusion checkpoint stitching: the ``load_*`` keys of the reference configs, applied the way
tools/train.py:270-425 applies them before training starts, followed by the runner's own ``load_from``
(mmcv ``load_checkpoint(strict=False)``).  The fusion recipe depends on it: ``bevfusion.py:288-290`` takes
the image branch + lift module from the camera-only run (``load_lift_from``) and the radar stream + head
from the radar-only run (``load_from``); it works only if the three detectors use the same state-dict
names, which is what tests/test_checkpoint_cpu.py checks end to end.

Rules (first to last, later ones overwrite earlier ones):
  load_img_from                              ``backbone.*`` -> ``img_backbone.*``, ``neck.*`` -> ``img_neck.*``; rest dropped
  load_img_from_and_not_change_state_dict   everything except ``bbox_head*``
  load_lift_from                             everything except ``pts_bbox_head*``
  load_pts_from                              ``backbone/neck/voxel_encoder.*`` -> ``pts_*``; ``bbox_head*`` and the rest dropped
  load_from                                  everything (a leading ``module.`` stripped)
All non-strict: tensors whose name or shape does not match the model are skipped and reported.
"""
from collections import OrderedDict

import torch

__all__ = ["state_dict_of", "stitch_checkpoints", "RULES"]


def state_dict_of(checkpoint):
    for k in ("state_dict", "model"):
        if isinstance(checkpoint, dict) and k in checkpoint and isinstance(checkpoint[k], dict):
            return checkpoint[k]
    return checkpoint


def _strip_module(sd):
    if sd and next(iter(sd)).startswith("module."):
        return OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def _rename(prefix_map, drop=()):
    def rule(sd):
        out = OrderedDict()
        for k, v in sd.items():
            if k.startswith(tuple(drop)):
                continue
            for src, dst in prefix_map:
                if k.startswith(src):
                    out[dst + k[len(src):]] = v
                    break
        return out
    return rule


def _all_but(*prefixes):
    return lambda sd: OrderedDict((k, v) for k, v in sd.items() if not k.startswith(prefixes))


RULES = OrderedDict([
    ("load_img_from", lambda sd: _rename([("backbone.", "img_backbone."), ("neck.", "img_neck.")])(_strip_module(sd))),
    ("load_img_from_and_not_change_state_dict", _all_but("bbox_head")),
    ("load_lift_from", _all_but("pts_bbox_head")),
    ("load_pts_from", _rename([("backbone.", "pts_backbone."), ("neck.", "pts_neck."), ("voxel_encoder.", "pts_voxel_encoder.")],
                              drop=("bbox_head",))),
    ("load_from", _strip_module),
])


def stitch_checkpoints(model, cfg, loader=None):
    """Apply every ``load_*`` key present (and not None) in ``cfg`` (the dict ``load_config`` returns).  ``loader(path)``
    defaults to ``torch.load(path, map_location='cpu')``.  -> {rule: dict(loaded=[...], skipped=[...])}."""
    loader = loader or (lambda path: torch.load(path, map_location="cpu"))
    own = model.state_dict()
    report = OrderedDict()
    for key, rule in RULES.items():
        path = cfg.get(key)
        if not path:
            continue
        sd = rule(state_dict_of(loader(path)))
        take = OrderedDict((k, v) for k, v in sd.items() if k in own and own[k].shape == v.shape)
        model.load_state_dict(take, strict=False)
        report[key] = dict(loaded=list(take), skipped=[k for k in sd if k not in take])
    return report
