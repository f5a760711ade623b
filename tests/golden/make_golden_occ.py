#!/usr/bin/env python3
"""Golden vectors for the occupancy head's losses: the reference's own
``BEVOCCHead2Dv2.sem_scal_loss`` / ``geo_scal_loss`` / ``loss`` / ``get_occ``
(projects/mmdet3d_plugin/bevfusion/dense_heads/bev_occ_head.py:770-895) executed on seeded logits and labels.
mmcv / mmdet3d are absent: the file is loaded behind inert stand-ins (decorator registry, ``BaseModule`` =
nn.Module, ``ConvModule`` unused here); ``loss_occ`` is set to the arithmetic mmdet 2.14's
``CrossEntropyLoss(use_sigmoid=False, loss_weight=1.0)`` performs: mean of ``F.cross_entropy(reduction='none')``.
Runs only in the authoring container.  Usage: python tests/golden/make_golden_occ.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


class _Reg:
    def register_module(self, *a, **k):
        return (lambda c: c) if not (len(a) == 1 and callable(a[0])) else a[0]


def main():
    _mod("mmcv"); _mod("mmcv.cnn", ConvModule=None); _mod("mmcv.runner", BaseModule=nn.Module)
    _mod("mmdet3d"); _mod("mmdet3d.models"); _mod("mmdet3d.models.builder", HEADS=_Reg(), build_loss=lambda cfg: None)
    spec = importlib.util.spec_from_file_location("ref_occ", os.path.join(REF, "projects/mmdet3d_plugin/bevfusion/dense_heads/bev_occ_head.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    Head = ref.BEVOCCHead2Dv2
    out = {}
    for case, (seed, shape, n_cls, unknown, drop) in enumerate([(0, (2, 12, 10, 4), 12, 0.1, ()), (1, (1, 24, 16, 16), 12, 0.0, (3, 7)),
                                                                (2, (1, 6, 5, 3), 5, 0.3, (2,))]):
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(*shape, n_cls, generator=g) * 2.0
        labels = torch.randint(0, n_cls, shape, generator=g)
        for d in drop:                                   # classes absent from the target
            labels[labels == d] = (d + 1) % n_cls
        unk = torch.rand(shape, generator=g) < unknown
        fake = types.SimpleNamespace(num_classes=n_cls)
        fake.sem_scal_loss = lambda p, t: Head.sem_scal_loss(fake, p, t)
        fake.geo_scal_loss = lambda p, t, semantic=True: Head.geo_scal_loss(fake, p, t, semantic)
        fake.loss_occ = lambda pred, lab: F.cross_entropy(pred, lab, reduction="none").mean()
        lab_unk = labels.clone()
        lab_unk[unk] = 255
        out[f"c{case}_logits"], out[f"c{case}_labels"], out[f"c{case}_labels_unknown"] = logits.numpy(), labels.numpy(), lab_unk.numpy()
        out[f"c{case}_sem"] = Head.sem_scal_loss(fake, logits, lab_unk).numpy()
        out[f"c{case}_geo"] = Head.geo_scal_loss(fake, logits, lab_unk).numpy()
        full = Head.loss(fake, logits, labels)            # labels without 255: the reference's CE would raise on 255
        out[f"c{case}_loss_ssc"], out[f"c{case}_loss_occ"] = full["loss_ssc"].numpy(), full["loss_occ"].numpy()
        out[f"c{case}_occ"] = np.stack(Head.get_occ(fake, logits))
        print(case, float(out[f"c{case}_sem"]), float(out[f"c{case}_geo"]), float(out[f"c{case}_loss_occ"]))
    path = os.path.join(HERE, "occ_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()


def occupancy_scores():
    """Second part: the reference's occupancy counters (datasets/evaluation_metrics.py:98-120), which import only
    numpy and torch, on seeded class maps -> appended to occ_golden.npz."""
    spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "projects/mmdet3d_plugin/datasets/evaluation_metrics.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    path = os.path.join(HERE, "occ_golden.npz")
    out = dict(np.load(path))
    g = torch.Generator().manual_seed(5)
    pred = torch.randint(0, 12, (3, 20, 12, 8), generator=g)
    gt = torch.randint(0, 12, (3, 20, 12, 8), generator=g)
    gt[torch.rand(gt.shape, generator=g) < 0.6] = 0
    pred[torch.rand(pred.shape, generator=g) < 0.5] = 0
    out["score_pred"], out["score_gt"] = pred.numpy(), gt.numpy()
    out["score_tables"] = ref.aug_evaluation_semantic(pred, gt, None, 12)
    np.savez_compressed(path, **out)
    print("scores", out["score_tables"].shape)


if __name__ == "__main__":
    occupancy_scores()
