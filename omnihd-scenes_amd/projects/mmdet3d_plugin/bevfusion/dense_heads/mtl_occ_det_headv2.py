"""Multi-task head (3-D detection and/or occupancy on the same fused BEV feature) — mirror of the reference's
``MultiTaskHeadv2`` (projects/mmdet3d_plugin/bevfusion/dense_heads/mtl_occ_det_headv2.py:21-183) and of
``BevFeatureSlicer`` (dense_heads/map_head.py:37-76)."""
import torch
import torch.nn.functional as F
from torch import nn

from omnihd_amd.mm import anchor_head  # noqa: F401  (registers the detection head)
from omnihd_amd.mm.registry import HEADS


def _bev_params(xbound, ybound, zbound):
    res = torch.tensor([row[2] for row in (xbound, ybound, zbound)])
    start = torch.tensor([row[0] + row[2] / 2.0 for row in (xbound, ybound, zbound)])
    return res, start


class BevFeatureSlicer(nn.Module):
    """Crop / resample the BEV feature to a task's grid; the identity when the grids coincide (the shipped config)."""

    def __init__(self, grid_conf, map_grid_conf):
        super().__init__()
        self.identity_mapping = grid_conf == map_grid_conf
        if not self.identity_mapping:
            _, start = _bev_params(grid_conf["xbound"], grid_conf["ybound"], grid_conf["zbound"])
            mres, mstart = _bev_params(map_grid_conf["xbound"], map_grid_conf["ybound"], map_grid_conf["zbound"])
            map_x = torch.arange(float(mstart[0]), map_grid_conf["xbound"][1], float(mres[0]))
            map_y = torch.arange(float(mstart[1]), map_grid_conf["ybound"][1], float(mres[1]))
            grid = torch.stack(torch.meshgrid(map_x / (-start[0]), map_y / (-start[1]), indexing="xy"), dim=2)
            self.register_buffer("map_grid", grid, persistent=False)

    def forward(self, x):
        if self.identity_mapping:
            return x
        grid = self.map_grid.unsqueeze(0).type_as(x).repeat(x.shape[0], 1, 1, 1)
        return F.grid_sample(x, grid=grid, mode="bilinear", align_corners=True)


@HEADS.register_module()
class MultiTaskHeadv2(nn.Module):
    def __init__(self, init_cfg=None, in_channels=64, out_channels=256, bev_encode_block="Basic",
                 bev_encoder_type="resnet18", bev_encode_depth=(1, 1, 1), num_channels=None, backbone_output_ids=None,
                 norm_cfg=dict(type="BN"), bev_encoder_fpn_type="lssfpn", grid_conf=None, det_grid_conf=None,
                 occ_grid_conf=None, task_enbale=None, task_weights=None, out_with_activision=False,
                 shared_feature=False, cfg_3dod=None, cfg_occ=None, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert bev_encoder_type == "resnet18" and not shared_feature
        self.task_enbale, self.task_weights = task_enbale, task_weights or {}
        det_grid_conf = grid_conf if det_grid_conf is None else det_grid_conf
        self.task_decoders = nn.ModuleDict()
        self.task_feat_cropper = nn.ModuleDict()
        if task_enbale.get("3dod", False):
            cfg = dict(cfg_3dod)
            cfg.update(train_cfg=train_cfg, test_cfg=test_cfg)
            self.task_feat_cropper["3dod"] = BevFeatureSlicer(grid_conf, det_grid_conf)
            self.task_decoders["3dod"] = HEADS.build(cfg)
        if task_enbale.get("occ", False):
            self.task_feat_cropper["occ"] = BevFeatureSlicer(grid_conf, occ_grid_conf)
            self.task_decoders["occ"] = HEADS.build(cfg_occ)

    def scale_task_losses(self, task_name, task_loss_dict):
        """Task weight applied per loss; ``<task>_sum`` added (the reference also reads every value to the host
        with ``.item()`` for a variable it never uses — dropped: it is a device synchronisation per loss)."""
        w = self.task_weights.get(task_name, 1.0)
        out = {k: (v[0] if isinstance(v, (list, tuple)) else v) * w for k, v in task_loss_dict.items()}
        out["{}_sum".format(task_name)] = sum(out.values())
        return out

    def loss(self, predictions, img_metas, targets):
        losses = {}
        if self.task_enbale.get("3dod", False):
            det = self.task_decoders["3dod"].loss(*predictions["3dod"], targets["gt_bboxes_3d"], targets["gt_labels_3d"],
                                                  img_metas, gt_bboxes_ignore=targets["gt_bboxes_ignore"])
            losses.update(self.scale_task_losses("3dod", det))
        if self.task_enbale.get("occ", False):
            occ = self.task_decoders["occ"].loss(predictions["occ"], targets["gt_occ"])
            losses.update(self.scale_task_losses("occ", occ))
        return losses

    def inference(self, predictions, img_metas, rescale):
        res = {}
        if self.task_enbale.get("3dod", False):
            res["bbox_list"] = self.task_decoders["3dod"].get_bboxes(*predictions["3dod"], img_metas, rescale=rescale)
        if self.task_enbale.get("occ", False):
            res["occ_pred"] = predictions["occ"]
        return res

    def forward(self, bev_feats, targets=None):
        """``bev_feats``: [ (B, C, Dy, Dx) ] -> {'3dod': (cls, reg, dir) lists, 'occ': (B, Dx, Dy, Dz, n_cls)}."""
        out = {}
        for name, crop in self.task_feat_cropper.items():
            out[name] = self.task_decoders[name]([crop(f) for f in bev_feats])
        return out
