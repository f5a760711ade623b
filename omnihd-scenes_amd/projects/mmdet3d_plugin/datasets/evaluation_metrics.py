"""Occupancy scoring — mirror of the reference's semantic occupancy counters
(projects/mmdet3d_plugin/datasets/evaluation_metrics.py: ``aug_evaluation_semantic`` :98-120,
``evaluation_semantic`` :53-75) and of the mIoU reduction in ``NewScenesDataset_MTL.evaluate``
(datasets/newscenes_dataset_MTL.py:548-572).

Per sample a (class_num, 3) table: row 0 = geometry (occupied vs free: both non-zero, gt non-zero, pred non-zero),
row j>0 = (true positives, gt voxels, predicted voxels) of class j.  Computed with two bincounts on the device
(no per-class host loop); returned as float64 numpy like the reference."""
import numpy as np
import torch


def _count_tables(pred, gt, class_num, known=None):
    B = pred.shape[0]
    pred, gt = pred.reshape(B, -1).long(), gt.reshape(B, -1).long()
    if known is None:
        known = torch.ones_like(gt, dtype=torch.bool)
    else:
        known = known.reshape(B, -1)
    out = torch.zeros(B, class_num, 3, dtype=torch.float64, device=pred.device)
    for b in range(B):
        p, g = pred[b][known[b]].clamp(0, class_num - 1), gt[b][known[b]]
        g_in = (g >= 0) & (g < class_num)
        gc = torch.bincount(g[g_in], minlength=class_num)[:class_num]
        pc = torch.bincount(p, minlength=class_num)[:class_num]
        tp = torch.bincount(g[g_in & (p == g)], minlength=class_num)[:class_num]
        out[b, :, 0], out[b, :, 1], out[b, :, 2] = tp, gc, pc
        out[b, 0, 0] = ((g != 0) & (p != 0)).sum()
        out[b, 0, 1] = (g != 0).sum()
        out[b, 0, 2] = (p != 0).sum()
    return out.cpu().numpy()


def aug_evaluation_semantic(pred_occ, gt_occ, img_metas, class_num):
    """pred_occ, gt_occ: (B, Dx, Dy, Dz) class maps -> (B, class_num, 3)."""
    return _count_tables(torch.as_tensor(pred_occ), torch.as_tensor(gt_occ).to(torch.as_tensor(pred_occ).device), class_num)


def evaluation_semantic(pred_occ, gt_occ, img_metas, class_num):
    """gt_occ: (B, N, 4) sparse voxels (x, y, z, class) scattered into ``img_metas['occ_size']``; 255 = unknown."""
    pred = torch.as_tensor(pred_occ)
    dense = torch.zeros((pred.shape[0], *img_metas["occ_size"]), dtype=torch.long, device=pred.device)
    for b in range(pred.shape[0]):
        g = torch.as_tensor(gt_occ[b]).to(pred.device)
        dense[b][g[:, 0].long(), g[:, 1].long(), g[:, 2].long()] = g[:, 3].long()
    return _count_tables(pred, dense, class_num, known=dense != 255)


def occupancy_miou(score_tables, occ_class_names):
    """List of per-sample tables -> {'IoU': geometric IoU, <class>: IoU..., 'mIoU': mean over the semantic classes}.
    (The reference labels its columns tp / p / g with p = column 1; the union is symmetric in the two.)"""
    res = np.stack([np.asarray(t).reshape(-1, 3) if np.asarray(t).ndim == 2 else np.asarray(t)[0] for t in score_tables], 0).mean(0)
    ious = res[:, 0] / (res[:, 1] + res[:, 2] - res[:, 0])
    out = {"IoU": ious[0]}
    for i, name in enumerate(occ_class_names):
        out[name] = ious[i + 1]
    out["mIoU"] = float(np.mean(ious[1:]))
    return out
