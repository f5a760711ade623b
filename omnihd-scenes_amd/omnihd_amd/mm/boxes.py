"""Box structure and test-time post-process helpers with the mmdet3d v0.17.1 names the reference
calls (un-vendored upstream; call sites in the reference:
projects/mmdet3d_plugin/bevfusion/dense_heads/det_anchor3d_head.py:492-516 [xywhr2xyxyr,
box3d_multiclass_nms, box_type_3d(...).bev], datasets/newscenes_dataset.py:273-277,552-560
[LiDARInstance3DBoxes(origin=(0.5,0.5,0.5)), gravity_center, dims, yaw], detectors'
simple_test -> bbox3d2result).

Boxes are (x, y, z_bottom, x_size(w), y_size(l), z_size(h), yaw[, vx, vy]) in the LiDAR frame — the
v0.17.1 convention.  The rotated NMS itself is the HIP kernel behind ``ops.nms_rotated``; there is no
CPU path (tests run the host logic over oracle/torch_shim.py).
"""
import torch

from .. import ops


class LiDARInstance3DBoxes:
    """Minimal restatement of mmdet3d's box container: only what the reference path touches."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:                      # no yaw given: upstream appends zeros
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            box_dim, with_yaw = 7, False
        self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            src = self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    bottom_center = property(lambda self: self.tensor[:, :3])
    dims = property(lambda self: self.tensor[:, 3:6])
    yaw = property(lambda self: self.tensor[:, 6])
    bev = property(lambda self: self.tensor[:, [0, 1, 3, 4, 6]])
    device = property(lambda self: self.tensor.device)

    @property
    def gravity_center(self):
        gc = torch.zeros_like(self.bottom_center)
        gc[:, :2] = self.bottom_center[:, :2]
        gc[:, 2] = self.bottom_center[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        if t.dim() == 1:
            t = t.view(1, -1)
        return LiDARInstance3DBoxes(t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def to(self, device):
        return LiDARInstance3DBoxes(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def convert_to(self, dst, rt_mat=None):            # LiDAR -> LiDAR is the only mode on this path
        return self

    def __repr__(self):
        return f"LiDARInstance3DBoxes(\n    {self.tensor})"


def xywhr2xyxyr(boxes_xywhr):
    """(x, y, w, h, r) centre form -> (x1, y1, x2, y2, r) corner form of the unrotated rectangle."""
    out = torch.zeros_like(boxes_xywhr)
    half_w, half_h = boxes_xywhr[:, 2] / 2, boxes_xywhr[:, 3] / 2
    out[:, 0] = boxes_xywhr[:, 0] - half_w
    out[:, 1] = boxes_xywhr[:, 1] - half_h
    out[:, 2] = boxes_xywhr[:, 0] + half_w
    out[:, 3] = boxes_xywhr[:, 1] + half_h
    out[:, 4] = boxes_xywhr[:, 4]
    return out


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """Upstream name of the rotated BEV NMS op (mmdet3d.ops.iou3d.nms_gpu)."""
    return ops.nms_rotated(boxes.contiguous().float(), scores, thresh, pre_maxsize, post_max_size)


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg,
                         mlvl_dir_scores=None):
    """Per-class score threshold + rotated NMS, then a global top-``max_num`` by score.  The last
    column of ``mlvl_scores`` is the background slot.  Returns (bboxes, scores, labels[, dir_scores])."""
    if not cfg.get("use_rotate_nms", False):
        raise NotImplementedError("only use_rotate_nms=True (the reference's test_cfg) is built")
    num_classes = mlvl_scores.shape[1] - 1
    picked = []                                        # (box rows, class scores, class id) per class
    for c in range(num_classes):
        above = (mlvl_scores[:, c] > score_thr).nonzero(as_tuple=False).squeeze(1)
        if above.numel() == 0:
            continue
        cls_scores = mlvl_scores[above, c]
        keep = nms_gpu(mlvl_bboxes_for_nms[above], cls_scores, cfg["nms_thr"])
        picked.append((above[keep], cls_scores[keep], c))
    if not picked:
        empty = [mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1))), mlvl_scores.new_zeros((0,)),
                 mlvl_scores.new_zeros((0,), dtype=torch.long)]
        if mlvl_dir_scores is not None:
            empty.append(mlvl_scores.new_zeros((0,)))
        return tuple(empty)
    rows = torch.cat([p[0] for p in picked])
    scores = torch.cat([p[1] for p in picked])
    labels = torch.cat([rows.new_full((p[0].numel(),), p[2], dtype=torch.long) for p in picked])
    if rows.numel() > max_num:
        top = scores.sort(descending=True)[1][:max_num]
        rows, scores, labels = rows[top], scores[top], labels[top]
    out = [mlvl_bboxes[rows], scores, labels]
    if mlvl_dir_scores is not None:
        out.append(mlvl_dir_scores[rows])
    return tuple(out)


def bbox3d2result(bboxes, scores, labels):
    """Detection triple -> the CPU result dict the dataset formatter consumes."""
    return dict(boxes_3d=bboxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu())
