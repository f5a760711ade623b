"""GPU parity of the test-time post-process: rotated BEV IoU / NMS kernels
against the sequential oracle (oracle/nms_oracle.c; parity unpinned upstream, see its header),
bit-exact for every IoU and every kept index; then ``Anchor3DHead.get_bboxes`` and the detector's
``simple_test`` end to end."""
import numpy as np
import pytest
import torch

from oracle import cpu as OC
from tests.helpers import t

pytestmark = pytest.mark.gpu


def bev_boxes(rng, n, extent=(60.0, 40.0), clusters=0):
    """(n,5) boxes (x1,y1,x2,y2,ry) of vehicle/pedestrian sizes; `clusters` > 0 piles them up."""
    if clusters:
        cen = rng.uniform(-1, 1, (clusters, 2)) * np.array(extent)
        xy = cen[rng.integers(0, clusters, n)] + rng.normal(0, 1.2, (n, 2))
    else:
        xy = rng.uniform(-1, 1, (n, 2)) * np.array(extent)
    wl = np.stack([rng.uniform(0.5, 2.6, n), rng.uniform(0.5, 11.0, n)], 1)
    r = rng.uniform(-np.pi, np.pi, n)
    return np.concatenate([xy - wl / 2, xy + wl / 2, r[:, None]], 1).astype(np.float32)


@pytest.mark.parametrize("na,nb,clusters", [(1, 1, 0), (37, 91, 3), (300, 300, 12), (128, 64, 1)])
def test_iou_matrix_bit_exact(cuda, na, nb, clusters):
    from omnihd_amd import ops
    rng = np.random.default_rng(na * 1000 + nb)
    a, b = bev_boxes(rng, na, clusters=clusters), bev_boxes(rng, nb, clusters=clusters)
    if clusters:
        b[: min(na, nb) // 2] = a[: min(na, nb) // 2]                  # identical boxes: IoU exactly 1 path
    got = ops.iou_bev_matrix(t(a, cuda), t(b, cuda)).cpu().numpy()
    want = OC.iou_bev_matrix(a, b)
    assert (want > 0).sum() > 0 or clusters == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())


def test_iou_degenerate_and_axis_aligned_cases(cuda):
    from omnihd_amd import ops

    def box(cx, cy, w, h, r):
        return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, r]
    a = np.array([box(0, 0, 2, 2, 0)], np.float32)
    b = np.array([box(1, 0, 2, 2, 0), box(0, 0, 2, 2, np.pi / 4), box(5, 5, 1, 1, 0.3), box(0, 0, 1, 1, 1.0),
                  box(0, 0, 2, 2, 0), box(2, 0, 2, 2, 0), box(0, 0, 0, 0, 0), box(0, 0, 2, 2, np.pi / 2)], np.float32)
    got = ops.iou_bev_matrix(t(a, cuda), t(b, cuda)).cpu().numpy()
    want = OC.iou_bev_matrix(a, b)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(got[0, :5], [1 / 3, 8 * (2 ** 0.5 - 1) / (8 - 8 * (2 ** 0.5 - 1)), 0, 0.25, 1], atol=1e-6)


@pytest.mark.parametrize("n,clusters,thr", [(0, 0, 0.2), (1, 0, 0.2), (63, 2, 0.2), (64, 2, 0.2), (65, 2, 0.01),
                                            (1000, 40, 0.2), (1000, 6, 0.2), (2500, 30, 0.5), (4096, 100, 0.2)])
def test_nms_keep_identical(cuda, n, clusters, thr):
    from omnihd_amd import ops
    rng = np.random.default_rng(n + clusters)
    boxes = bev_boxes(rng, n, clusters=clusters) if n else np.zeros((0, 5), np.float32)
    scores = rng.permutation(n).astype(np.float32) / max(n, 1)          # distinct -> one sort order
    got = ops.nms_rotated(t(boxes, cuda), t(scores, cuda), thr).cpu().numpy()
    want = OC.nms_rotated(boxes, scores, thr)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    if n >= 1000:
        assert 0 < len(want) < n
    # pre/post size clamps of the upstream signature
    if n >= 64:
        got = ops.nms_rotated(t(boxes, cuda), t(scores, cuda), thr, pre_maxsize=50, post_max_size=7).cpu().numpy()
        assert np.array_equal(got, OC.nms_rotated(boxes, scores, thr, 50, 7))


def test_nms_rejects_cpu_tensors_and_oversize(cuda):
    from omnihd_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.nms_rotated(torch.zeros(4, 5), torch.zeros(4), 0.2)
    with pytest.raises(RuntimeError, match="4096"):
        ops.nms_rotated(torch.zeros(5000, 5, device=cuda), torch.zeros(5000, device=cuda), 0.2)


def _head_and_outputs(device, seed=0, hw=(20, 30)):
    from omnihd_amd.harness import tiny_model_cfg
    from omnihd_amd.mm import anchor_head  # noqa: F401  (registers Anchor3DHead)
    from omnihd_amd.mm.registry import HEADS
    cfg = tiny_model_cfg()
    head_cfg = dict(cfg["pts_bbox_head"])
    head_cfg.update(train_cfg=None, test_cfg=dict(cfg["test_cfg"]["pts"]))
    torch.manual_seed(seed)
    head = HEADS.build(head_cfg).to(device)
    g = torch.Generator().manual_seed(seed + 1)
    na = head.num_anchors
    n_cls = 2 * na * head.num_classes * hw[0] * hw[1]                  # distinct, well separated logits: the
    cls = (torch.randperm(n_cls, generator=g).float() / n_cls * 8 - 5)  # GPU and CPU sigmoid sort identically
    cls = cls.view(2, na * head.num_classes, *hw)
    reg = torch.randn(2, na * head.box_code_size, *hw, generator=g) * 0.3
    dirs = torch.randn(2, na * 2, *hw, generator=g)
    return head, [cls.to(device)], [reg.to(device)], [dirs.to(device)]


def test_get_bboxes_matches_host_logic_over_the_oracle(cuda):
    """Same logits/deltas: GPU path (HIP NMS) vs CPU torch + sequential oracle NMS."""
    from oracle.torch_shim import oracle_ops
    head, cls, reg, dirs = _head_and_outputs(cuda)
    metas = [dict(), dict()]
    got = head.get_bboxes(cls, reg, dirs, metas)
    with oracle_ops():
        chead, ccls, creg, cdirs = _head_and_outputs("cpu")
        want = chead.get_bboxes(ccls, creg, cdirs, metas)
    for (gb, gs, gl), (wb, ws, wl) in zip(got, want):
        assert 0 < len(wb) <= 500
        assert len(gb) == len(wb)
        assert torch.equal(gl.cpu(), wl)
        torch.testing.assert_close(gs.cpu(), ws, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(gb.tensor.cpu(), wb.tensor, rtol=1e-4, atol=1e-4)
        assert gb.tensor.shape[1] == 9 and float(gs.min()) > 0.05


def test_simple_test_returns_reference_result_dicts(cuda):
    from omnihd_amd.harness import FusionTrainStep
    st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cuda:0", seed=5, dtype="fp32", channels_last=False, sets=1)
    m, b = st.raw_model, st.batches[0]
    m.eval()
    torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, 0.0)          # random-init head: let boxes through score_thr
    out = m(return_loss=False, points=[b["points"]], img_metas=[b["img_metas"]], img=[b["img"]])
    assert len(out) == 2
    for r in out:
        d = r["pts_bbox"]
        assert set(d) == {"boxes_3d", "scores_3d", "labels_3d"}
        assert d["boxes_3d"].tensor.device.type == "cpu" and d["boxes_3d"].tensor.shape[1] == 9
        assert len(d["boxes_3d"]) == len(d["scores_3d"]) == len(d["labels_3d"]) <= 500
        assert len(d["scores_3d"]) > 0 and d["labels_3d"].dtype == torch.long
