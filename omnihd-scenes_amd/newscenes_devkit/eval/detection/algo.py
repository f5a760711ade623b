"""Matching and metric integration of the NewScenes detection benchmark (reference:
newscenes_devkit/eval/detection/algo.py: accumulate :17-173, calc_ap :176-185, calc_tp :188-202).

Same results as the reference's per-box Python loops, computed on arrays: the boxes of one class
are flattened into struct-of-arrays once (``ClassTable``), the prediction x ground-truth distances
of a sample are one matrix, and only the inherently sequential part — greedy assignment in
descending confidence — is a loop, over row views of those matrices.
"""
from typing import Callable

import numpy as np

from newscenes_devkit.eval.common.data_classes import EvalBoxes
from newscenes_devkit.eval.common.utils import angle_diff, center_distance, cummean, quaternion_yaw_wxyz
from newscenes_devkit.eval.detection.data_classes import DetectionMetricData


class ClassTable:
    """All boxes of one class from an EvalBoxes, as arrays, plus per-sample slices."""

    def __init__(self, boxes: EvalBoxes, class_name: str, tokens=None):
        tokens = boxes.sample_tokens if tokens is None else tokens
        rows, self.slices, self.local_index = [], {}, {}
        for tok in tokens:
            start = len(rows)
            idx = [i for i, b in enumerate(boxes[tok]) if b.detection_name == class_name] if tok in boxes.boxes else []
            rows.extend(boxes[tok][i] for i in idx)
            self.slices[tok] = slice(start, len(rows))
            self.local_index[tok] = idx                     # position of each row inside its sample's list
        self.boxes = rows
        n = len(rows)
        self.xy = np.array([b.translation[:2] for b in rows], dtype=float).reshape(n, 2)
        self.vel = np.array([b.velocity for b in rows], dtype=float).reshape(n, 2)
        self.size = np.array([b.size for b in rows], dtype=float).reshape(n, 3)
        self.yaw = np.array([quaternion_yaw_wxyz(b.rotation) for b in rows], dtype=float)
        self.score = np.array([b.detection_score for b in rows], dtype=float)
        self.token = [b.sample_token for b in rows]


def _confidence_order(scores):
    """The reference sorts (score, index) pairs ascending and reverses: ties go to the LATER box."""
    return np.lexsort((np.arange(len(scores)), scores))[::-1]


def accumulate(gt_boxes: EvalBoxes, pred_boxes: EvalBoxes, class_name: str, dist_fcn: Callable, dist_th: float,
               verbose: bool = True) -> DetectionMetricData:
    """Precision/recall/confidence and the four TP error curves of one class at one match distance."""
    npos = sum(1 for b in gt_boxes.all if b.detection_name == class_name)
    if verbose:
        print("Found {} GT of class {} out of {} total across {} samples.".format(
            npos, class_name, len(gt_boxes.all), len(gt_boxes.sample_tokens)))
    if npos == 0:
        return DetectionMetricData.no_predictions()

    pred = ClassTable(pred_boxes, class_name)
    gt = ClassTable(gt_boxes, class_name, tokens=list(dict.fromkeys(pred_boxes.sample_tokens + gt_boxes.sample_tokens)))
    if verbose:
        print("Found {} PRED of class {} out of {} total across {} samples.".format(
            len(pred.boxes), class_name, len(pred_boxes.all), len(pred_boxes.sample_tokens)))

    # Distances prediction -> same-sample ground truth, one matrix per sample.
    dist = {}
    fast = dist_fcn is center_distance
    for tok, sl in pred.slices.items():
        g = gt.slices.get(tok, slice(0, 0))
        if sl.stop == sl.start or g.stop == g.start:
            continue
        if fast:
            d = np.linalg.norm(pred.xy[sl, None, :] - gt.xy[None, g, :], axis=2)
        else:
            d = np.array([[dist_fcn(gb, pb) for gb in gt.boxes[g]] for pb in pred.boxes[sl]], dtype=float)
        dist[tok] = d

    taken = np.zeros(len(gt.boxes), dtype=bool)
    order = _confidence_order(pred.score)
    tp = np.zeros(len(order))
    match_gt = np.full(len(order), -1)
    for rank, p in enumerate(order):
        tok = pred.token[p]
        d = dist.get(tok)
        if d is None:
            continue
        g = gt.slices[tok]
        row = np.where(taken[g], np.inf, d[p - pred.slices[tok].start])
        j = int(np.argmin(row))                           # first minimum = the reference's strict `<` scan
        if row[j] < dist_th:
            taken[g.start + j] = True
            tp[rank], match_gt[rank] = 1, g.start + j
    if not tp.any():
        return DetectionMetricData.no_predictions()

    conf_sorted = pred.score[order]
    hit = tp.astype(bool)
    pi, gi = order[hit], match_gt[hit]
    match = {
        "trans_err": np.linalg.norm(pred.xy[pi] - gt.xy[gi], axis=1),
        "vel_err": np.linalg.norm(pred.vel[pi] - gt.vel[gi], axis=1),
        "scale_err": 1 - _scale_iou(gt.size[gi], pred.size[pi]),
        "orient_err": np.array([abs(angle_diff(a, b, np.pi if class_name == "barrier" else 2 * np.pi))
                                for a, b in zip(gt.yaw[gi], pred.yaw[pi])]),
        "conf": conf_sorted[hit],
    }

    tp_c, fp_c = np.cumsum(tp).astype(float), np.cumsum(1 - tp).astype(float)
    prec, rec = tp_c / (fp_c + tp_c), tp_c / float(npos)
    rec_interp = np.linspace(0, 1, DetectionMetricData.nelem)
    prec = np.interp(rec_interp, rec, prec, right=0)
    conf = np.interp(rec_interp, rec, conf_sorted, right=0)
    for key in ("trans_err", "vel_err", "scale_err", "orient_err"):
        running = cummean(match[key])
        match[key] = np.interp(conf[::-1], match["conf"][::-1], running[::-1])[::-1]
    return DetectionMetricData(recall=rec_interp, precision=prec, confidence=conf, trans_err=match["trans_err"],
                               vel_err=match["vel_err"], scale_err=match["scale_err"], orient_err=match["orient_err"])


def _scale_iou(sa, sr):
    assert (sa > 0).all(), "Error: sample_annotation sizes must be >0."
    assert (sr > 0).all(), "Error: sample_result sizes must be >0."
    inter = np.prod(np.minimum(sa, sr), axis=1)
    return inter / (np.prod(sa, axis=1) + np.prod(sr, axis=1) - inter)


def calc_ap(md: DetectionMetricData, min_recall: float, min_precision: float) -> float:
    """Mean of the precision above ``min_precision`` over the recall bins above ``min_recall``, rescaled to [0, 1]."""
    assert 0 <= min_precision < 1
    assert 0 <= min_recall <= 1
    prec = np.copy(md.precision)[round(100 * min_recall) + 1:]
    prec = np.clip(prec - min_precision, 0, None)
    return float(np.mean(prec)) / (1.0 - min_precision)


def calc_tp(md: DetectionMetricData, min_recall: float, metric_name: str) -> float:
    """Mean TP error between ``min_recall`` and the highest recall reached; 1.0 when that range is empty."""
    first = round(100 * min_recall) + 1
    last = md.max_recall_ind
    if last < first:
        return 1.0
    return float(np.mean(getattr(md, metric_name)[first:last + 1]))
