#!/usr/bin/env python3
"""Golden vectors for the detection evaluator, produced by running the
REFERENCE's own devkit code: newscenes_devkit/eval/detection/algo.py (accumulate, calc_ap, calc_tp),
data_classes.py (DetectionMetrics incl. NOS), config.py (detection_newsc_config_final) and
eval/common/loaders.py-style filtering arithmetic are imported from /root/reference and executed on
seeded random boxes; inputs and outputs are stored as arrays in tests/golden/eval_golden.npz.

Two third-party packages the devkit imports are absent from the image:
  * ``nuscenes`` — used only for type annotations in eval/common/utils.py:9-10 -> empty stand-in;
  * ``pyquaternion`` — ``Quaternion(rot).rotation_matrix`` inside ``quaternion_yaw``
    (eval/common/utils.py:116-131).  The stand-in below implements exactly that property with the
    textbook formula (normalise, then R from w,x,y,z); it influences only ``orient_err``.
Runs only in the authoring container.  Usage: python tests/golden/make_golden_eval.py
"""
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Quaternion:
    def __init__(self, *args, **kw):
        if kw:
            ax = np.asarray(kw["axis"], dtype=float)
            ang = kw.get("radians", kw.get("angle", 0.0))
            ax = ax / np.linalg.norm(ax)
            self.q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        else:
            a = args[0]
            self.q = np.asarray(a.q if isinstance(a, Quaternion) else a, dtype=float)

    @property
    def elements(self):
        return self.q

    @property
    def rotation_matrix(self):
        q = self.q
        n2 = float(np.dot(q, q))
        if abs(1.0 - n2) >= 1e-14 and n2 > 0:
            q = q / np.sqrt(n2)
        w, x, y, z = q
        return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), w * w - x * x + y * y - z * z, 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), w * w - x * x - y * y + z * z]])


def install_stubs():
    sys.modules["pyquaternion"] = types.ModuleType("pyquaternion")
    sys.modules["pyquaternion"].Quaternion = Quaternion
    for name in ("nuscenes", "nuscenes.eval", "nuscenes.eval.common", "nuscenes.eval.common.data_classes",
                 "nuscenes.utils", "nuscenes.utils.data_classes"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["nuscenes.eval.common.data_classes"].EvalBox = object
    sys.modules["nuscenes.utils.data_classes"].Box = object


def mock_boxes(rng, names, n_samples, n_gt, n_pred, tie_scores=False):
    """-> dict of arrays describing GT and predictions (ego frame, visibility flag on GT)."""
    def block(n_per, spread, is_gt):
        n = n_samples * n_per
        d = dict(sample=np.repeat(np.arange(n_samples), n_per), cls=rng.integers(0, len(names), n),
                 trans=np.concatenate([rng.uniform(-1, 1, (n, 2)) * spread, rng.uniform(-2, 2, (n, 1))], 1),
                 size=rng.uniform(0.3, 6.0, (n, 3)), rot=rng.uniform(-1, 1, (n, 4)), vel=rng.uniform(-5, 5, (n, 2)),
                 score=np.full(n, -1.0) if is_gt else rng.uniform(0.05, 1.0, n),
                 vis=(rng.uniform(0, 1, n) < 0.9).astype(np.int64) if is_gt else np.ones(n, np.int64))
        if tie_scores and not is_gt:
            d["score"] = np.round(d["score"], 1)                 # many equal confidences: exercises the tie rule
        return d
    gt, pred = block(n_gt, np.array([70.0, 45.0]), True), block(n_pred, np.array([70.0, 45.0]), False)
    # make a third of the predictions near-hits of a ground-truth box of the same sample
    for i in range(len(pred["sample"])):
        if rng.uniform() < 0.35:
            cands = np.nonzero(gt["sample"] == pred["sample"][i])[0]
            j = rng.choice(cands)
            pred["trans"][i] = gt["trans"][j] + rng.normal(0, 0.8, 3)
            pred["cls"][i] = gt["cls"][j]
            pred["size"][i] = gt["size"][j] * rng.uniform(0.8, 1.25, 3)
            pred["rot"][i] = gt["rot"][j] + rng.normal(0, 0.1, 4)
            pred["vel"][i] = gt["vel"][j] + rng.normal(0, 0.5, 2)
    return gt, pred


def to_eval_boxes(d, names, EvalBoxes, DetectionBox, n_samples):
    eb = EvalBoxes()
    for s in range(n_samples):
        rows = np.nonzero(d["sample"] == s)[0]
        eb.add_boxes(str(s), [DetectionBox(sample_token=str(s), translation=tuple(d["trans"][i].tolist()),
                                           size=tuple(d["size"][i].tolist()), rotation=tuple(d["rot"][i].tolist()),
                                           velocity=tuple(d["vel"][i].tolist()),
                                           ego_translation=tuple(d["trans"][i].tolist()),
                                           detection_name=names[d["cls"][i]], detection_score=float(d["score"][i]),
                                           visibility=int(d["vis"][i])) for i in rows])
    return eb


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from newscenes_devkit.eval.common.data_classes import EvalBoxes
    from newscenes_devkit.eval.common.utils import center_distance
    from newscenes_devkit.eval.detection.algo import accumulate, calc_ap, calc_tp
    from newscenes_devkit.eval.detection.config import config_factory
    from newscenes_devkit.eval.detection.constants import DETECTION_NAMES, TP_METRICS
    from newscenes_devkit.eval.detection.data_classes import DetectionBox, DetectionMetrics

    cfg = config_factory("detection_newsc_config_final")
    out = dict(names=np.array(DETECTION_NAMES), tp_metrics=np.array(TP_METRICS),
               cfg_dist_ths=np.array(cfg.dist_ths), cfg_scalars=np.array([cfg.dist_th_tp, cfg.min_recall, cfg.min_precision,
                                                                         cfg.max_boxes_per_sample, cfg.mean_ap_weight]),
               cfg_class_range=np.array([cfg.class_range[n] for n in DETECTION_NAMES], dtype=float))
    for case, (seed, ns, ng, npred, ties) in enumerate([(42, 30, 6, 40, False), (7, 12, 3, 25, True), (3, 5, 2, 0, False)]):
        rng = np.random.default_rng(seed)
        random.seed(seed)
        gt, pred = mock_boxes(rng, DETECTION_NAMES, ns, ng, npred, ties)
        for k, v in gt.items():
            out[f"c{case}_gt_{k}"] = v
        for k, v in pred.items():
            out[f"c{case}_pred_{k}"] = v
        out[f"c{case}_n_samples"] = np.array(ns)
        gtb = to_eval_boxes(gt, DETECTION_NAMES, EvalBoxes, DetectionBox, ns)
        prb = to_eval_boxes(pred, DETECTION_NAMES, EvalBoxes, DetectionBox, ns)
        # the reference's filter (loaders.py:196-206), applied with its own expressions
        for eb in (gtb, prb):
            for tok in eb.sample_tokens:
                eb.boxes[tok] = [b for b in eb[tok] if abs(b.ego_translation[0]) <= cfg.class_range[b.detection_name][0]
                                 and abs(b.ego_translation[1]) <= cfg.class_range[b.detection_name][1]]
                eb.boxes[tok] = [b for b in eb[tok] if b.visibility == 1]
        metrics = DetectionMetrics(cfg)
        for name in cfg.class_names:
            for th in cfg.dist_ths:
                md = accumulate(gtb, prb, name, center_distance, th, verbose=False)
                for f in ("recall", "precision", "confidence", "trans_err", "vel_err", "scale_err", "orient_err"):
                    out[f"c{case}_md_{name}_{th}_{f}"] = np.asarray(getattr(md, f), dtype=float)
                metrics.add_label_ap(name, th, calc_ap(md, cfg.min_recall, cfg.min_precision))
                if th == cfg.dist_th_tp:
                    for m in TP_METRICS:
                        metrics.add_label_tp(name, m, calc_tp(md, cfg.min_recall, m))
        out[f"c{case}_label_aps"] = np.array([[metrics.get_label_ap(n, th) for th in cfg.dist_ths] for n in DETECTION_NAMES])
        out[f"c{case}_label_tps"] = np.array([[metrics.get_label_tp(n, m) for m in TP_METRICS] for n in DETECTION_NAMES])
        out[f"c{case}_summary"] = np.array([metrics.mean_ap, metrics.no_score] + [metrics.tp_errors[m] for m in TP_METRICS])
        print(f"case {case}: mAP {metrics.mean_ap:.4f}  NOS {metrics.no_score:.4f}  tp_errors {metrics.tp_errors}")
    path = os.path.join(HERE, "eval_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
