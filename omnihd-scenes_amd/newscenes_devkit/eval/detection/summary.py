"""Per-class APs / TP errors and the benchmark's summaries: mAP, mATE.., NOS
(reference: newscenes_devkit/eval/detection/data_classes.py:207-330, NOS formula :279-291)."""
from collections import defaultdict

import numpy as np

from newscenes_devkit.eval.detection.constants import TP_METRICS
from newscenes_devkit.eval.detection.settings import DetectionConfig


class DetectionMetrics:
    """Per-class APs and TP errors with the benchmark's summaries (mAP, mATE.., NOS)."""

    def __init__(self, cfg: DetectionConfig):
        self.cfg = cfg
        self._label_aps = defaultdict(lambda: defaultdict(float))
        self._label_tp_errors = defaultdict(lambda: defaultdict(float))
        self.eval_time = None

    def add_label_ap(self, detection_name, dist_th, ap):
        self._label_aps[detection_name][dist_th] = ap

    def get_label_ap(self, detection_name, dist_th):
        return self._label_aps[detection_name][dist_th]

    def add_label_tp(self, detection_name, metric_name, tp):
        self._label_tp_errors[detection_name][metric_name] = tp

    def get_label_tp(self, detection_name, metric_name):
        return self._label_tp_errors[detection_name][metric_name]

    def add_runtime(self, eval_time):
        self.eval_time = eval_time

    @property
    def mean_dist_aps(self):
        return {name: np.mean(list(d.values())) for name, d in self._label_aps.items()}

    @property
    def mean_ap(self) -> float:
        return float(np.mean(list(self.mean_dist_aps.values())))

    @property
    def tp_errors(self):
        return {m: float(np.nanmean([self.get_label_tp(n, m) for n in self.cfg.class_names])) for m in TP_METRICS}

    @property
    def tp_scores(self):
        return {m: max(0.0, 1.0 - e) for m, e in self.tp_errors.items()}

    @property
    def no_score(self) -> float:
        """NewScenes Overall Score = (w * mAP + sum of TP scores) / (w + number of TP metrics)."""
        scores = self.tp_scores
        total = float(self.cfg.mean_ap_weight * self.mean_ap + np.sum(list(scores.values())))
        return total / float(self.cfg.mean_ap_weight + len(scores))

    def serialize(self):
        return {"label_aps": self._label_aps, "mean_dist_aps": self.mean_dist_aps, "mean_ap": self.mean_ap,
                "label_tp_errors": self._label_tp_errors, "tp_errors": self.tp_errors, "tp_scores": self.tp_scores,
                "NOS": self.no_score, "eval_time": self.eval_time, "cfg": self.cfg.serialize()}

    @classmethod
    def deserialize(cls, content: dict):
        metrics = cls(cfg=DetectionConfig.deserialize(content["cfg"]))
        metrics.add_runtime(content["eval_time"])
        for name, aps in content["label_aps"].items():
            for dist_th, ap in aps.items():
                metrics.add_label_ap(name, float(dist_th), float(ap))
        for name, tps in content["label_tp_errors"].items():
            for metric, tp in tps.items():
                metrics.add_label_tp(name, metric, float(tp))
        return metrics

    def __eq__(self, other):
        return (self._label_aps == other._label_aps and self._label_tp_errors == other._label_tp_errors
                and self.eval_time == other.eval_time and self.cfg == other.cfg)
