"""Evaluation settings record (reference: newscenes_devkit/eval/detection/data_classes.py:17-86)."""
import numpy as np

from newscenes_devkit.eval.common.utils import center_distance
from newscenes_devkit.eval.detection.constants import DETECTION_NAMES

_CFG_FIELDS = ("class_range", "dist_fcn", "dist_ths", "dist_th_tp", "min_recall", "min_precision",
               "max_boxes_per_sample", "mean_ap_weight")


class DetectionConfig:
    def __init__(self, class_range, dist_fcn, dist_ths, dist_th_tp, min_recall, min_precision, max_boxes_per_sample,
                 mean_ap_weight):
        assert set(class_range.keys()) == set(DETECTION_NAMES), "Class count mismatch."
        assert dist_th_tp in dist_ths, "dist_th_tp must be in set of dist_ths."
        for k, v in zip(_CFG_FIELDS, (class_range, dist_fcn, dist_ths, dist_th_tp, min_recall, min_precision,
                                      max_boxes_per_sample, mean_ap_weight)):
            setattr(self, k, v)
        self.class_names = self.class_range.keys()

    def serialize(self) -> dict:
        return {k: getattr(self, k) for k in _CFG_FIELDS}

    @classmethod
    def deserialize(cls, content: dict):
        return cls(*[content[k] for k in _CFG_FIELDS])

    def __eq__(self, other):
        return all(np.array_equal(getattr(self, k), getattr(other, k)) for k in _CFG_FIELDS)

    @property
    def dist_fcn_callable(self):
        if self.dist_fcn == "center_distance":
            return center_distance
        raise Exception("Error: Unknown distance function %s!" % self.dist_fcn)
