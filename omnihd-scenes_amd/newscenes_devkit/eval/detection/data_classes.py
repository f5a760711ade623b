"""Detection evaluation records (reference: newscenes_devkit/eval/detection/data_classes.py:
DetectionConfig :17-86, DetectionMetricData :89-204, DetectionMetrics :207-330 incl. the NOS formula
:279-291, DetectionBox :333-410, DetectionMetricDataList :413-435)."""
from collections import defaultdict

import numpy as np

from newscenes_devkit.eval.common.data_classes import EvalBox, MetricData
from newscenes_devkit.eval.common.utils import center_distance
from newscenes_devkit.eval.detection.constants import ATTRIBUTE_NAMES, DETECTION_NAMES, TP_METRICS

_CFG_FIELDS = ("class_range", "dist_fcn", "dist_ths", "dist_th_tp", "min_recall", "min_precision",
               "max_boxes_per_sample", "mean_ap_weight")
_MD_FIELDS = ("recall", "precision", "confidence", "trans_err", "vel_err", "scale_err", "orient_err")


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


class DetectionMetricData(MetricData):
    """Curves sampled at 101 recall points."""
    nelem = 101

    def __init__(self, recall, precision, confidence, trans_err, vel_err, scale_err, orient_err):
        for name, arr in zip(_MD_FIELDS, (recall, precision, confidence, trans_err, vel_err, scale_err, orient_err)):
            assert len(arr) == self.nelem, name
            setattr(self, name, arr)
        assert all(confidence == sorted(confidence, reverse=True))          # descending confidences
        assert all(recall == sorted(recall))                                # ascending recalls

    def __eq__(self, other):
        return all(np.array_equal(getattr(self, k), getattr(other, k)) for k in _MD_FIELDS)

    @property
    def max_recall_ind(self):
        nz = np.nonzero(self.confidence)[0]
        return 0 if len(nz) == 0 else nz[-1]

    @property
    def max_recall(self):
        return self.recall[self.max_recall_ind]

    def serialize(self):
        return {k: getattr(self, k).tolist() for k in _MD_FIELDS}

    @classmethod
    def deserialize(cls, content: dict):
        return cls(**{k: np.array(content[k]) for k in _MD_FIELDS})

    @classmethod
    def no_predictions(cls):
        one = lambda: np.ones(cls.nelem)
        return cls(recall=np.linspace(0, 1, cls.nelem), precision=np.zeros(cls.nelem), confidence=np.zeros(cls.nelem),
                   trans_err=one(), vel_err=one(), scale_err=one(), orient_err=one())

    @classmethod
    def random_md(cls):
        r = lambda: np.random.random(cls.nelem)
        return cls(recall=np.linspace(0, 1, cls.nelem), precision=r(), confidence=np.linspace(0, 1, cls.nelem)[::-1],
                   trans_err=r(), vel_err=r(), scale_err=r(), orient_err=r())


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


class DetectionBox(EvalBox):
    _FIELDS = ("sample_token", "translation", "size", "rotation", "velocity", "ego_translation", "num_pts",
               "detection_name", "detection_score", "attribute_name", "visibility")

    def __init__(self, sample_token="", translation=(0, 0, 0), size=(0, 0, 0), rotation=(0, 0, 0, 0), velocity=(0, 0),
                 ego_translation=(0, 0, 0), num_pts=-1, detection_name="car", detection_score=-1.0, attribute_name="",
                 visibility=1):
        super().__init__(sample_token, translation, size, rotation, velocity, ego_translation, num_pts)
        assert detection_name is not None, "Error: detection_name cannot be empty!"
        assert detection_name in DETECTION_NAMES, "Error: Unknown detection_name %s" % detection_name
        assert attribute_name in ATTRIBUTE_NAMES or attribute_name == "", "Error: Unknown attribute_name %s" % attribute_name
        assert type(detection_score) == float, "Error: detection_score must be a float!"
        assert not np.any(np.isnan(detection_score)), "Error: detection_score may not be NaN!"
        self.detection_name, self.detection_score = detection_name, detection_score
        self.attribute_name, self.visibility = attribute_name, visibility

    def __eq__(self, other):
        return all(getattr(self, k) == getattr(other, k) for k in self._FIELDS)

    def serialize(self) -> dict:
        return {k: getattr(self, k) for k in self._FIELDS}

    @classmethod
    def deserialize(cls, content: dict):
        # As in the reference (:396-410) the ego translation of a loaded box IS its translation: results and
        # annotations are both expressed in the ego/LiDAR frame of their sample.
        return cls(sample_token=content["sample_token"], translation=tuple(content["translation"]),
                   size=tuple(content["size"]), rotation=tuple(content["rotation"]), velocity=tuple(content["velocity"]),
                   ego_translation=tuple(content["translation"]),
                   num_pts=-1 if "num_pts" not in content else int(content["num_pts"]),
                   detection_name=content["detection_name"],
                   detection_score=-1.0 if "detection_score" not in content else float(content["detection_score"]),
                   attribute_name="" if "attribute_name" not in content else content["attribute_name"],
                   visibility=1 if "visibility" not in content else content["visibility"])


class DetectionMetricDataList:
    """MetricData keyed by (class name, match distance)."""

    def __init__(self):
        self.md = {}

    def __getitem__(self, key):
        return self.md[key]

    def __eq__(self, other):
        return self.md.keys() == other.md.keys() and all(self.md[k] == other.md[k] for k in self.md)

    def get_class_data(self, detection_name):
        return [(md, dist_th) for (name, dist_th), md in self.md.items() if name == detection_name]

    def get_dist_data(self, dist_th):
        return [(md, name) for (name, dist), md in self.md.items() if dist == dist_th]

    def set(self, detection_name, match_distance, data):
        self.md[(detection_name, match_distance)] = data

    def serialize(self) -> dict:
        return {key[0] + ":" + str(key[1]): value.serialize() for key, value in self.md.items()}

    @classmethod
    def deserialize(cls, content: dict):
        mdl = cls()
        for key, md in content.items():
            name, distance = key.split(":")
            mdl.set(name, float(distance), DetectionMetricData.deserialize(md))
        return mdl
