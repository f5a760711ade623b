"""Per-class curves sampled at 101 recall points and their container keyed by (class, match distance)
(reference: newscenes_devkit/eval/detection/data_classes.py:89-204, :413-435)."""
import numpy as np

from newscenes_devkit.eval.common.data_classes import MetricData

_MD_FIELDS = ("recall", "precision", "confidence", "trans_err", "vel_err", "scale_err", "orient_err")


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
