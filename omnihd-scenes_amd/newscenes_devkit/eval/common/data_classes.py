"""EvalBox / EvalBoxes containers (reference: newscenes_devkit/eval/common/data_classes.py:11-127)."""
import abc

import numpy as np


class EvalBox(abc.ABC):
    def __init__(self, sample_token="", translation=(0, 0, 0), size=(0, 0, 0), rotation=(0, 0, 0, 0), velocity=(0, 0),
                 ego_translation=(0, 0, 0), num_pts=-1):
        assert type(sample_token) == str, "Error: sample_token must be a string!"
        for name, val, n in (("Translation", translation, 3), ("Size", size, 3), ("Rotation", rotation, 4),
                             ("Translation", ego_translation, 3)):
            assert len(val) == n, f"Error: {name} must have {n} elements!"
            assert not np.any(np.isnan(val)), f"Error: {name} may not be NaN!"
        assert len(velocity) == 2, "Error: Velocity must have 2 elements!"
        assert type(num_pts) == int, "Error: num_pts must be int!"
        self.sample_token, self.translation, self.size, self.rotation = sample_token, translation, size, rotation
        self.velocity, self.ego_translation, self.num_pts = velocity, ego_translation, num_pts

    @property
    def ego_dist(self) -> float:
        return float(np.sqrt(np.sum(np.array(self.ego_translation[:2]) ** 2)))

    def __repr__(self):
        return str(self.serialize())

    @abc.abstractmethod
    def serialize(self) -> dict:
        ...

    @classmethod
    @abc.abstractmethod
    def deserialize(cls, content: dict):
        ...


class MetricData(abc.ABC):
    @abc.abstractmethod
    def serialize(self):
        ...

    @classmethod
    @abc.abstractmethod
    def deserialize(cls, content: dict):
        ...


from newscenes_devkit.eval.common.collection import EvalBoxes  # noqa: E402,F401  (reference import path)
