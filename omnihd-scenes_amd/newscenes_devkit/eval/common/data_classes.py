"""EvalBox / EvalBoxes containers (reference: newscenes_devkit/eval/common/data_classes.py:11-127)."""
import abc
from collections import defaultdict

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


class EvalBoxes:
    """Boxes grouped by sample token, insertion-ordered."""

    def __init__(self):
        self.boxes = defaultdict(list)

    def __repr__(self):
        return "EvalBoxes with {} boxes across {} samples".format(len(self.all), len(self.sample_tokens))

    def __getitem__(self, item):
        return self.boxes[item]

    def __len__(self):
        return len(self.boxes)

    def __eq__(self, other):
        if set(self.sample_tokens) != set(other.sample_tokens):
            return False
        return all(len(self[t]) == len(other[t]) and all(a == b for a, b in zip(self[t], other[t]))
                   for t in self.sample_tokens)

    @property
    def all(self):
        return [b for t in self.sample_tokens for b in self[t]]

    @property
    def sample_tokens(self):
        return list(self.boxes.keys())

    def add_boxes(self, sample_token, boxes):
        self.boxes[sample_token].extend(boxes)

    def serialize(self) -> dict:
        return {key: [box.serialize() for box in boxes] for key, boxes in self.boxes.items()}

    @classmethod
    def deserialize(cls, content: dict, box_cls):
        eb = cls()
        for sample_token, boxes in content.items():
            eb.add_boxes(sample_token, [box_cls.deserialize(box) for box in boxes])
        return eb


class MetricData(abc.ABC):
    @abc.abstractmethod
    def serialize(self):
        ...

    @classmethod
    @abc.abstractmethod
    def deserialize(cls, content: dict):
        ...
