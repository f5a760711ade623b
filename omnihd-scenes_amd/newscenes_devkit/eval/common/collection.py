"""Boxes grouped by sample token (reference: newscenes_devkit/eval/common/data_classes.py:62-127)."""
from collections import defaultdict


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
