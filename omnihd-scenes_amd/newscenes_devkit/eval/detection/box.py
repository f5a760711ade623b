"""One ground-truth or predicted box of the detection evaluation
(reference: newscenes_devkit/eval/detection/data_classes.py:333-410)."""
import numpy as np

from newscenes_devkit.eval.common.data_classes import EvalBox
from newscenes_devkit.eval.detection.constants import ATTRIBUTE_NAMES, DETECTION_NAMES


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
