"""The reference keeps every record of the detection evaluation in this one module
(newscenes_devkit/eval/detection/data_classes.py); here each lives in its own file and this module re-exports them
under the reference's import path."""
from newscenes_devkit.eval.detection.box import DetectionBox  # noqa: F401
from newscenes_devkit.eval.detection.curves import DetectionMetricData, DetectionMetricDataList  # noqa: F401
from newscenes_devkit.eval.detection.settings import DetectionConfig  # noqa: F401
from newscenes_devkit.eval.detection.summary import DetectionMetrics  # noqa: F401
