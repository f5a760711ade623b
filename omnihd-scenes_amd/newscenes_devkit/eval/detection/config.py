"""Evaluation settings (reference: newscenes_devkit/eval/detection/config.py:10-29 reading
configs/detection_newsc_config_final.json — the version NewScenesDataset selects,
datasets/newscenes_dataset.py:98).  The JSON's content is restated here as data."""
from newscenes_devkit.eval.detection.data_classes import DetectionConfig

_CONFIGS = {
    "detection_newsc_config_final": {
        "class_range": {"car": [60, 40], "pedestrian": [60, 40], "rider": [60, 40], "large_vehicle": [60, 40]},
        "dist_fcn": "center_distance",
        "dist_ths": [1.0, 2.0, 3.0, 4.0],
        "dist_th_tp": 3.0,
        "min_recall": 0.1,
        "min_precision": 0.1,
        "max_boxes_per_sample": 500,
        "mean_ap_weight": 4,
    },
}


def config_factory(configuration_name: str) -> DetectionConfig:
    assert configuration_name in _CONFIGS, "Requested unknown configuration {}".format(configuration_name)
    cfg = _CONFIGS[configuration_name]
    return DetectionConfig.deserialize({k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
