"""Class and metric names of the NewScenes detection task (reference: eval/detection/constants.py:10-11)."""
DETECTION_NAMES = ["car", "pedestrian", "rider", "large_vehicle"]
TP_METRICS = ["trans_err", "scale_err", "orient_err", "vel_err"]          # no attribute error in this benchmark
ATTRIBUTE_NAMES = [""]
PRETTY_DETECTION_NAMES = {"car": "Car", "pedestrian": "Pedestrian", "rider": "Rider", "large_vehicle": "Large_Vehicle"}
PRETTY_TP_METRICS = {"trans_err": "Trans.", "scale_err": "Scale", "orient_err": "Orient.", "vel_err": "Vel."}
TP_METRICS_UNITS = {"trans_err": "m", "scale_err": "1-IOU", "orient_err": "rad.", "vel_err": "m/s"}
