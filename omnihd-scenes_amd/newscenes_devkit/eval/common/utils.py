"""Box-pair measures of the detection evaluation (reference: newscenes_devkit/eval/common/utils.py:
center_distance :15, velocity_l2 :26, yaw_diff :38, angle_diff :53, scale_iou :90, quaternion_yaw
:116, cummean :158).  pyquaternion is not a dependency here: a (w, x, y, z) tuple is turned into a
yaw with the first column of its rotation matrix, normalising the quaternion first as pyquaternion's
``rotation_matrix`` does (an all-zero quaternion stays zero -> yaw 0)."""
import numpy as np


def center_distance(gt_box, pred_box) -> float:
    """L2 distance of the box centres in the ground plane."""
    return float(np.linalg.norm(np.array(pred_box.translation[:2]) - np.array(gt_box.translation[:2])))


def velocity_l2(gt_box, pred_box) -> float:
    return float(np.linalg.norm(np.array(pred_box.velocity) - np.array(gt_box.velocity)))


def quaternion_yaw_wxyz(q) -> float:
    q = np.asarray(getattr(q, "elements", q), dtype=np.float64)
    n2 = float(np.dot(q, q))
    if abs(1.0 - n2) >= 1e-14 and n2 > 0:
        q = q / np.sqrt(n2)
    w, x, y, z = q
    return float(np.arctan2(2 * (x * y + z * w), w * w + x * x - y * y - z * z))


quaternion_yaw = quaternion_yaw_wxyz


def angle_diff(x: float, y: float, period: float) -> float:
    """Signed smallest difference x - y for angles of the given periodicity, in (-pi, pi]."""
    diff = (x - y + period / 2) % period - period / 2
    if diff > np.pi:
        diff = diff - (2 * np.pi)
    return diff


def yaw_diff(gt_box, eval_box, period: float = 2 * np.pi) -> float:
    return abs(angle_diff(quaternion_yaw_wxyz(gt_box.rotation), quaternion_yaw_wxyz(eval_box.rotation), period))


def scale_iou(sample_annotation, sample_result) -> float:
    """IoU of the two boxes once centres and headings are aligned."""
    sa, sr = np.array(sample_annotation.size, dtype=float), np.array(sample_result.size, dtype=float)
    assert all(sa > 0), "Error: sample_annotation sizes must be >0."
    assert all(sr > 0), "Error: sample_result sizes must be >0."
    inter = np.prod(np.minimum(sa, sr))
    return float(inter / (np.prod(sa) + np.prod(sr) - inter))


def cummean(x: np.ndarray) -> np.ndarray:
    """Running mean that skips NaNs; all-NaN input -> ones (error 1 at every operating point)."""
    x = np.asarray(x, dtype=float)
    nan = np.isnan(x)
    if nan.all():
        return np.ones(len(x))
    total = np.nancumsum(x)
    count = np.cumsum(~nan)
    return np.divide(total, count, out=np.zeros_like(total), where=count != 0)
