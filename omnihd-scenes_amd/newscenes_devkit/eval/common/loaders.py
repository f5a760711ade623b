"""Result loading and box filtering (reference: newscenes_devkit/eval/common/loaders.py:
load_prediction :22-49, filter_eval_boxes :178-230).  Ground truth comes from the dataset's info
records (``gt_boxes_from_infos``) instead of the NewScenes database tables, which are dataset IO and
out of scope; the produced DetectionBoxes are what reference ``load_gt`` :118-140 builds."""
import json

import numpy as np

from newscenes_devkit.eval.common.data_classes import EvalBoxes


def load_prediction(result_path, max_boxes_per_sample, box_cls, verbose=False):
    if isinstance(result_path, dict):
        data = result_path
    else:
        with open(result_path) as f:
            data = json.load(f)
    assert "results" in data, "Error: No field `results` in result file. Please note that the result format changed."
    all_results = EvalBoxes.deserialize(data["results"], box_cls)
    if verbose:
        print("Loaded results from {}. Found detections for {} samples.".format(
            "<dict>" if isinstance(result_path, dict) else result_path, len(all_results.sample_tokens)))
    for tok in all_results.sample_tokens:
        assert len(all_results.boxes[tok]) <= max_boxes_per_sample, \
            "Error: Only <= %d boxes per sample allowed!" % max_boxes_per_sample
    return all_results, data["meta"]


def yaw_to_wxyz(yaw):
    """Quaternion (w, x, y, z) of a rotation by ``yaw`` about +z."""
    return (float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2)))


def filter_eval_boxes(newsc, eval_boxes, max_dist, verbose=False, bad_conditions=False, bad_condition_tokens=None):
    """Keep boxes with |x| <= range_x and |y| <= range_y of their class (ego frame) and visibility 1.
    ``bad_conditions`` keeps only the samples recorded in rain or at night: the reference looks that up
    in the database (``newsc``); here the caller passes the qualifying tokens."""
    total = dist_kept = vis_kept = 0
    for tok in eval_boxes.sample_tokens:
        total += len(eval_boxes[tok])
        eval_boxes.boxes[tok] = [b for b in eval_boxes[tok]
                                 if abs(b.ego_translation[0]) <= max_dist[b.detection_name][0]
                                 and abs(b.ego_translation[1]) <= max_dist[b.detection_name][1]]
        dist_kept += len(eval_boxes[tok])
        eval_boxes.boxes[tok] = [b for b in eval_boxes[tok] if b.visibility == 1]
        vis_kept += len(eval_boxes[tok])
    if verbose:
        print("=> Original number of boxes: %d" % total)
        print("=> After distance based filtering: %d" % dist_kept)
        print("=> After Camera visibility based filtering: %d" % vis_kept)
    if bad_conditions:
        if bad_condition_tokens is None:
            raise ValueError("bad_conditions=True needs bad_condition_tokens (no database access here)")
        for tok in list(eval_boxes.sample_tokens):
            if tok not in bad_condition_tokens:
                del eval_boxes.boxes[tok]
    return eval_boxes
