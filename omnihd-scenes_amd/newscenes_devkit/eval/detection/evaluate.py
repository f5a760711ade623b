"""The evaluation driver (reference: newscenes_devkit/eval/detection/evaluate.py: NewScenesEval
__init__ :48-107, evaluate :109-154, main :186-260).  Differences: ground truth is handed in as
EvalBoxes (see eval/common/loaders.py), nothing is rendered, and files are written only when an
``output_dir`` is given."""
import json
import os
import time


from newscenes_devkit.eval.common.loaders import filter_eval_boxes, load_prediction
from newscenes_devkit.eval.detection.algo import accumulate, calc_ap, calc_tp
from newscenes_devkit.eval.detection.constants import TP_METRICS
from newscenes_devkit.eval.detection.data_classes import (DetectionBox, DetectionConfig, DetectionMetricDataList,
                                                          DetectionMetrics)


class NewScenesEval:
    def __init__(self, gt_boxes, config: DetectionConfig, result_path, output_dir=None, verbose=True,
                 bad_conditions=False, bad_condition_tokens=None):
        self.cfg, self.output_dir, self.verbose = config, output_dir, verbose
        self.pred_boxes, self.meta = load_prediction(result_path, config.max_boxes_per_sample, DetectionBox, verbose)
        self.gt_boxes = gt_boxes
        assert set(self.pred_boxes.sample_tokens) == set(self.gt_boxes.sample_tokens), \
            "Samples in split doesn't match samples in predictions."
        kw = dict(verbose=verbose, bad_conditions=bad_conditions, bad_condition_tokens=bad_condition_tokens)
        self.pred_boxes = filter_eval_boxes(None, self.pred_boxes, config.class_range, **kw)
        self.gt_boxes = filter_eval_boxes(None, self.gt_boxes, config.class_range, **kw)
        assert set(self.pred_boxes.sample_tokens) == set(self.gt_boxes.sample_tokens), \
            "Samples in split doesn't match samples in predictions."
        self.sample_tokens = self.gt_boxes.sample_tokens

    def evaluate(self):
        start = time.time()
        mdl = DetectionMetricDataList()
        for name in self.cfg.class_names:
            for dist_th in self.cfg.dist_ths:
                mdl.set(name, dist_th, accumulate(self.gt_boxes, self.pred_boxes, name, self.cfg.dist_fcn_callable,
                                                  dist_th, verbose=False))
        metrics = DetectionMetrics(self.cfg)
        for name in self.cfg.class_names:
            for dist_th in self.cfg.dist_ths:
                metrics.add_label_ap(name, dist_th, calc_ap(mdl[(name, dist_th)], self.cfg.min_recall,
                                                            self.cfg.min_precision))
            for metric in TP_METRICS:
                metrics.add_label_tp(name, metric, calc_tp(mdl[(name, self.cfg.dist_th_tp)], self.cfg.min_recall, metric))
        metrics.add_runtime(time.time() - start)
        return metrics, mdl

    def main(self, plot_examples=0, render_curves=False):
        metrics, mdl = self.evaluate()
        summary = metrics.serialize()
        summary["meta"] = dict(self.meta)
        if self.output_dir:
            os.makedirs(self.output_dir, exist_ok=True)
            with open(os.path.join(self.output_dir, "metrics_summary.json"), "w") as f:
                json.dump(summary, f, indent=2)
            with open(os.path.join(self.output_dir, "metrics_details.json"), "w") as f:
                json.dump(mdl.serialize(), f, indent=2)
        if self.verbose:
            names = {"trans_err": "mATE", "scale_err": "mASE", "orient_err": "mAOE", "vel_err": "mAVE"}
            print("mAP: %.4f" % summary["mean_ap"])
            for k, v in summary["tp_errors"].items():
                print("%s: %.4f" % (names[k], v))
            print("NOS: %.4f" % summary["NOS"])
            print("%-20s\t%-6s\t%-6s\t%-6s\t%-6s\t%-6s" % ("Object Class", "AP", "ATE", "ASE", "AOE", "AVE"))
            for c, ap in summary["mean_dist_aps"].items():
                e = summary["label_tp_errors"][c]
                print("%-20s\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f\t%-6.3f" % (c, ap, e["trans_err"], e["scale_err"],
                                                                        e["orient_err"], e["vel_err"]))
        return summary
