"""Evaluation metrics next to the path: mirrors of reference pytocr/metrics/det_metric.py:6-54,
eval_det_iou.py:12-202 (ICDAR IoU >= 0.5 matching; P / R / hmean) and rec_metric.py:5-54 (exact-match accuracy and
1 - normalised edit distance).  shapely and python-Levenshtein are replaced by polygon.py and a DP edit distance."""
import string

import numpy as np

from .polygon import area, intersection_area, is_valid_simple, union_area

__all__ = ["DetectionIoUEvaluator", "ClsMetric", "DetMetric", "RecMetric", "build_metric", "levenshtein"]


class DetectionIoUEvaluator(object):
    def __init__(self, iou_constraint=0.5, area_precision_constraint=0.5):
        self.iou_constraint = iou_constraint
        self.area_precision_constraint = area_precision_constraint

    def evaluate_image(self, gt, pred):
        gtPols, detPols, gtDontCare, detDontCare = [], [], [], []
        for g in gt:
            if not is_valid_simple(g["points"]):
                continue
            gtPols.append(g["points"])
            if g["ignore"]:
                gtDontCare.append(len(gtPols) - 1)
        for d in pred:
            pts = d["points"]
            if not is_valid_simple(pts):
                continue
            detPols.append(pts)
            for k in gtDontCare:
                inter = intersection_area(gtPols[k], pts)
                dim = area(pts)
                precision = 0 if dim == 0 else inter / dim
                if precision > self.area_precision_constraint:
                    detDontCare.append(len(detPols) - 1)
                    break
        detMatched = 0
        if gtPols and detPols:
            iouMat = np.empty([len(gtPols), len(detPols)])
            for i, pG in enumerate(gtPols):
                for j, pD in enumerate(detPols):
                    iouMat[i, j] = intersection_area(pD, pG) / union_area(pD, pG)
            gtUsed = np.zeros(len(gtPols), np.int8)
            detUsed = np.zeros(len(detPols), np.int8)
            for i in range(len(gtPols)):
                for j in range(len(detPols)):
                    if gtUsed[i] == 0 and detUsed[j] == 0 and i not in gtDontCare and j not in detDontCare:
                        if iouMat[i, j] > self.iou_constraint:
                            gtUsed[i] = 1
                            detUsed[j] = 1
                            detMatched += 1
        return {"gtCare": len(gtPols) - len(gtDontCare), "detCare": len(detPols) - len(detDontCare), "detMatched": detMatched}

    def combine_results(self, results):
        numGt = sum(r["gtCare"] for r in results)
        numDet = sum(r["detCare"] for r in results)
        matched = sum(r["detMatched"] for r in results)
        recall = 0 if numGt == 0 else float(matched) / numGt
        precision = 0 if numDet == 0 else float(matched) / numDet
        hmean = 0 if recall + precision == 0 else 2 * recall * precision / (recall + precision)
        return {"precision": precision, "recall": recall, "hmean": hmean}


class DetMetric(object):
    def __init__(self, main_indicator="hmean", **kwargs):
        self.evaluator = DetectionIoUEvaluator()
        self.main_indicator = main_indicator
        self.reset()

    def __call__(self, preds, batch, **kwargs):
        for pred, gt_polyons, ignore_tags in zip(preds, batch[2], batch[3]):
            gt_info = [{"points": p, "text": "", "ignore": t} for p, t in zip(gt_polyons, ignore_tags)]
            det_info = [{"points": p, "text": ""} for p in pred["points"]]
            self.results.append(self.evaluator.evaluate_image(gt_info, det_info))

    def get_metric(self):
        m = self.evaluator.combine_results(self.results)
        self.reset()
        return m

    def reset(self):
        self.results = []


def levenshtein(a, b):
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class RecMetric(object):
    def __init__(self, main_indicator="acc", is_filter=False, **kwargs):
        self.main_indicator = main_indicator
        self.is_filter = is_filter
        self.reset()

    def _normalize_text(self, text):
        return "".join(filter(lambda x: x in (string.digits + string.ascii_letters), text)).lower()

    def __call__(self, pred_label, *args, **kwargs):
        preds, labels = pred_label
        correct_num, all_num, norm_edit_dis = 0, 0, 0.0
        for (pred, _), (target, _) in zip(preds, labels):
            pred, target = pred.replace(" ", ""), target.replace(" ", "")
            if self.is_filter:
                pred, target = self._normalize_text(pred), self._normalize_text(target)
            norm_edit_dis += levenshtein(pred, target) / max(len(pred), len(target), 1)
            correct_num += pred == target
            all_num += 1
        self.correct_num += correct_num
        self.all_num += all_num
        self.norm_edit_dis += norm_edit_dis
        return {"acc": correct_num / all_num, "norm_edit_dis": 1 - norm_edit_dis / (all_num + 1e-3)}

    def get_metric(self):
        acc = 1.0 * self.correct_num / (self.all_num + 1e-3)
        ned = 1 - self.norm_edit_dis / (self.all_num + 1e-3)
        self.reset()
        return {"acc": acc, "norm_edit_dis": ned}

    def reset(self):
        self.correct_num, self.all_num, self.norm_edit_dis = 0, 0, 0


class ClsMetric(object):
    """reference pytocr/metrics/cls_metric.py:1-30: share of lines whose predicted direction equals the label"""

    def __init__(self, main_indicator="acc", **kwargs):
        self.main_indicator = main_indicator
        self.reset()

    def __call__(self, pred_label, *args, **kwargs):
        preds, labels = pred_label
        hits = sum(1 for (pred, _), (target, _) in zip(preds, labels) if pred == target)
        n = min(len(preds), len(labels))
        self.correct_num += hits
        self.all_num += n
        return {"acc": hits / n}

    def get_metric(self):
        acc = self.correct_num / self.all_num
        self.reset()
        return {"acc": acc}

    def reset(self):
        self.correct_num, self.all_num = 0, 0


def build_metric(config):
    config = dict(config)
    name = config.pop("name")
    support = {"DetMetric": DetMetric, "RecMetric": RecMetric, "ClsMetric": ClsMetric}
    assert name in support, "metric only support {} (pytorchocr_amd)".format(list(support))
    return support[name](**config)
