"""Eval metrics: the reference's only known-answer test is the __main__ of eval_det_iou.py:205-225
(2 GT unit squares, 1 prediction -> precision 1.0, recall 0.5, hmean 0.6667)."""
import numpy as np

from pytorchocr_amd.metrics import DetectionIoUEvaluator, DetMetric, RecMetric, levenshtein
from pytorchocr_amd.metrics.polygon import area, intersection_area, is_valid_simple, union_area


def test_reference_main_known_answer():
    ev = DetectionIoUEvaluator()
    gts = [[{"points": [(0, 0), (1, 0), (1, 1), (0, 1)], "text": 1234, "ignore": False},
            {"points": [(2, 2), (3, 2), (3, 3), (2, 3)], "text": 5678, "ignore": False}]]
    preds = [[{"points": [(0.1, 0.1), (1, 0), (1, 1), (0, 1)], "text": 123, "ignore": False}]]
    m = ev.combine_results([ev.evaluate_image(g, p) for g, p in zip(gts, preds)])
    assert m["precision"] == 1.0 and m["recall"] == 0.5 and abs(m["hmean"] - 2 / 3) < 1e-12


def test_polygon_primitives():
    sq = [(0, 0), (2, 0), (2, 2), (0, 2)]
    assert area(sq) == 4 and area(sq[::-1]) == 4
    assert abs(intersection_area(sq, [(1, 1), (3, 1), (3, 3), (1, 3)]) - 1) < 1e-12
    assert abs(union_area(sq, [(1, 1), (3, 1), (3, 3), (1, 3)]) - 7) < 1e-12
    assert intersection_area(sq, [(5, 5), (6, 5), (6, 6), (5, 6)]) == 0
    # concave (arrow-head) quad against a square, orientation independent
    arrow = [(0, 0), (4, 2), (0, 4), (1, 2)]
    box = [(0, 1), (4, 1), (4, 3), (0, 3)]
    a1 = intersection_area(arrow, box)
    assert abs(a1 - intersection_area(arrow[::-1], box[::-1])) < 1e-12 and 0 < a1 < area(arrow)
    # rotated squares: diamond of diagonal 2 inside the 2x2 square
    assert abs(intersection_area(sq, [(1, 0), (2, 1), (1, 2), (0, 1)]) - 2) < 1e-12
    assert is_valid_simple(sq) and is_valid_simple(arrow)
    assert not is_valid_simple([(0, 0), (2, 2), (2, 0), (0, 2)])        # bow-tie
    assert not is_valid_simple([(0, 0), (1, 1), (2, 2), (3, 3)])        # zero area
    assert not is_valid_simple([(0, 0), (1, 0)])


def test_det_metric_ignore_and_matching():
    m = DetMetric()
    gt = np.array([[[[0, 0], [10, 0], [10, 10], [0, 10]], [[20, 0], [30, 0], [30, 10], [20, 10]], [[50, 50], [60, 50], [60, 60], [50, 60]]]])
    ign = np.array([[False, True, False]])
    pred = [{"points": np.array([[[1, 0], [10, 0], [10, 10], [1, 10]],          # matches gt 0
                                 [[20, 0], [29, 0], [29, 10], [20, 10]],        # falls on the ignored gt -> don't care
                                 [[80, 80], [90, 80], [90, 90], [80, 90]]])}]   # false positive
    m(pred, [None, None, gt, ign])
    r = m.get_metric()
    assert abs(r["precision"] - 0.5) < 1e-12 and abs(r["recall"] - 0.5) < 1e-12 and abs(r["hmean"] - 0.5) < 1e-12
    assert m.results == []


def test_rec_metric():
    assert levenshtein("kitten", "sitting") == 3 and levenshtein("", "abc") == 3 and levenshtein("abc", "abc") == 0
    m = RecMetric()
    out = m(([("abc", 0.9), ("a b", 0.5), ("xyz", 0.1)], [("abc", 1), ("ab", 1), ("xy", 1)]))
    assert abs(out["acc"] - 2 / 3) < 1e-12
    g = m.get_metric()
    assert abs(g["acc"] - 2 / 3.001) < 1e-9 and abs(g["norm_edit_dis"] - (1 - (1 / 3) / 3.001)) < 1e-9
    f = RecMetric(is_filter=True)
    assert f(([("A-b_C!", 1.0)], [("abc", 1)]))["acc"] == 1.0
