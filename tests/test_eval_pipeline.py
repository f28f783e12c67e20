"""Label operators, variable-width recognition batching and the evaluation loop: results recorded from
the reference's own classes (tests/golden/label_encode.json, tools/gen_golden.py) and hand-derived cases."""
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DICT = os.path.join(ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt")


def test_ctc_and_cls_label_encode_match_the_reference(gold_dir):
    from pytorchocr_amd.data.label_ops import ClsLabelEncode, CTCLabelEncode
    cases = json.load(open(os.path.join(gold_dir, "label_encode.json"), encoding="utf-8"))
    assert len(cases) >= 30
    encs = {}
    for c in cases:
        if c["enc"] == "cls":
            r = ClsLabelEncode(label_list=["0", "180"])({"label": c["text"]})
            assert (r is None and c["out"] is None) or r["label"] == c["out"]
            continue
        key = json.dumps(c["kw"], sort_keys=True) + c["enc"]
        if key not in encs:
            encs[key] = CTCLabelEncode(character_dict_path=None if c["enc"] == "default36" else DICT, **c["kw"])
        enc = encs[key]
        assert len(enc.character) == c["nclass"]
        r = enc({"label": c["text"]})
        if c["out"] is None:
            assert r is None, c
            continue
        assert r["label"].tolist() == c["out"]["label"] and int(r["length"]) == c["out"]["length"]
        assert {str(i): int(v) for i, v in enumerate(r["label_ace"].tolist()) if v} == c["out"]["ace_nonzero"]


def test_det_label_encode_hand_case():
    from pytorchocr_amd.data.label_ops import DetLabelEncode
    lab = json.dumps([{"points": [[1, 2], [30, 2], [30, 12], [1, 12]], "transcription": "abc"},
                      {"points": [[5, 20], [9, 20], [9, 26]], "transcription": "###"}])
    d = DetLabelEncode(ignore_txt=["###", "#####"])({"label": lab})
    assert d["polys"].dtype == np.float32 and d["polys"].shape == (2, 4, 2)
    assert d["polys"][1].tolist() == [[5, 20], [9, 20], [9, 26], [9, 26]]          # padded with its last point
    assert d["texts"] == ["abc", "###"] and d["ignore_tags"].tolist() == [False, True]
    assert DetLabelEncode()({"label": "[]"}) is None


def test_rec_resize_for_test_batches_by_width():
    from pytorchocr_amd.data.imaug import RecResizeImgForTest, resize_norm_img
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for h, w in ((32, 100), (16, 100), (40, 35), (32, 5000), (20, 61))]
    op = RecResizeImgForTest(imgC=1, imgH=32, max_w=1200, batch_size=2)
    out = op(imgs)
    assert [tuple(t.shape) for t in out] == [(2, 1, 32, 200), (2, 1, 32, 1200), (1, 1, 32, 98)]      # widths 100, 200 | 28, 1200 | 98
    assert torch.equal(out[0][0, :, :, :100], resize_norm_img(imgs[0], [1, 32, 100], resized_w=100)) and float(out[0][0, :, :, 100:].abs().max()) == 0
    assert torch.equal(out[1][0, :, :, :28], resize_norm_img(imgs[2], [1, 32, 28], resized_w=28))
    one = op(imgs[4])
    assert tuple(one.shape) == (1, 1, 32, 98)


def test_eval_loop_with_stub_model():
    """the loop's contract on the host: list batches, post-process(preds, batch[1]), metric(post, batch), fps"""
    from pytorchocr_amd.eval import eval as run_eval
    from pytorchocr_amd.metrics import build_metric

    class Stub(torch.nn.Module):
        def forward(self, x):
            return {"maps": x.mean()}

    gt = np.array([[[[0, 0], [10, 0], [10, 10], [0, 10]], [[20, 20], [30, 20], [30, 30], [20, 30]]]], np.float32)
    batch = [torch.zeros(1, 3, 32, 32), np.array([[32, 32, 1.0, 1.0]]), gt, np.array([[False, False]])]

    def post(preds, shape_list):            # finds the first box exactly, misses the second
        return [{"points": np.array([[[0, 0], [10, 0], [10, 10], [0, 10]]], np.int16)}]

    m = run_eval(Stub(), torch.device("cpu"), [batch, batch], post, build_metric(dict(name="DetMetric")), model_type="det")
    assert m["precision"] == 1.0 and m["recall"] == 0.5 and abs(m["hmean"] - 2 / 3) < 1e-9 and m["fps"] > 0
    with pytest.raises(NotImplementedError):
        run_eval(Stub(), torch.device("cpu"), [batch], post, build_metric(dict(name="DetMetric")), model_type="table")
