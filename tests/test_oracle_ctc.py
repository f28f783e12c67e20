"""ctc_oracle against CTCLabelDecode results recorded from the reference class (tests/golden/ctc_decode.json)."""
import json
import math
import os

import numpy as np

from oracle import ctc_oracle
from pytorchocr_amd.utils.synth import uniform

DICT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorchocr_amd", "utils", "char_dict_6623.txt")


def test_ctc_known_answers(gold_dir):
    cases = json.load(open(os.path.join(gold_dir, "ctc_decode.json"), encoding="utf-8"))
    chars36 = ctc_oracle.load_characters(None)
    assert len(chars36) == 37
    n = 0
    for c in cases:
        if c["dict"] != "default36":
            continue
        T, C = c["T"], c["C"]
        pr = uniform((T, 1, C), c["seed"], 0.0, 0.5)
        for t, k in enumerate(c["seq"]):
            pr[t, 0, k] = 0.6 + 0.01 * t
        (text, conf), = ctc_oracle.ctc_label_decode(pr, chars36)
        assert text == c["text"]
        if c["conf"] is None:
            assert math.isnan(conf)
        else:
            assert abs(float(conf) - c["conf"]) < 1e-7
        n += 1
    assert n >= 9


def test_ctc_survey_known_answer():
    # indices [1,1,0,1,2,2] -> "001"; all blank -> ("", nan)
    chars = ctc_oracle.load_characters(None)
    idx = np.array([[1, 1, 0, 1, 2, 2], [0, 0, 0, 0, 0, 0]])
    prob = np.full(idx.shape, 0.5, np.float32)
    r = ctc_oracle.decode(idx, prob, chars)
    assert r[0][0] == "001" and r[1][0] == "" and math.isnan(r[1][1])


def test_ctc_big_dictionary(gold_dir):
    if not os.path.exists(DICT):
        import pytest
        pytest.skip("dictionary not present")
    cases = json.load(open(os.path.join(gold_dir, "ctc_decode.json"), encoding="utf-8"))
    c = [x for x in cases if x["dict"] == "char_dict_6623"][0]
    chars = ctc_oracle.load_characters(DICT)
    assert len(chars) == c["nclass"] == 6624
    T, C = c["T"], c["C"]
    pr = np.zeros((T, 2, C), np.float32)
    for t, k in enumerate(c["seq"]):
        pr[t, 0, k] = 0.9
        pr[t, 1, (k * 7 + 1) % C] = 0.8
    res = ctc_oracle.ctc_label_decode(pr, chars)
    assert [r[0] for r in res] == c["text"]
    assert np.allclose([r[1] for r in res], c["conf"], atol=1e-7)
