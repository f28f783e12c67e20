"""Host logic held to fixtures RECORDED FROM THE REFERENCE's own functions (tools/gen_golden.py --shapes-only: the reference's
utility.py / operators.py / rec_img_aug.py loaded by file path with a `cv2` stub that returns zeros of the requested size and records
its arguments): sort_boxes (utility.py:32-50), DetResizeForTest's size / ratio logic (operators.py:155-275), resize_norm_img's width
logic and RecResizeImgForTest's batching (rec_img_aug.py:55-134), get_part_img's crop geometry (utility.py:53-78) and, on the sizes the
reference's get_part_img returned, the rot90 rule of run_ocr.py:189-190.  No GPU."""
import json
import os
import warnings

import numpy as np
import pytest

from pytorchocr_amd.data import imaug
from pytorchocr_amd.data import gpu_preprocess
from pytorchocr_amd.utils import warp
from pytorchocr_amd.utils.utility import sort_boxes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def shapes():
    with open(os.path.join(GOLD, "host_shapes.json")) as f:
        return json.load(f)


@pytest.fixture()
def resize_stub(monkeypatch):
    """resize_bilinear -> zeros of the requested size; the requested sizes are recorded (the pixel arithmetic has its own tests)"""
    calls = []

    def fake(img, dsize):
        calls.append((int(dsize[0]), int(dsize[1])))
        return np.zeros((dsize[1], dsize[0]) + tuple(img.shape[2:]), img.dtype)

    monkeypatch.setattr(imaug, "resize_bilinear", fake)
    return calls


def test_sort_boxes_matches_the_reference_on_200_recorded_sets():
    z = np.load(os.path.join(GOLD, "sort_boxes.npz"))
    counts, bin_, bout = z["sort_counts"], z["sort_in"], z["sort_out"]
    assert len(counts) == 200 and (counts == 0).any() and (bin_[:, 0, 1] == -32768).any()
    off = 0
    moved = 0
    for k in counts.tolist():
        b, exp = bin_[off:off + k], bout[off:off + k]
        off += k
        for arr in (b, b.astype(np.int64).astype(np.int16)):               # the int16 fast path
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = sort_boxes(arr if k else np.zeros((0,), np.float32))
            assert len(got) == k
            if k:
                assert np.array_equal(np.array(got).reshape(-1, 4, 2), exp)
        moved += int(k and not np.array_equal(b, exp))
    assert moved > 150                                                      # the fixture is not made of sorted inputs


class _Listlike(list):
    shape = property(lambda self: (len(self),))


def test_sort_boxes_generic_branch_matches_too():
    z = np.load(os.path.join(GOLD, "sort_boxes.npz"))
    counts, bin_, bout = z["sort_counts"], z["sort_in"], z["sort_out"]
    off = 0
    for k in counts.tolist():
        b, exp = bin_[off:off + k], bout[off:off + k]
        off += k
        if not k:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = sort_boxes(_Listlike(b))                                  # forces the sorted() + one-pass form
        assert np.array_equal(np.array(got).reshape(-1, 4, 2), exp)


def test_det_resize_for_test_sizes_and_ratios(shapes, resize_stub):
    cases = shapes["det_resize"]
    assert len(cases) >= 300
    kinds = set()
    for c in cases:
        del resize_stub[:]
        op = imaug.DetResizeForTest(**c["kw"])
        d = op({"image": np.zeros((c["h"], c["w"], 3), np.uint8)})
        assert resize_stub == [(c["resize_w"], c["resize_h"])], c
        assert op.target_size(c["h"], c["w"]) == (c["resize_h"], c["resize_w"])
        assert d["shape"].tolist() == c["shape"], c                         # src_h, src_w, ratio_h, ratio_w: the same doubles
        kinds.add(json.dumps(c["kw"], sort_keys=True))
    assert len(kinds) == 7
    # the fixture holds half-way cases of round(x / 32) (Python rounds half to even)
    assert any((int(c["h"] * 736.0 / min(c["h"], c["w"])) % 32) == 16 or (int(c["w"] * 736.0 / min(c["h"], c["w"])) % 32) == 16
               for c in cases if c["kw"].get("limit_type") == "min" and c["kw"].get("limit_side_len") == 736)


def test_resize_norm_img_width_logic(shapes, resize_stub):
    for c in shapes["resize_norm_img"]:
        del resize_stub[:]
        img = np.zeros((c["h"], c["w"]) if c["gray"] else (c["h"], c["w"], 3), np.uint8)
        t = imaug.resize_norm_img(img, c["image_shape"], resized_w=c["resized_w_arg"], padding=c["padding"])
        assert resize_stub == [tuple(c["resize_dsize"])], c
        assert list(t.shape) == c["out_shape"], c


def test_rec_resize_for_test_batching(shapes, resize_stub):
    for c in shapes["rec_resize_for_test"]:
        op = imaug.RecResizeImgForTest(**c["kw"])
        del resize_stub[:]
        ts = op([np.zeros(tuple(s), np.uint8) for s in c["hw"]])
        assert [list(t.shape) for t in ts] == c["batch_shapes"], c
        assert [list(d) for d in resize_stub] == c["resize_dsizes"], c
        del resize_stub[:]
        t1 = op(np.zeros(tuple(c["hw"][0]), np.uint8))
        assert list(t1.shape) == c["single_shape"] and [list(d) for d in resize_stub] == [c["single_dsize"]], c


def test_get_part_img_geometry_and_the_rot90_rule(shapes, monkeypatch):
    rec = {}

    def fake_transform(src, dst):
        rec["src"], rec["dst"] = np.asarray(src).copy(), np.asarray(dst).copy()
        return np.eye(3)

    def fake_warp(img, M, dsize):
        rec["dsize"], rec["in_shape"] = (int(dsize[0]), int(dsize[1])), tuple(img.shape)
        return np.zeros((dsize[1], dsize[0]) + tuple(img.shape[2:]), img.dtype)

    monkeypatch.setattr(warp, "get_perspective_transform", fake_transform)
    monkeypatch.setattr(warp, "warp_perspective_replicate", fake_warp)
    img = np.zeros((960, 1280, 3), np.uint8)
    cases = shapes["get_part_img"]
    turned = 0
    for c in cases:
        pts = np.array(c["pts"], np.int16)
        crop = warp.get_part_img(img, pts)
        assert list(crop.shape) == c["out_shape"], c
        assert rec["dsize"] == tuple(c["dsize"]) and list(rec["in_shape"]) == c["crop_in_shape"], c
        assert rec["src"].dtype == np.float32 and np.array_equal(rec["src"], np.array(c["src"], np.float32)), c
        assert np.array_equal(rec["dst"], np.array(c["dst"], np.float32)), c
        # the batched GPU path plans the same rectangle (data/gpu_preprocess.py) ...
        left, top, cw, ch = gpu_preprocess.crop_rects(pts.astype(np.float32)[None], 960, 1280)
        assert (int(cw[0]), int(ch[0])) == tuple(c["dsize"]), c
        assert np.array_equal(pts.astype(np.float32) - np.array([left[0], top[0]], np.float32), np.array(c["src"], np.float32))
        # ... and turns exactly the crops run_ocr.py:189-190 turns, judged on the size the reference's get_part_img returned
        h, w = c["out_shape"][:2]
        assert bool(gpu_preprocess.rot90_rule(ch, cw)[0]) == (h >= 1.5 * w)
        turned += int(h >= 1.5 * w)
    assert 10 < turned < len(cases) - 10
