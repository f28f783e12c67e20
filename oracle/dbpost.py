"""ORACLE (test infrastructure only): ctypes front-end of oracle/libdbpost_oracle.so (+ oracle/_ref).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


class Dbg(C.Structure):
    _fields_ = [("status", C.c_int), ("is_hole", C.c_int), ("trig_x", C.c_int), ("trig_y", C.c_int),
                ("start_x", C.c_int), ("start_y", C.c_int), ("npts", C.c_int), ("npix", C.c_int),
                ("npoly", C.c_int), ("rect", C.c_float * 5), ("minibox", C.c_float * 8),
                ("score", C.c_float), ("distance", C.c_float), ("urect", C.c_float * 5), ("box", C.c_int * 8)]


_lib = None
_ref = None


def build(ref=True):
    """(Re)build the C restatement; and oracle/_ref when the reference tree is present."""
    subprocess.check_call(["make", "-s", "-C", _DIR, "all"])
    if ref and os.path.exists("/root/reference/pytocr/postprocess/db_postprocess_fast/src/clipper.cpp"):
        subprocess.check_call(["make", "-s", "-C", _DIR, "ref"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_DIR, "libdbpost_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.dbpost_oracle_run.restype = C.c_int
        L.dbpost_oracle_contours.restype = C.c_int
        L.dbpost_oracle_clipper_offset.restype = C.c_int
        _lib = L
    return _lib


def ref_lib():
    """The reference's vendored Clipper (oracle/_ref/libclipper_ref.so) or None."""
    global _ref
    if _ref is None:
        path = os.path.join(_DIR, "_ref", "libclipper_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.clipper_ref_offset.restype = C.c_int
    return _ref


def use_reference_clipper(on=True):
    """Route the oracle's unclip step through the reference's vendored Clipper (oracle/_ref) when present.
    Returns True when the real Clipper is in use."""
    L = lib()
    R = ref_lib() if on else None
    if R is None:
        L.dbpost_oracle_set_clipper_ref(C.c_void_p(0))
        return False
    L.dbpost_oracle_set_clipper_ref(C.cast(R.clipper_ref_offset, C.c_void_p))
    return True


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def binarize(pred, thresh):
    pred = np.ascontiguousarray(pred, np.float32)
    out = np.empty(pred.shape, np.uint8)
    lib().dbpost_oracle_binarize(_p(pred, C.c_float), C.c_size_t(pred.size), C.c_float(thresh), _p(out, C.c_uint8))
    return out


def dilate2x2(bitmap):
    bitmap = np.ascontiguousarray(bitmap, np.uint8)
    out = np.empty_like(bitmap)
    lib().dbpost_oracle_dilate2x2(_p(bitmap, C.c_uint8), bitmap.shape[0], bitmap.shape[1], _p(out, C.c_uint8))
    return out


def boxes_from_bitmap(pred, bitmap, box_thresh, unclip_ratio, src_w, src_h, with_debug=False, use_padding_resize=False):
    """Same contract as the reference pybind `db_postprocess` (use_padding_resize=False):
    pred f32[H,W], bitmap u8[H,W] -> int[K,4,2] (+ per-contour debug records)."""
    pred = np.ascontiguousarray(pred, np.float32)
    bitmap = np.ascontiguousarray(bitmap, np.uint8)
    H, W = pred.shape
    boxes = np.zeros((1000, 8), np.int32)
    dbg = (Dbg * 1000)()
    ncont = C.c_int(0)
    lib().dbpost_oracle_set_padding_resize(int(bool(use_padding_resize)))
    k = lib().dbpost_oracle_run(_p(pred, C.c_float), _p(bitmap, C.c_uint8), H, W, C.c_float(box_thresh),
                                C.c_float(unclip_ratio), int(src_w), int(src_h), _p(boxes, C.c_int), 1000,
                                dbg, 1000, C.byref(ncont))
    lib().dbpost_oracle_set_padding_resize(0)
    out = boxes[:k].reshape(k, 4, 2).copy()
    if with_debug:
        return out, [dbg[i] for i in range(min(ncont.value, 1000))], ncont.value
    return out


def contours(bitmap, cap=200000, pts_cap=4000000):
    bitmap = np.ascontiguousarray(bitmap, np.uint8)
    H, W = bitmap.shape
    npts = np.zeros(cap, np.int32); hole = np.zeros(cap, np.int32); trig = np.zeros((cap, 2), np.int32)
    pts = np.zeros((pts_cap, 2), np.int32)
    n = lib().dbpost_oracle_contours(_p(bitmap, C.c_uint8), H, W, _p(npts, C.c_int), _p(hole, C.c_int),
                                     _p(trig, C.c_int), cap, _p(pts, C.c_int), pts_cap)
    assert n <= cap
    off = np.concatenate([[0], np.cumsum(npts[:n])])
    assert off[-1] <= pts_cap
    return [pts[off[i]:off[i + 1]].copy() for i in range(n)], hole[:n].copy(), trig[:n].copy()


def min_area_rect(pts):
    pts = np.ascontiguousarray(pts, np.float32)
    rect = np.zeros(5, np.float32); mb = np.zeros(8, np.float32); ssid = C.c_float(0)
    lib().dbpost_oracle_min_area_rect(_p(pts, C.c_float), len(pts), _p(rect, C.c_float), _p(mb, C.c_float), C.byref(ssid))
    return rect, mb.reshape(4, 2), ssid.value


def clipper_offset(path, delta):
    path = np.ascontiguousarray(path, np.int64)
    out = np.zeros((1024, 2), np.int64)
    n = lib().dbpost_oracle_clipper_offset(_p(path, C.c_longlong), len(path), C.c_double(delta), _p(out, C.c_longlong), 1024)
    return out[:n].copy()


def clipper_unclip(path, delta):
    """restated offset + restated union: the solution's vertices as far as their hull goes (int64 [n,2]; n = 0: no solution)"""
    path = np.ascontiguousarray(path, np.int64).reshape(-1, 2)
    out = np.zeros((1024, 2), np.int64)
    L = lib()
    L.dbpost_oracle_clipper_unclip.restype = C.c_int
    n = L.dbpost_oracle_clipper_unclip(_p(path, C.c_longlong), len(path), C.c_double(delta), _p(out, C.c_longlong), 1024)
    return out[:max(n, 0)].copy()


def clipper_ref_offset(path, delta):
    """Real Clipper: returns list of paths."""
    L = ref_lib()
    if L is None:
        return None
    path = np.ascontiguousarray(path, np.int64)
    out = np.zeros((4096, 2), np.int64); sizes = np.zeros(64, np.int32)
    r = L.clipper_ref_offset(_p(path, C.c_longlong), len(path), C.c_double(delta), _p(out, C.c_longlong), 4096, _p(sizes, C.c_int), 64)
    npaths, n = divmod(r, 100000)
    res, off = [], 0
    for j in range(npaths):
        res.append(out[off:off + sizes[j]].copy()); off += sizes[j]
    return res
