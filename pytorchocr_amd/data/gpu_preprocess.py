"""GPU pre-process wrappers: the u8 image goes to the device once (4x less PCIe traffic than the
normalised fp32 tensor) and resize + normalise + layout, the per-box perspective crops and the recognition crops'
resize/pad all run there, batched through descriptor arrays."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..utils.warp import get_perspective_transform


class PreItem(C.Structure):
    _fields_ = [("src_off", C.c_long), ("sh", C.c_int), ("sw", C.c_int), ("rh", C.c_int), ("rw", C.c_int),
                ("dst_off", C.c_long), ("dh", C.c_int), ("dw", C.c_int)]


class WarpItem(C.Structure):
    _fields_ = [("minv", C.c_double * 9), ("left", C.c_int), ("top", C.c_int), ("cw", C.c_int), ("ch", C.c_int),
                ("rot90", C.c_int), ("dst_off", C.c_long)]


def _to_dev(arr, device):
    buf = bytes(arr) if not isinstance(arr, (bytes, bytearray)) else arr
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(device)


def _u8_dev(img, device):
    if isinstance(img, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(img)).to(device)
    return img.contiguous()


def det_preprocess(img_bgr, target_hw, mean, std, device, swap_rb=True):
    """u8 HxWx3 BGR (ndarray or device tensor) -> f32[1, rh, rw, 4] NHWC4 network input (channel 3 zero)."""
    src = _u8_dev(img_bgr, device)
    sh, sw = int(src.shape[0]), int(src.shape[1])
    rh, rw = int(target_hw[0]), int(target_hw[1])
    out = torch.empty((1, rh, rw, 4), dtype=torch.float32, device=device)
    items = (PreItem * 1)(PreItem(0, sh, sw, rh, rw, 0, rh, rw))
    d_items = _to_dev(items, device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.lib().ptocr_preprocess_u8_f32(_lib.ptr(src), _lib.ptr(out), _lib.ptr(d_items), 1, rh * rw, 0, int(swap_rb), 4, m, s,
                                                  _lib.cur_stream()), "ptocr_preprocess_u8_f32")
    return out


def warp_crops(img_dev, boxes):
    """img_dev: u8[H,W,3] device tensor; boxes: list of int (4,2) arrays -> (packed u8 device buffer, [(off, h, w)] per crop)
    with get_part_img + the h >= 1.5 w rotation of run_ocr applied."""
    H, W = int(img_dev.shape[0]), int(img_dev.shape[1])
    items, metas, off, maxpix = [], [], 0, 1
    for box in boxes:
        pts = np.asarray(box).astype(np.float32)
        left, right = int(np.min(pts[:, 0])), int(np.max(pts[:, 0]))
        top, bottom = int(np.min(pts[:, 1])), int(np.max(pts[:, 1]))
        left, top = max(left, 0), max(top, 0)
        right, bottom = min(right, W), min(bottom, H)
        cw, ch = right - left, bottom - top
        if cw <= 1 or ch <= 1:
            metas.append(None)
            continue
        p = pts - np.array([left, top], np.float32)
        dst = np.array([[0, 0], [cw - 1, 0], [cw - 1, ch - 1], [0, ch - 1]], np.float32)
        minv = np.linalg.inv(get_perspective_transform(p, dst))
        rot = 1 if ch >= 1.5 * cw else 0
        it = WarpItem((C.c_double * 9)(*minv.reshape(-1).tolist()), left, top, cw, ch, rot, off)
        items.append(it)
        metas.append((off, cw, ch) if rot else (off, ch, cw))          # (offset, rows, cols) of the stored crop
        off += cw * ch * 3
        maxpix = max(maxpix, cw * ch)
    buf = torch.empty(max(off, 1), dtype=torch.uint8, device=img_dev.device)
    if items:
        arr = (WarpItem * len(items))(*items)
        d_items = _to_dev(arr, img_dev.device)
        _lib.check(_lib.lib().ptocr_warp_crops_u8(_lib.ptr(img_dev), H, W, _lib.ptr(buf), _lib.ptr(d_items), len(items), maxpix,
                                                  _lib.cur_stream()), "ptocr_warp_crops_u8")
    return buf, metas


def rec_preprocess(buf, metas, image_shape, device):
    """packed BGR u8 crops -> f32[n, imgH, imgW, 4] (gray in channel 0, (x/255-0.5)/0.5, right zero padding)."""
    imgC, imgH, imgW = image_shape
    assert imgC == 1, "the GPU recognition pre-process implements the GRAY (1-channel) CRNN input"
    valid = [m for m in metas if m is not None]
    n = len(valid)
    out = torch.empty((max(n, 1), imgH, imgW, 4), dtype=torch.float32, device=device)
    if n == 0:
        return out[:0]
    items = []
    for i, (off, h, w) in enumerate(valid):
        ratio = w / float(h)
        rw = imgW if math.ceil(imgH * ratio) > imgW else int(math.ceil(imgH * ratio))
        items.append(PreItem(off, h, w, imgH, max(rw, 1), i * imgH * imgW * 4, imgH, imgW))
    arr = (PreItem * n)(*items)
    d_items = _to_dev(arr, device)
    _lib.check(_lib.lib().ptocr_preprocess_u8_f32(_lib.ptr(buf), _lib.ptr(out), _lib.ptr(d_items), n, imgH * imgW, 1, 0, 4, None, None,
                                                  _lib.cur_stream()), "ptocr_preprocess_u8_f32")
    return out
