"""GPU pre-process wrappers: the u8 image goes to the device once (4x less PCIe traffic than the
normalised fp32 tensor) and resize + normalise + layout, the per-box perspective crops and the recognition crops'
resize/pad all run there, batched through descriptor arrays."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib


class PreItem(C.Structure):
    _fields_ = [("src_off", C.c_long), ("sh", C.c_int), ("sw", C.c_int), ("rh", C.c_int), ("rw", C.c_int),
                ("dst_off", C.c_long), ("dh", C.c_int), ("dw", C.c_int), ("flip", C.c_int), ("pad_", C.c_int)]


class WarpItem(C.Structure):
    _fields_ = [("minv", C.c_double * 9), ("left", C.c_int), ("top", C.c_int), ("cw", C.c_int), ("ch", C.c_int),
                ("rot90", C.c_int), ("img", C.c_int), ("dst_off", C.c_long)]


def _to_dev(arr, device):
    buf = bytes(arr) if not isinstance(arr, (bytes, bytearray)) else arr
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(device)


def _u8_dev(img, device):
    if isinstance(img, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(img)).to(device)
    return img.contiguous()


PRE_DT = np.dtype([("src_off", "<i8"), ("sh", "<i4"), ("sw", "<i4"), ("rh", "<i4"), ("rw", "<i4"), ("dst_off", "<i8"),
                   ("dh", "<i4"), ("dw", "<i4"), ("flip", "<i4"), ("pad_", "<i4")], align=True)
WARP_DT = np.dtype([("minv", "<f8", (9,)), ("left", "<i4"), ("top", "<i4"), ("cw", "<i4"), ("ch", "<i4"), ("rot90", "<i4"),
                    ("img", "<i4"), ("dst_off", "<i8")], align=True)
assert PRE_DT.itemsize == C.sizeof(PreItem) and WARP_DT.itemsize == C.sizeof(WarpItem)


def _items_dev(arr, device):
    """numpy structured descriptor array -> device bytes"""
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(device)


def det_preprocess(img_bgr, target_hw, mean, std, device, swap_rb=True):
    """u8 HxWx3 BGR (ndarray or device tensor) -> f32[1, rh, rw, 4] NHWC4 network input (channel 3 zero)."""
    src = _u8_dev(img_bgr, device)
    return det_preprocess_batch(src[None], target_hw, mean, std, swap_rb)


def det_preprocess_batch(imgs_dev, target_hw, mean, std, swap_rb=True):
    """u8[N,H,W,3] BGR device tensor (equally sized images) -> f32[N, rh, rw, 4]: resize + normalise + layout of the whole
    batch in ONE launch (DetResizeForTest + ToTensor + Normalize, operators.py:41-112,155-252)."""
    imgs_dev = imgs_dev.contiguous()
    n, sh, sw = int(imgs_dev.shape[0]), int(imgs_dev.shape[1]), int(imgs_dev.shape[2])
    rh, rw = int(target_hw[0]), int(target_hw[1])
    out = torch.empty((n, rh, rw, 4), dtype=torch.float32, device=imgs_dev.device)
    items = np.zeros(n, PRE_DT)
    items["src_off"] = np.arange(n, dtype=np.int64) * (sh * sw * 3)
    items["sh"], items["sw"], items["rh"], items["rw"], items["dh"], items["dw"] = sh, sw, rh, rw, rh, rw
    items["dst_off"] = np.arange(n, dtype=np.int64) * (rh * rw * 4)
    d_items = _items_dev(items, imgs_dev.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.lib().ptocr_preprocess_u8_f32(_lib.ptr(imgs_dev), _lib.ptr(out), _lib.ptr(d_items), n, rh * rw, 0, int(swap_rb), 4, m, s,
                                                  _lib.cur_stream()), "ptocr_preprocess_u8_f32")
    return out


def warp_crops(img_dev, boxes):
    """img_dev: u8[H,W,3] device tensor; boxes: list of int (4,2) arrays -> (packed u8 device buffer, [(off, h, w)] per crop)
    with get_part_img + the h >= 1.5 w rotation of run_ocr applied."""
    buf, metas = warp_crops_batch(img_dev[None], [boxes])
    return buf, metas[0]


def crop_rects(pts, H, W):
    """get_part_img's crop rectangle (reference utility.py:57-66) for float32 boxes [K,4,2] of an H x W image: (left, top, width,
    height) per box, truncating like int(); the slice img[top:bottom, left:right] of the reference clamps at the image edge."""
    left = np.maximum(pts[:, :, 0].min(1).astype(np.int64), 0)
    top = np.maximum(pts[:, :, 1].min(1).astype(np.int64), 0)
    right = np.minimum(pts[:, :, 0].max(1).astype(np.int64), W)
    bottom = np.minimum(pts[:, :, 1].max(1).astype(np.int64), H)
    return left, top, right - left, bottom - top


def rot90_rule(ch, cw):
    """run_ocr.py:189-190: a crop at least 1.5 times as high as wide is turned counter-clockwise"""
    return ch >= 1.5 * cw


def warp_crops_batch(imgs_dev, boxes_per_image):
    """imgs_dev: u8[N,H,W,3] device tensor; boxes_per_image: N lists of int (4,2) boxes -> (packed u8 device buffer, per image a
    list of (off, rows, cols) / None per box): the perspective crops of ALL boxes of ALL images in one launch.  The 8x8 solves
    of cv2.getPerspectiveTransform and the 3x3 inverses run batched on the host (LAPACK, one call each)."""
    from ..utils.warp import get_perspective_transforms, invert_transforms
    H, W = int(imgs_dev.shape[1]), int(imgs_dev.shape[2])
    counts = [len(b) for b in boxes_per_image]
    total = int(sum(counts))
    metas = [[None] * c for c in counts]
    if total == 0:
        return torch.empty(1, dtype=torch.uint8, device=imgs_dev.device), metas
    pts = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4, 2) for b in boxes_per_image if len(b)]).astype(np.float32)
    img_of = np.repeat(np.arange(len(counts)), counts)
    left, top, cw, ch = crop_rects(pts, H, W)
    ok = (cw > 1) & (ch > 1)
    if not ok.any():
        return torch.empty(1, dtype=torch.uint8, device=imgs_dev.device), metas
    sel = np.nonzero(ok)[0]
    p = pts[sel] - np.stack([left[sel], top[sel]], 1).astype(np.float32)[:, None, :]
    cwf, chf = cw[sel].astype(np.float32), ch[sel].astype(np.float32)
    z = np.zeros_like(cwf)
    dst = np.stack([np.stack([z, z], 1), np.stack([cwf - 1, z], 1), np.stack([cwf - 1, chf - 1], 1), np.stack([z, chf - 1], 1)], 1)
    minv = invert_transforms(get_perspective_transforms(p, dst))
    rot = rot90_rule(ch[sel], cw[sel]).astype(np.int32)
    sizes = cw[sel] * ch[sel] * 3
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    items = np.zeros(len(sel), WARP_DT)
    items["minv"] = minv.reshape(-1, 9)
    items["left"], items["top"], items["cw"], items["ch"] = left[sel], top[sel], cw[sel], ch[sel]
    items["rot90"], items["img"], items["dst_off"] = rot, img_of[sel], offs
    # (off, rows, cols) per kept box -- a rotated crop swaps rows and columns -- built column-wise (a Python loop over 11 000 boxes with
    # numpy scalar conversions cost 10 ms per batch of 64 images), scattered into the flat box order, then cut per image
    rows = np.where(rot != 0, cw[sel], ch[sel])
    cols = np.where(rot != 0, ch[sel], cw[sel])
    flat = [None] * total
    for g, t in zip(sel.tolist(), zip(offs.tolist(), rows.tolist(), cols.tolist())):
        flat[g] = t
    first = np.concatenate([[0], np.cumsum(counts)]).tolist()
    metas = [flat[first[i]:first[i + 1]] for i in range(len(counts))]
    buf = torch.empty(int(sizes.sum()), dtype=torch.uint8, device=imgs_dev.device)
    d_items = _items_dev(items, imgs_dev.device)
    _lib.check(_lib.lib().ptocr_warp_crops_u8(_lib.ptr(imgs_dev.contiguous()), H, W, _lib.ptr(buf), _lib.ptr(d_items), len(sel),
                                              int((cw[sel] * ch[sel]).max()), _lib.cur_stream()), "ptocr_warp_crops_u8")
    return buf, metas


def cls_preprocess(buf, metas, image_shape, device, swap_rb=True):
    """packed BGR u8 crops -> f32[n, imgH, imgW, 4] direction-classifier input: ClsResizeImg (rec_img_aug.py:29-37,108-134) --
    aspect-keeping resize to height imgH, 3 channels (RGB when swap_rb), (x/255-0.5)/0.5, right zero padding."""
    return rec_preprocess(buf, metas, image_shape, device, _mode3=(1 if swap_rb else 0))


def rec_preprocess(buf, metas, image_shape, device, flip=None, _mode3=None):
    """packed BGR u8 crops -> f32[n, imgH, imgW, 4] (gray in channel 0, (x/255-0.5)/0.5, right zero padding).
    flip: optional bool per valid crop -- read that crop rotated by 180 degrees (the classifier said "180")."""
    imgC, imgH, imgW = image_shape
    if _mode3 is None:
        assert imgC == 1, "the GPU recognition pre-process implements the GRAY (1-channel) CRNN input"
    else:
        assert imgC == 3, "the GPU classifier pre-process implements the 3-channel input"
    valid = [m for m in metas if m is not None]
    n = len(valid)
    out = torch.empty((max(n, 1), imgH, imgW, 4), dtype=torch.float32, device=device)
    if n == 0:
        return out[:0]
    v = np.asarray(valid, np.int64)                                   # (off, rows, cols)
    ratio = v[:, 2] / v[:, 1].astype(np.float64)
    need = np.ceil(imgH * ratio)                                      # math.ceil(imgH * ratio), rec_img_aug.py:119-123
    rw = np.where(need > imgW, imgW, need).astype(np.int64)
    items = np.zeros(n, PRE_DT)
    items["src_off"], items["sh"], items["sw"] = v[:, 0], v[:, 1], v[:, 2]
    items["rh"], items["rw"] = imgH, np.maximum(rw, 1)
    items["dst_off"] = np.arange(n, dtype=np.int64) * (imgH * imgW * 4)
    items["dh"], items["dw"] = imgH, imgW
    if flip is not None:
        items["flip"] = np.asarray(flip, np.int32)
    d_items = _items_dev(items, device)
    if _mode3 is None:
        _lib.check(_lib.lib().ptocr_preprocess_u8_f32(_lib.ptr(buf), _lib.ptr(out), _lib.ptr(d_items), n, imgH * imgW, 1, 0, 4, None, None,
                                                      _lib.cur_stream()), "ptocr_preprocess_u8_f32")
    else:
        half = (C.c_float * 3)(0.5, 0.5, 0.5)
        _lib.check(_lib.lib().ptocr_preprocess_u8_f32(_lib.ptr(buf), _lib.ptr(out), _lib.ptr(d_items), n, imgH * imgW, 0, int(_mode3), 4, half,
                                                      half, _lib.cur_stream()), "ptocr_preprocess_u8_f32")
    return out
