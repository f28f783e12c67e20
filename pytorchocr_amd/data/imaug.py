"""Inference pre-process operators (host side, numpy).  Mirrors of reference pytocr/data/imaug/operators.py
(`DecodeImage` :14-38, `ToTensor` :41-72, `Normalize` :75-112, `KeepKeys` :115-124, `DetResizeForTest` :155-252) and
rec_img_aug.py (`ClsResizeImg` :29-37, `RecResizeImg` :40-53, `resize_norm_img` :108-134).

cv2 is not a dependency here: `resize_bilinear` / `bgr_to_gray` restate cv2.resize(INTER_LINEAR) and cv2.cvtColor(BGR2GRAY) for uint8
images in OpenCV's own fixed-point arithmetic (resize.cpp: 11-bit coefficient tables, `HResizeLinear`, the truncating u8
`VResizeLinear`, the 2x2 area re-route; color_rgb: 14-bit B2Y / G2Y / R2Y of the pinned opencv-python 4.1.2.30).  Held bit-exact to
oracle/cv2_oracle.py by tests/test_oracle_cv2.py; UNPINNED against OpenCV itself (absent from the image).
"""
import math

import numpy as np
import torch


def _resize_tables(dn, sn, clamp):
    """resize.cpp: per destination index the source offset and the two 11-bit coefficients; x offsets are clamped with the fraction
    zeroed, y offsets are not (the rows are clipped when they are fetched)"""
    scale = 1.0 / (float(dn) / float(sn))                                                # hal::resize: scale = 1. / inv_scale
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)       # fx = (float)((dx + 0.5) * scale_x - 0.5)
    s0 = np.floor(f).astype(np.int64)
    fr = (f - s0.astype(np.float32)).astype(np.float32)
    if clamp:
        lo = s0 < 0
        fr[lo] = 0
        s0[lo] = 0
        hi = s0 >= sn - 1
        fr[hi] = 0
        s0[hi] = sn - 1
    return s0, fr


def resize_bilinear(img, dsize):
    """cv2.resize(img, (w, h)) with INTER_LINEAR for uint8 (H,W[,C]) images; float images are interpolated in float."""
    dw, dh = int(dsize[0]), int(dsize[1])
    src = np.asarray(img)
    sh, sw = src.shape[:2]
    if (sh, sw) == (dh, dw):
        return src.copy()
    x0, fx = _resize_tables(dw, sw, True)
    x1 = np.minimum(x0 + 1, sw - 1)
    ys, fy = _resize_tables(dh, sh, False)
    y0, y1 = np.clip(ys, 0, sh - 1), np.clip(ys + 1, 0, sh - 1)
    if src.dtype == np.uint8:
        s = src.astype(np.int64)
        if sw == 2 * dw and sh == 2 * dh:              # exact 2x2 down-scale: OpenCV runs INTER_LINEAR as INTER_AREA (fast)
            return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        ONE = np.float32(2048)
        ax0 = np.clip(np.rint((np.float32(1) - fx) * ONE), -32768, 32767).astype(np.int64)   # saturate_cast<short>((1.f - fx) * 2048)
        ax1 = np.clip(np.rint(fx * ONE), -32768, 32767).astype(np.int64)
        by0 = np.clip(np.rint((np.float32(1) - fy) * ONE), -32768, 32767).astype(np.int64)
        by1 = np.clip(np.rint(fy * ONE), -32768, 32767).astype(np.int64)
        shp = (1, dw) + (1,) * (s.ndim - 2)
        rows = s[:, x0] * ax0.reshape(shp) + s[:, x1] * ax1.reshape(shp)                # HResizeLinear, scale 2^11
        shp = (dh, 1) + (1,) * (s.ndim - 2)
        # VResizeLinear<uchar, ...>: two truncating 16-bit products, then (+ 2) >> 2
        out = (((by0.reshape(shp) * (rows[y0] >> 4)) >> 16) + ((by1.reshape(shp) * (rows[y1] >> 4)) >> 16) + 2) >> 2
        return (out & 255).astype(np.uint8)
    s = src.astype(np.float32)
    shp = (1, dw) + (1,) * (s.ndim - 2)
    rows = s[:, x0] * (1 - fx).reshape(shp) + s[:, x1] * fx.reshape(shp)
    shp = (dh, 1) + (1,) * (s.ndim - 2)
    return (rows[y0] * (1 - fy).reshape(shp) + rows[y1] * fy.reshape(shp)).astype(src.dtype)


def bgr_to_gray(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for uint8: CV_DESCALE(B * 1868 + G * 9617 + R * 4899, 14) (opencv-python 4.1.2.30)."""
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


class DecodeImage(object):
    def __init__(self, img_mode="RGB", channel_first=False, **kwargs):
        self.img_mode = img_mode
        self.channel_first = channel_first

    def __call__(self, data):
        from PIL import Image
        import io
        img = data["image"]
        if isinstance(img, (bytes, bytearray)):
            img = np.array(Image.open(io.BytesIO(img)).convert("RGB"))[:, :, ::-1]      # BGR like cv2.imdecode
        if self.img_mode == "GRAY":
            img = bgr_to_gray(img)
        elif self.img_mode == "RGB":
            img = img[:, :, ::-1]
        if self.channel_first:
            img = img.transpose((2, 0, 1))
        data["image"] = np.ascontiguousarray(img)
        return data


class ToTensor(object):
    """torchvision to_tensor: uint8 HWC -> float32 CHW / 255"""

    def __init__(self, **kwargs):
        pass

    def __call__(self, data):
        img = data["image"]
        if img.ndim == 2:
            img = img[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(img.transpose((2, 0, 1))))
        data["image"] = t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t
        return data


class Normalize(object):
    def __init__(self, mean, std, inplace=False, **kwargs):
        self.mean, self.std, self.inplace = mean, std, inplace

    def __call__(self, data):
        img = data["image"]
        mean = torch.as_tensor(self.mean, dtype=img.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=img.dtype).view(-1, 1, 1)
        data["image"] = (img - mean) / std
        return data


class KeepKeys(object):
    def __init__(self, keep_keys, **kwargs):
        self.keep_keys = keep_keys

    def __call__(self, data):          # returns a list, not a dict (reference operators.py:119-124)
        return [data[key] for key in self.keep_keys]


class DetResizeForTest(object):
    def __init__(self, **kwargs):
        self.resize_type = 0
        if "image_shape" in kwargs:
            self.image_shape = kwargs["image_shape"]
            self.resize_type = 1
        elif "limit_side_len" in kwargs:
            self.limit_side_len = kwargs["limit_side_len"]
            self.limit_type = kwargs.get("limit_type", "min")
        elif "resize_long" in kwargs:
            self.resize_type = 2
            self.resize_long = kwargs.get("resize_long", 960)
        else:
            self.limit_side_len = 736
            self.limit_type = "min"

    def target_size(self, h, w):
        """(resize_h, resize_w) the reference computes for an h x w image"""
        if self.resize_type == 1:
            return int(self.image_shape[0]), int(self.image_shape[1])
        if self.resize_type == 2:
            ratio = float(self.resize_long) / max(h, w)
            rh, rw = int(h * ratio), int(w * ratio)
            return (rh + 127) // 128 * 128, (rw + 127) // 128 * 128
        if self.limit_type == "max":
            ratio = float(self.limit_side_len) / (h if h > w else w)
        elif self.limit_type == "min":
            ratio = float(self.limit_side_len) / (h if h < w else w)
        elif self.limit_type == "resize_long":
            ratio = float(self.limit_side_len) / max(h, w)
        else:
            raise Exception("not support limit type, image ")
        rh, rw = int(h * ratio), int(w * ratio)
        return max(int(round(rh / 32) * 32), 32), max(int(round(rw / 32) * 32), 32)

    def __call__(self, data):
        img = data["image"]
        src_h, src_w, _ = img.shape
        rh, rw = self.target_size(src_h, src_w)
        data["image"] = resize_bilinear(img, (rw, rh))
        data["shape"] = np.array([src_h, src_w, rh / float(src_h), rw / float(src_w)])
        return data


def resize_norm_img(img, image_shape, resized_w=None, padding=True):
    imgC, imgH, imgW = image_shape
    h, w = img.shape[:2]
    if not padding:
        resized_image = resize_bilinear(img, (imgW, imgH))
        resized_w = imgW
    elif resized_w is not None:
        resized_image = resize_bilinear(img, (resized_w, imgH))
    else:
        ratio = w / float(h)
        resized_w = imgW if math.ceil(imgH * ratio) > imgW else int(math.ceil(imgH * ratio))
        resized_image = resize_bilinear(img, (resized_w, imgH))
    resized_image = resized_image.astype("float32")
    if image_shape[0] == 1 and len(img.shape) == 2:
        resized_image = resized_image / 255
        resized_image = resized_image[np.newaxis, :]
    else:
        resized_image = resized_image.transpose((2, 0, 1)) / 255
    resized_image -= 0.5
    resized_image /= 0.5
    padding_im = np.zeros((imgC, imgH, imgW), dtype=np.float32)
    padding_im[:, :, 0:resized_w] = resized_image
    return torch.from_numpy(padding_im)


class RecResizeImg(object):
    def __init__(self, image_shape, padding=True, **kwargs):
        self.image_shape = image_shape
        self.padding = padding

    def __call__(self, data):
        data["image"] = resize_norm_img(data["image"], self.image_shape, resized_w=None, padding=self.padding)
        return data


class ClsResizeImg(object):
    """reference rec_img_aug.py:29-37: the recognition resize (aspect kept, right zero padding) at the classifier's shape"""

    def __init__(self, image_shape, **kwargs):
        self.image_shape = image_shape

    def __call__(self, data):
        data["image"] = resize_norm_img(data["image"], self.image_shape)
        return data


class RecResizeImgForTest(object):
    """Variable-width recognition batching (reference rec_img_aug.py:55-106): every crop is resized to height imgH keeping its
    aspect (width = ceil(w * imgH / h), capped at max_w); a LIST of crops is cut into batches of batch_size in input order, each
    batch zero-padded on the right to ITS widest crop -> list of tensors f32[b, imgC, imgH, batch_max_w]; a single crop ->
    f32[1, imgC, imgH, its own width]."""

    def __init__(self, imgC=1, imgH=32, max_w=1200, batch_size=16, padding=True, **kwargs):
        self.imgC, self.imgH, self.max_w, self.batch_size, self.padding = imgC, imgH, max_w, batch_size, padding

    def width_of(self, img):
        h, w = img.shape[:2]
        return min(int(math.ceil(w * (self.imgH / float(h)))), self.max_w)

    def __call__(self, imgs):
        if not isinstance(imgs, list):
            w = self.width_of(imgs)
            return resize_norm_img(imgs, [self.imgC, self.imgH, w], resized_w=w, padding=self.padding).unsqueeze(dim=0)
        widths = [self.width_of(i) for i in imgs]
        out = []
        for b0 in range(0, len(imgs), self.batch_size):
            ws = widths[b0:b0 + self.batch_size]
            shape = [self.imgC, self.imgH, max(ws)]
            out.append(torch.stack([resize_norm_img(i, shape, resized_w=w, padding=self.padding)
                                    for i, w in zip(imgs[b0:b0 + self.batch_size], ws)], dim=0))
        return out
