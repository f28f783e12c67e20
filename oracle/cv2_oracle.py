"""ORACLE (test infrastructure only): literal restatements of the OpenCV calls that sit immediately before the path in the
reference -- cv2.resize(INTER_LINEAR) (pytocr/data/imaug/operators.py:236-250, rec_img_aug.py:112-123), cv2.cvtColor
(COLOR_BGR2GRAY) (deploy/pytorch/infer_rec.py:92-93), cv2.getPerspectiveTransform + cv2.warpPerspective(INTER_LINEAR,
BORDER_REPLICATE) (pytocr/utils/utility.py:53-78).

OpenCV is an un-vendored dependency absent from this image and the reference holds no fixture for these calls: PARITY
UNPINNED.  What pins this file is tests/test_oracle_cv2.py: hand-derivable known answers of OpenCV's published
arithmetic (half-pixel centres, 11-bit fixed-point resize coefficients with round-half-up at bit 22, 15-bit BGR2GRAY weights
4899/9617/1868 x2, 1/32-pixel remap quantisation with 5-bit bilinear weights).  Plain per-pixel Python loops on purpose: slow,
obvious, and written independently of the product's vectorised host operators (pytorchocr_amd/data/imaug.py,
utils/warp.py) and of the HIP kernels (csrc/preprocess.hip), both of which the tests compare against THIS file.

Only tests/ may import this module."""
import math

import numpy as np


def _axis_table(dst_n, src_n):
    """per destination index: (i0, i1, w0, w1) with 11-bit integer weights, cv2's resize.cpp linear table for uint8"""
    scale = src_n / float(dst_n)
    out = []
    for d in range(dst_n):
        f = np.float32((d + 0.5) * scale - 0.5)               # computed in double, stored as float
        s = int(math.floor(f))
        frac = np.float32(f - np.float32(s))
        if s < 0:
            s, frac = 0, np.float32(0)
        if s >= src_n - 1:
            s, frac = src_n - 1, np.float32(0)
        w1 = int(np.rint(np.float32(frac * np.float32(2048))))  # saturate_cast<short>(float): round half to even
        out.append((s, min(s + 1, src_n - 1), 2048 - w1, w1))
    return out


def resize_linear_u8(img, dsize):
    """cv2.resize(img, (w, h)) for uint8 HxW or HxWxC, INTER_LINEAR"""
    dw, dh = int(dsize[0]), int(dsize[1])
    src = np.asarray(img)
    assert src.dtype == np.uint8
    sh, sw = src.shape[:2]
    if (sh, sw) == (dh, dw):
        return src.copy()
    chans = 1 if src.ndim == 2 else src.shape[2]
    s3 = src.reshape(sh, sw, chans).astype(np.int64)
    xt, yt = _axis_table(dw, sw), _axis_table(dh, sh)
    out = np.zeros((dh, dw, chans), np.uint8)
    for y in range(dh):
        y0, y1, b0, b1 = yt[y]
        for x in range(dw):
            x0, x1, a0, a1 = xt[x]
            for c in range(chans):
                top = s3[y0, x0, c] * a0 + s3[y0, x1, c] * a1       # horizontal pass, scale 2^11
                bot = s3[y1, x0, c] * a0 + s3[y1, x1, c] * a1
                v = (top * b0 + bot * b1 + (1 << 21)) >> 22          # vertical pass, round half up
                out[y, x, c] = min(max(int(v), 0), 255)
    return out.reshape((dh, dw) if src.ndim == 2 else (dh, dw, chans))


def bgr2gray_u8(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for uint8: (B*3735 + G*19235 + R*9798 + 2^14) >> 15"""
    h, w = img.shape[:2]
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            b, g, r = (int(v) for v in img[y, x])
            out[y, x] = (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15
    return out


def perspective_matrix(src, dst):
    """cv2.getPerspectiveTransform: the 8x8 system solved in double"""
    a, b = np.zeros((8, 8)), np.zeros(8)
    for i in range(4):
        x, y, u, v = float(src[i][0]), float(src[i][1]), float(dst[i][0]), float(dst[i][1])
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(a, b), 1.0).reshape(3, 3)


def warp_perspective_replicate_u8(img, M, dsize):
    """cv2.warpPerspective(img, M, (w, h), flags=INTER_LINEAR, borderMode=BORDER_REPLICATE) for uint8 HxWxC"""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    inv = np.linalg.inv(np.asarray(M, np.float64))
    out = np.zeros((h, w) + img.shape[2:], np.uint8)
    for y in range(h):
        for x in range(w):
            den = inv[2, 0] * x + inv[2, 1] * y + inv[2, 2]
            den = 1.0 / den if den != 0 else 0.0
            fx = (inv[0, 0] * x + inv[0, 1] * y + inv[0, 2]) * den
            fy = (inv[1, 0] * x + inv[1, 1] * y + inv[1, 2]) * den
            X, Y = int(np.rint(fx * 32)), int(np.rint(fy * 32))      # INTER_TAB_SIZE = 32
            x0, y0 = X >> 5, Y >> 5
            ax, ay = np.float32((X & 31) / 32.0), np.float32((Y & 31) / 32.0)
            cx0, cx1 = min(max(x0, 0), W - 1), min(max(x0 + 1, 0), W - 1)
            cy0, cy1 = min(max(y0, 0), H - 1), min(max(y0 + 1, 0), H - 1)
            p00, p01, p10, p11 = (img[cy0, cx0].astype(np.float32), img[cy0, cx1].astype(np.float32),
                                  img[cy1, cx0].astype(np.float32), img[cy1, cx1].astype(np.float32))
            top = p00 * (np.float32(1) - ax) + p01 * ax
            bot = p10 * (np.float32(1) - ax) + p11 * ax
            out[y, x] = np.clip(np.rint(top * (np.float32(1) - ay) + bot * ay), 0, 255).astype(np.uint8)
    return out


def get_part_img(img, pts):
    """pytocr/utils/utility.py:53-78 on top of the two calls above"""
    pts = np.asarray(pts).astype(np.float32)
    left, right = int(np.min(pts[:, 0])), int(np.max(pts[:, 0]))
    top, bottom = int(np.min(pts[:, 1])), int(np.max(pts[:, 1]))
    crop = img[top:bottom, left:right, :].copy()
    q = pts - np.array([left, top], np.float32)
    w, h = right - left, bottom - top
    dst = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], np.float32)
    return warp_perspective_replicate_u8(crop, perspective_matrix(q, dst), (w, h))
