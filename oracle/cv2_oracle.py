"""ORACLE (test infrastructure only): literal restatements of the OpenCV calls that sit immediately before the path in the
reference -- cv2.resize(INTER_LINEAR) (pytocr/data/imaug/operators.py:236-250, rec_img_aug.py:112-123), cv2.cvtColor
(COLOR_BGR2GRAY) (deploy/pytorch/infer_rec.py:92-93), cv2.getPerspectiveTransform + cv2.warpPerspective(INTER_LINEAR,
BORDER_REPLICATE) (pytocr/utils/utility.py:53-78).  These are PYTHON call sites: the pinned build is opencv-python 4.1.2.30
(requirements.txt:3-4).

OpenCV is an un-vendored dependency absent from this image and the reference holds no fixture for these calls: PARITY
UNPINNED.  This file follows OpenCV's u8 arithmetic function by function, as published in its sources
(modules/imgproc/src and modules/core/src of the 4.1.x / 3.4.x line):

* resize.cpp, `cv::hal::resize`: `scale = 1. / (dsize / ssize)` in double; an exact 2x2 down-scale of INTER_LINEAR is re-routed to
  INTER_AREA (`ResizeAreaFast_Invoker`: `(a + b + c + d + 2) >> 2`); otherwise per-axis tables `fx = (float)((dx + 0.5) * scale - 0.5)`,
  `sx = cvFloor(fx)`, `fx -= sx`, x clamped to the image with fx = 0, y NOT clamped (rows are clipped when they are fetched:
  `resizeGeneric_Invoker`, `clip(sy0 - ksize2 + 1 + k, 0, ssize.height)`), both coefficients `saturate_cast<short>(c * 2048)` of
  `1.f - fx` and `fx`; `HResizeLinear<uchar,int,short,2048>`: `S[sx] * a0 + S[sx + cn] * a1`; the u8 specialisation of
  `VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,VResizeLinearVec_32s8u>`:
  `uchar((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2)` -- two TRUNCATING products (what `_mm_mulhi_epi16`
  computes), not one rounded 22-bit shift.  (`ipp_resize` declines 8u linear unless `useIPP_NotExact()`.)
* color_rgb.simd.hpp / color.hpp, `RGB2Gray<uchar>` of the 4.1.x line: `CV_DESCALE(b * 1868 + g * 9617 + r * 4899, 14)` (B2Y / G2Y / R2Y,
  `yuv_shift` = 14; the comment there reads "can be changed to 15-shift coeffs", which later releases did: 3735 / 19235 / 9798 at
  shift 15 -- `bgr2gray_u8(img, bits=15)`).
* imgwarp.cpp, `cv::getPerspectiveTransform` (4.x: `solveMethod = DECOMP_LU`) on `cv::solve` -> `hal::LU64f` (matrix_decomp.cpp
  `LUImpl`: partial pivoting with the FIRST largest pivot, `d = -1 / A[i][i]`, row updates `A[j][k] += alpha * A[i][k]`, back
  substitution `s -= A[i][k] * b[k]`, k ascending).
* imgwarp.cpp, `cv::warpPerspective`: `invert(M)` (3x3 double: the adjugate times `1. / det3`), `WarpPerspectiveInvoker` in blocks of
  `bw0 x bh0` destination pixels (`bh0 = min(16, h)`, `bw0 = min(1024 / bh0, w)`, `bh0 = min(1024 / bw0, h)`): per block row
  `X0 = M[0] * x + M[1] * y + M[2]` at the block's first column x, per pixel `W = W0 + M[6] * x1`, `W = W ? 32 / W : 0`,
  `fX = (X0 + M[0] * x1) * W` clamped to the int range, `X = saturate_cast<int>(fX)` (cvRound: half to even),
  `xy = saturate_cast<short>(X >> 5)`, `alpha = (Y & 31) * 32 + (X & 31)`; then `remap` with the fixed-point maps:
  `remapBilinear<FixedPtCast<int,uchar,15>, RemapVec_8u, short>` with `BilinearTab_i` built by `initInterTab2D`
  (`saturate_cast<short>(vy * vx * 32768)` + the sum fix-up, reproduced literally incl. its reads past the 2x2 entry) and
  `(p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + (1 << 14)) >> 15`, BORDER_REPLICATE by clipping sx, sx + 1, sy, sy + 1.

What pins this file is tests/test_oracle_cv2.py: hand-derivable known answers of that arithmetic, including operands on which the
textbook forms (one rounded shift; float lerp + rint) give a different byte.  Plain per-pixel Python loops on purpose: slow, obvious,
and written independently of the product's vectorised host operators (pytorchocr_amd/data/imaug.py, utils/warp.py) and of the
HIP kernels (csrc/preprocess.hip), both of which the tests compare against THIS file.

Only tests/ may import this module."""
import math

import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def _sat_short(v):
    """saturate_cast<short>(float): cvRound (round half to even), then clamp"""
    return int(min(max(int(np.rint(np.float32(v))), -32768), 32767))


def _axis_tables(dst_n, src_n, clamp):
    """resize.cpp's xofs / alpha (clamp=True) or yofs / beta (clamp=False) for INTER_LINEAR, 8-bit (fixpt) path"""
    inv_scale = float(dst_n) / float(src_n)                  # (double)dsize.width / ssize.width
    scale = 1.0 / inv_scale                                  # hal::resize: scale_x = 1. / inv_scale_x
    ofs, coef = [], []
    for d in range(dst_n):
        f = np.float32((d + 0.5) * scale - 0.5)              # double expression, stored in a float
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
        if clamp:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src_n - 1:
                f, s = np.float32(0), src_n - 1
        ofs.append(s)
        coef.append((_sat_short(np.float32(np.float32(1) - f) * np.float32(COEF_ONE)), _sat_short(f * np.float32(COEF_ONE))))
    return ofs, coef


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
    out = np.zeros((dh, dw, chans), np.uint8)
    if sw == 2 * dw and sh == 2 * dh:                        # INTER_LINEAR with iscale_x == iscale_y == 2 runs as INTER_AREA (fast)
        for y in range(dh):
            for x in range(dw):
                for c in range(chans):
                    out[y, x, c] = (s3[2 * y, 2 * x, c] + s3[2 * y, 2 * x + 1, c] + s3[2 * y + 1, 2 * x, c] + s3[2 * y + 1, 2 * x + 1, c] + 2) >> 2
        return out.reshape((dh, dw) if src.ndim == 2 else (dh, dw, chans))
    xofs, alpha = _axis_tables(dw, sw, True)
    yofs, beta = _axis_tables(dh, sh, False)
    for y in range(dh):
        y0 = min(max(yofs[y], 0), sh - 1)                    # clip(sy, 0, ssize.height), clip(sy + 1, ...)
        y1 = min(max(yofs[y] + 1, 0), sh - 1)
        b0, b1 = beta[y]
        for x in range(dw):
            x0 = xofs[x]
            x1 = min(x0 + 1, sw - 1)                         # beyond xmax the row pass is S[sx] * ONE: a1 is 0 there
            a0, a1 = alpha[x]
            for c in range(chans):
                top = int(s3[y0, x0, c] * a0 + s3[y0, x1, c] * a1)       # HResizeLinear, scale 2^11
                bot = int(s3[y1, x0, c] * a0 + s3[y1, x1, c] * a1)
                v = (((b0 * (top >> 4)) >> 16) + ((b1 * (bot >> 4)) >> 16) + 2) >> 2
                out[y, x, c] = v & 255                       # uchar(...): the value never leaves 0..255
    return out.reshape((dh, dw) if src.ndim == 2 else (dh, dw, chans))


def bgr2gray_u8(img, bits=14):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for uint8.  bits=14: the 3.4.x / 4.1.x line (1868, 9617, 4899); bits=15: later releases."""
    cb, cg, cr = (1868, 9617, 4899) if bits == 14 else (3735, 19235, 9798)
    h, w = img.shape[:2]
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            b, g, r = (int(v) for v in img[y, x])
            out[y, x] = (b * cb + g * cg + r * cr + (1 << (bits - 1))) >> bits
    return out


def lu_solve(a, b):
    """hal::LU64f (matrix_decomp.cpp LUImpl) on one m x m system with one right-hand side, in place on copies; None when singular"""
    a = [[float(v) for v in row] for row in a]
    b = [float(v) for v in b]
    m = len(a)
    eps = np.finfo(np.float64).eps * 100
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(a[j][i]) > abs(a[k][i]):
                k = j
        if abs(a[k][i]) < eps:
            return None
        if k != i:
            a[i], a[k] = a[k], a[i]
            b[i], b[k] = b[k], b[i]
        d = -1 / a[i][i]
        for j in range(i + 1, m):
            al = a[j][i] * d
            for k2 in range(i + 1, m):
                a[j][k2] += al * a[i][k2]
            b[j] += al * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, m):
            s -= a[i][k] * b[k]
        b[i] = s / a[i][i]
    return b


def perspective_matrix(src, dst):
    """cv2.getPerspectiveTransform(src, dst) (float32 points; DECOMP_LU): 3x3 double"""
    a = [[0.0] * 8 for _ in range(8)]
    b = [0.0] * 8
    for i in range(4):
        x, y = float(np.float32(src[i][0])), float(np.float32(src[i][1]))
        u, v = float(np.float32(dst[i][0])), float(np.float32(dst[i][1]))
        a[i][0] = a[i + 4][3] = x
        a[i][1] = a[i + 4][4] = y
        a[i][2] = a[i + 4][5] = 1.0
        a[i][6], a[i][7] = -x * u, -y * u
        a[i + 4][6], a[i + 4][7] = -x * v, -y * v
        b[i], b[i + 4] = u, v
    x = lu_solve(a, b)
    if x is None:
        x = [0.0] * 8
    return np.array(x + [1.0], np.float64).reshape(3, 3)


def invert3(m):
    """cv::invert of a 3x3 double matrix (DECOMP_LU): adjugate x 1/det, operation by operation"""
    s = [[float(m[i][j]) for j in range(3)] for i in range(3)]
    d = (s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0])
         + s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]))
    if d == 0.0:
        return np.zeros((3, 3))
    d = 1.0 / d
    t = [(s[1][1] * s[2][2] - s[1][2] * s[2][1]) * d, (s[0][2] * s[2][1] - s[0][1] * s[2][2]) * d, (s[0][1] * s[1][2] - s[0][2] * s[1][1]) * d,
         (s[1][2] * s[2][0] - s[1][0] * s[2][2]) * d, (s[0][0] * s[2][2] - s[0][2] * s[2][0]) * d, (s[0][2] * s[1][0] - s[0][0] * s[1][2]) * d,
         (s[1][0] * s[2][1] - s[1][1] * s[2][0]) * d, (s[0][1] * s[2][0] - s[0][0] * s[2][1]) * d, (s[0][0] * s[1][1] - s[0][1] * s[1][0]) * d]
    return np.array(t, np.float64).reshape(3, 3)


_TAB = None


def bilinear_tab_i():
    """imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt): short[32 * 32][2][2], built the way OpenCV builds it (the fix-up loop reads
    k1, k2 in [1, 3): past the 2x2 entry into the zero-initialised next one)"""
    global _TAB
    if _TAB is None:
        n = 32
        t1 = [(np.float32(1) - np.float32(i) * np.float32(1.0 / n), np.float32(i) * np.float32(1.0 / n)) for i in range(n)]
        flat = [0] * (n * n * 4 + 8)
        for i in range(n):
            for j in range(n):
                base = (i * n + j) * 4
                isum = 0
                for k1 in range(2):
                    for k2 in range(2):
                        v = np.float32(t1[i][k1] * t1[j][k2])
                        flat[base + k1 * 2 + k2] = _sat_short(v * np.float32(32768))
                        isum += flat[base + k1 * 2 + k2]
                if isum != 32768:
                    diff = isum - 32768
                    Mk1 = Mk2 = mk1 = mk2 = 1
                    for k1 in range(1, 3):
                        for k2 in range(1, 3):
                            if flat[base + k1 * 2 + k2] < flat[base + mk1 * 2 + mk2]:
                                mk1, mk2 = k1, k2
                            elif flat[base + k1 * 2 + k2] > flat[base + Mk1 * 2 + Mk2]:
                                Mk1, Mk2 = k1, k2
                    if diff < 0:
                        flat[base + Mk1 * 2 + Mk2] -= diff
                    else:
                        flat[base + mk1 * 2 + mk2] -= diff
        _TAB = np.array(flat[:n * n * 4], np.int64).reshape(n * n, 4)
    return _TAB


def _clip(v, lo, hi):
    return min(max(v, lo), hi)


def warp_perspective_replicate_u8(img, M, dsize):
    """cv2.warpPerspective(img, M, (w, h), flags=INTER_LINEAR, borderMode=BORDER_REPLICATE) for uint8 HxWxC"""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    m = invert3(np.asarray(M, np.float64)).reshape(9)
    tab = bilinear_tab_i()
    src = img.astype(np.int64)
    out = np.zeros((h, w) + img.shape[2:], np.uint8)
    bh0 = min(16, h)
    bw0 = min(1024 // bh0, w)
    bh0 = min(1024 // bw0, h)
    imin, imax = float(-2 ** 31), float(2 ** 31 - 1)
    for y in range(h):
        for xb in range(0, w, bw0):
            X0 = m[0] * xb + m[1] * y + m[2]
            Y0 = m[3] * xb + m[4] * y + m[5]
            W0 = m[6] * xb + m[7] * y + m[8]
            for x1 in range(min(bw0, w - xb)):
                Wd = W0 + m[6] * x1
                Wd = 32.0 / Wd if Wd != 0 else 0.0
                fX = max(imin, min(imax, (X0 + m[0] * x1) * Wd))
                fY = max(imin, min(imax, (Y0 + m[3] * x1) * Wd))
                X, Y = int(np.rint(fX)), int(np.rint(fY))
                sx, sy = _clip(X >> 5, -32768, 32767), _clip(Y >> 5, -32768, 32767)
                wt = tab[(Y & 31) * 32 + (X & 31)]
                x0, xx1 = _clip(sx, 0, W - 1), _clip(sx + 1, 0, W - 1)
                y0, yy1 = _clip(sy, 0, H - 1), _clip(sy + 1, 0, H - 1)
                t = src[y0, x0] * wt[0] + src[y0, xx1] * wt[1] + src[yy1, x0] * wt[2] + src[yy1, xx1] * wt[3]
                out[y, xb + x1] = np.clip((t + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
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
