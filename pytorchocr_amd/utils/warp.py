"""Text-box crop for run_ocr: `get_part_img` mirrors reference pytocr/utils/utility.py:53-78 (axis-aligned crop of the
box's bounding rectangle, cv2.getPerspectiveTransform to the crop's corners, cv2.warpPerspective with INTER_LINEAR and
BORDER_REPLICATE).  Restated in numpy in OpenCV's own arithmetic (imgwarp.cpp / matrix_decomp.cpp of the pinned opencv-python
4.1.2.30): the 8x8 system solved by OpenCV's LU (`hal::LU64f`, partial pivoting), `cv::invert`'s 3x3 adjugate form, source
coordinates formed per destination block as `WarpPerspectiveInvoker` forms them and quantised to 1/32 pixel, `remapBilinear`'s
15-bit integer weight table with `(sum + 2^14) >> 15`.  Held bit-exact to oracle/cv2_oracle.py by tests/test_oracle_cv2.py;
UNPINNED against OpenCV itself (absent from this image)."""
import numpy as np


def get_perspective_transforms(src, dst):
    """n 3x3 matrices mapping src[i][k] -> dst[i][k] (k = 0..3): cv2.getPerspectiveTransform (DECOMP_LU) for a batch of boxes.
    Every step of OpenCV's LUImpl (first largest pivot, d = -1 / pivot, row updates, back substitution in ascending k) is one
    numpy operation over the batch, so a system's result does not depend on the batch it is solved in."""
    src = np.asarray(src, np.float32).astype(np.float64).reshape(-1, 4, 2)
    dst = np.asarray(dst, np.float32).astype(np.float64).reshape(-1, 4, 2)
    n = src.shape[0]
    a = np.zeros((n, 8, 8), np.float64)
    b = np.zeros((n, 8), np.float64)
    for i in range(4):
        a[:, i, 0] = a[:, i + 4, 3] = src[:, i, 0]
        a[:, i, 1] = a[:, i + 4, 4] = src[:, i, 1]
        a[:, i, 2] = a[:, i + 4, 5] = 1
        a[:, i, 6] = -src[:, i, 0] * dst[:, i, 0]
        a[:, i, 7] = -src[:, i, 1] * dst[:, i, 0]
        a[:, i + 4, 6] = -src[:, i, 0] * dst[:, i, 1]
        a[:, i + 4, 7] = -src[:, i, 1] * dst[:, i, 1]
        b[:, i] = dst[:, i, 0]
        b[:, i + 4] = dst[:, i, 1]
    m = 8
    idx = np.arange(n)
    singular = np.zeros(n, bool)
    eps = np.finfo(np.float64).eps * 100
    for i in range(m):
        k = i + np.argmax(np.abs(a[:, i:, i]), axis=1)              # argmax returns the FIRST maximum, like the strict `>` scan
        singular |= np.abs(a[idx, k, i]) < eps
        ri, rk = a[idx, i, :].copy(), a[idx, k, :].copy()           # whole rows: columns left of i are never read again
        a[idx, i, :], a[idx, k, :] = rk, ri
        bi, bk = b[idx, i].copy(), b[idx, k].copy()
        b[idx, i], b[idx, k] = bk, bi
        with np.errstate(divide="ignore", invalid="ignore"):
            d = -1.0 / a[:, i, i]
            for j in range(i + 1, m):
                al = a[:, j, i] * d
                a[:, j, i + 1:] += al[:, None] * a[:, i, i + 1:]
                b[:, j] += al * b[:, i]
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(m - 1, -1, -1):
            s_ = b[:, i].copy()
            for k in range(i + 1, m):
                s_ -= a[:, i, k] * b[:, k]
            b[:, i] = s_ / a[:, i, i]
    b[singular] = 0.0                                                # cv::solve returns false and the caller keeps going
    return np.concatenate([b, np.ones((n, 1))], 1).reshape(n, 3, 3)


def invert_transforms(m):
    """batched cv::invert of 3x3 double matrices: adjugate x (1 / det3), operation by operation"""
    s = np.asarray(m, np.float64).reshape(-1, 3, 3)
    det = (s[:, 0, 0] * (s[:, 1, 1] * s[:, 2, 2] - s[:, 1, 2] * s[:, 2, 1]) - s[:, 0, 1] * (s[:, 1, 0] * s[:, 2, 2] - s[:, 1, 2] * s[:, 2, 0])
           + s[:, 0, 2] * (s[:, 1, 0] * s[:, 2, 1] - s[:, 1, 1] * s[:, 2, 0]))
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.where(det != 0, 1.0 / det, 0.0)
    t = np.stack([(s[:, 1, 1] * s[:, 2, 2] - s[:, 1, 2] * s[:, 2, 1]) * d, (s[:, 0, 2] * s[:, 2, 1] - s[:, 0, 1] * s[:, 2, 2]) * d,
                  (s[:, 0, 1] * s[:, 1, 2] - s[:, 0, 2] * s[:, 1, 1]) * d, (s[:, 1, 2] * s[:, 2, 0] - s[:, 1, 0] * s[:, 2, 2]) * d,
                  (s[:, 0, 0] * s[:, 2, 2] - s[:, 0, 2] * s[:, 2, 0]) * d, (s[:, 0, 2] * s[:, 1, 0] - s[:, 0, 0] * s[:, 1, 2]) * d,
                  (s[:, 1, 0] * s[:, 2, 1] - s[:, 1, 1] * s[:, 2, 0]) * d, (s[:, 0, 1] * s[:, 2, 0] - s[:, 0, 0] * s[:, 2, 1]) * d,
                  (s[:, 0, 0] * s[:, 1, 1] - s[:, 0, 1] * s[:, 1, 0]) * d], 1).reshape(-1, 3, 3)
    t[det == 0] = 0.0
    return t.reshape(np.asarray(m).shape)


def get_perspective_transform(src, dst):
    """3x3 matrix mapping src[i] -> dst[i] (4 points), solved in double like cv2.getPerspectiveTransform."""
    return get_perspective_transforms(np.asarray(src)[None], np.asarray(dst)[None])[0]


def warp_block_width(w, h):
    """WarpPerspectiveInvoker's destination block width: coordinates are formed as (value at the block's first column) + M * x1"""
    bh0 = min(16, h)
    return min(1024 // bh0, w)


def warp_perspective_replicate(img, M, dsize):
    w, h = dsize
    mi = invert_transforms(np.asarray(M, np.float64)).reshape(9)
    bw0 = warp_block_width(w, h)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.int64), np.arange(h, dtype=np.int64))
    xb = ((xs // bw0) * bw0).astype(np.float64)
    x1 = (xs - (xs // bw0) * bw0).astype(np.float64)
    ysf = ys.astype(np.float64)
    X0 = mi[0] * xb + mi[1] * ysf + mi[2]
    Y0 = mi[3] * xb + mi[4] * ysf + mi[5]
    W0 = mi[6] * xb + mi[7] * ysf + mi[8]
    Wd = W0 + mi[6] * x1
    with np.errstate(divide="ignore", invalid="ignore"):
        Wd = np.where(Wd != 0, 32.0 / Wd, 0.0)
    lim = (float(-2 ** 31), float(2 ** 31 - 1))
    X = np.rint(np.clip((X0 + mi[0] * x1) * Wd, *lim)).astype(np.int64)          # saturate_cast<int>: cvRound, half to even
    Y = np.rint(np.clip((Y0 + mi[3] * x1) * Wd, *lim)).astype(np.int64)
    x0, y0 = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = X & 31, Y & 31
    H, W = img.shape[:2]
    cx0, cx1 = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    cy0, cy1 = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    if img.dtype == np.uint8:
        # BilinearTab_i: (32 - ay)(32 - ax) * 32, ...; the one entry that does not fit a short, (0, 0), is stored as {32767, 0, 0, 1}
        w00, w01, w10, w11 = (32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32
        whole = (ax == 0) & (ay == 0)
        w00 = np.where(whole, 32767, w00)
        w11 = np.where(whole, 1, w11)
        im = img.astype(np.int64)
        if im.ndim == 3:
            w00, w01, w10, w11 = w00[..., None], w01[..., None], w10[..., None], w11[..., None]
        out = (im[cy0, cx0] * w00 + im[cy0, cx1] * w01 + im[cy1, cx0] * w10 + im[cy1, cx1] * w11 + (1 << 14)) >> 15
        return np.clip(out, 0, 255).astype(np.uint8)
    im = img.astype(np.float32)
    fx, fy = ax.astype(np.float32) / 32, ay.astype(np.float32) / 32
    if im.ndim == 3:
        fx = fx[..., None]; fy = fy[..., None]
    top = im[cy0, cx0] * (1 - fx) + im[cy0, cx1] * fx
    bot = im[cy1, cx0] * (1 - fx) + im[cy1, cx1] * fx
    return (top * (1 - fy) + bot * fy).astype(img.dtype)


def get_part_img(img, pts):
    """pts: text box vertices, shape (4, 2)"""
    pts = pts.astype(np.float32)
    left, right = int(np.min(pts[:, 0])), int(np.max(pts[:, 0]))
    top, bottom = int(np.min(pts[:, 1])), int(np.max(pts[:, 1]))
    img_crop = img[top:bottom, left:right, :].copy()
    pts = pts - np.array([left, top], dtype=np.float32)
    w, h = int(right - left), int(bottom - top)
    dst = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], dtype=np.float32)
    M = get_perspective_transform(pts, dst)
    return warp_perspective_replicate(img_crop, M, (w, h))
