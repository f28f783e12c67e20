"""Text-box crop for run_ocr: `get_part_img` mirrors reference pytocr/utils/utility.py:53-78 (axis-aligned crop of the
box's bounding rectangle, cv2.getPerspectiveTransform to the crop's corners, cv2.warpPerspective with INTER_LINEAR and
BORDER_REPLICATE).  Restated in numpy from OpenCV's documented algorithm (source coordinates quantised to 1/32 pixel as
cv2's remap tables do); UNPINNED against OpenCV (absent from this image)."""
import numpy as np


def get_perspective_transforms(src, dst):
    """n 3x3 matrices mapping src[i][k] -> dst[i][k] (k = 0..3), each solved in double like cv2.getPerspectiveTransform
    (one batched LAPACK call: every 8x8 system is factorised on its own, so a batch gives the single-box result bit for bit)."""
    src = np.asarray(src, np.float64).reshape(-1, 4, 2)
    dst = np.asarray(dst, np.float64).reshape(-1, 4, 2)
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
    x = np.linalg.solve(a, b[:, :, None])[:, :, 0]            # (chunks on a thread pool were tried: 3x slower on the GPU box's host share)
    return np.concatenate([x, np.ones((n, 1))], 1).reshape(n, 3, 3)


def invert_transforms(m):
    """batched 3x3 inverse"""
    return np.linalg.inv(np.asarray(m, np.float64))


def get_perspective_transform(src, dst):
    """3x3 matrix mapping src[i] -> dst[i] (4 points), solved in double like cv2.getPerspectiveTransform."""
    return get_perspective_transforms(np.asarray(src)[None], np.asarray(dst)[None])[0]


def warp_perspective_replicate(img, M, dsize):
    w, h = dsize
    Minv = np.linalg.inv(M)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    den = Minv[2, 0] * xs + Minv[2, 1] * ys + Minv[2, 2]
    den = np.where(den != 0, 1.0 / den, 0.0)
    fx = (Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]) * den
    fy = (Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]) * den
    X = np.rint(fx * 32).astype(np.int64)
    Y = np.rint(fy * 32).astype(np.int64)
    x0, y0 = X >> 5, Y >> 5
    ax = (X & 31).astype(np.float32) / 32
    ay = (Y & 31).astype(np.float32) / 32
    H, W = img.shape[:2]
    cx0, cx1 = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    cy0, cy1 = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    im = img.astype(np.float32)
    if im.ndim == 3:
        ax = ax[..., None]; ay = ay[..., None]
    top = im[cy0, cx0] * (1 - ax) + im[cy0, cx1] * ax
    bot = im[cy1, cx0] * (1 - ax) + im[cy1, cx1] * ax
    out = top * (1 - ay) + bot * ay
    return np.clip(np.rint(out), 0, 255).astype(img.dtype) if img.dtype == np.uint8 else out.astype(img.dtype)


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
