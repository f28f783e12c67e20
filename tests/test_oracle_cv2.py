"""oracle/cv2_oracle.py (the restated OpenCV calls next to the path) against hand-derivable known answers of OpenCV's u8 arithmetic
-- including operands on which the textbook forms (one rounded 22-bit shift; float lerp + rint; LAPACK) give a different byte --
and the product's host operators (data/imaug.py, utils/warp.py) against the oracle.  PARITY UNPINNED against OpenCV itself (absent
from this image; the reference holds no fixture)."""
import numpy as np

from oracle import cv2_oracle as cvo


def test_resize_known_answers():
    # identity
    a = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert np.array_equal(cvo.resize_linear_u8(a, (4, 3)), a)
    # 2 -> 4 along x: centres at -0.25, 0.25, 0.75, 1.25 -> clamp, 0.25, 0.75, clamp:  weights 2048*(0.25) = 512
    row = np.array([[0, 100]], np.uint8)
    assert cvo.resize_linear_u8(row, (4, 1)).tolist() == [[0, 25, 75, 100]]
    # 4 -> 2 along x only: centres at 0.5, 2.5 -> exact midpoints
    row = np.array([[10, 20, 30, 50]], np.uint8)
    assert cvo.resize_linear_u8(row, (2, 1)).tolist() == [[15, 40]]
    # 0.25 -> 0, 0.75 -> 1 at the final (+ 2) >> 2
    assert cvo.resize_linear_u8(np.array([[0, 1]], np.uint8), (4, 1)).tolist() == [[0, 0, 1, 1]]
    # both axes, colour: a constant image stays constant for any size (the truncation loses < 2 quarter units, the + 2 restores them)
    c = np.full((5, 7, 3), 137, np.uint8)
    assert (cvo.resize_linear_u8(c, (11, 3)) == 137).all() and (cvo.resize_linear_u8(c, (3, 13)) == 137).all()
    # 3 -> 2 along y: centres 0.25, 1.75: values v0*0.75 + v1*0.25, v1*0.25 + v2*0.75 (coefficients 1536 / 512: exact products)
    col = np.array([[0], [40], [200]], np.uint8)
    assert cvo.resize_linear_u8(col, (1, 2))[:, 0].tolist() == [10, 160]


def test_resize_vertical_pass_truncates_like_VResizeLinear_u8():
    """5 -> 3 rows: scale 1/(3/5), row 0 at (float)(0.5 * 1.6667 - 0.5) = 0.3333 -> rows 0, 1 with coefficients
    saturate_cast<short>(0.6667 * 2048) = 1365 and 683.  Column values 3, 2 (S = v * 2048 after the row pass):
        OpenCV:   ((1365 * (6144 >> 4)) >> 16) + ((683 * (4096 >> 4)) >> 16) + 2 >> 2 = (7 + 2 + 2) >> 2 = 2       (7.998 and 2.668 truncated)
        textbook: (6144 * 1365 + 4096 * 683 + 2^21) >> 22 = 2.667 + 0.5 -> 3"""
    col = np.array([[3], [2], [0], [0], [0]], np.uint8)
    yofs, beta = cvo._axis_tables(3, 5, False)
    assert yofs[0] == 0 and beta[0] == (1365, 683)
    assert (3 * 2048 * 1365 + 2 * 2048 * 683 + (1 << 21)) >> 22 == 3
    assert cvo.resize_linear_u8(col, (1, 3))[0, 0] == 2
    # the y table is not clamped: row -1 / row H are clipped when fetched (3 -> 5: first row at -0.2 -> offset -1, fraction 0.8)
    yofs, beta = cvo._axis_tables(5, 3, False)
    assert yofs[0] == -1 and beta[0] == (410, 1638) and yofs[4] == 2
    xofs, alpha = cvo._axis_tables(5, 3, True)
    assert xofs[0] == 0 and alpha[0] == (2048, 0) and xofs[4] == 2 and alpha[4] == (2048, 0)
    img = np.array([[7, 9, 250]], np.uint8).T
    assert cvo.resize_linear_u8(img, (1, 5))[[0, 4], 0].tolist() == [7, 250]


def test_resize_exact_half_runs_as_area():
    """cv::resize re-routes INTER_LINEAR with iscale_x == iscale_y == 2 to INTER_AREA: (a + b + c + d + 2) >> 2"""
    img = np.array([[1, 2, 250, 251], [2, 2, 251, 253]], np.uint8)
    assert cvo.resize_linear_u8(img, (2, 1)).tolist() == [[(1 + 2 + 2 + 2 + 2) >> 2, (250 + 251 + 251 + 253 + 2) >> 2]]
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (8, 12, 3), dtype=np.uint8)
    exp = (big.astype(int)[0::2, 0::2] + big.astype(int)[0::2, 1::2] + big.astype(int)[1::2, 0::2] + big.astype(int)[1::2, 1::2] + 2) >> 2
    assert np.array_equal(cvo.resize_linear_u8(big, (6, 4)), exp.astype(np.uint8))


def test_bgr2gray_known_answers():
    img = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30], [0, 0, 5]]], np.uint8)
    # 14-bit coefficients of the 4.1.x line (1868 + 9617 + 4899 = 16384): 29.07 -> 29, 149.68 -> 150, 76.25 -> 76, 255, 21.85 -> 22, 1.4951 -> 1
    assert cvo.bgr2gray_u8(img)[0].tolist() == [29, 150, 76, 255, 22, 1]
    assert (5 * 4899 + (1 << 13)) >> 14 == 1 and (5 * 9798 + (1 << 14)) >> 15 == 1
    # a pixel on which the later 15-bit coefficients (3735, 19235, 9798) give another byte: B = 250 alone is
    # 250 * 1868 / 16384 = 28.5034 -> 29 but 250 * 3735 / 32768 = 28.4958 -> 28
    px = np.array([[[250, 0, 0]]], np.uint8)
    assert cvo.bgr2gray_u8(px, bits=14)[0, 0] == 29 and cvo.bgr2gray_u8(px, bits=15)[0, 0] == 28
    assert cvo.bgr2gray_u8(px)[0, 0] == 29                        # the default follows the pinned opencv-python 4.1.2.30


def test_remap_weight_table():
    tab = cvo.bilinear_tab_i()
    assert tab.shape == (1024, 4) and (tab.sum(1) == 32768).all()
    assert tab[0].tolist() == [32767, 0, 0, 1]                    # 32768 does not fit a short; the fix-up puts the missing 1 on the last weight
    assert tab[16 * 32 + 16].tolist() == [8192] * 4
    assert tab[8].tolist() == [24 * 32 * 32, 8 * 32 * 32, 0, 0]   # ay = 0, ax = 8


def test_lu_solve_and_invert():
    # LUImpl: partial pivoting picks the first largest pivot; 2x2 check by hand: [[0, 2], [4, 1]] x = [2, 9] -> x = [2, 1]
    assert cvo.lu_solve([[0.0, 2.0], [4.0, 1.0]], [2.0, 9.0]) == [2.0, 1.0]
    assert cvo.lu_solve([[1.0, 2.0], [2.0, 4.0]], [1.0, 2.0]) is None
    m = np.array([[2.0, 0, 3], [0, 4, -1], [0, 0, 1]])
    assert np.array_equal(cvo.invert3(m), np.array([[0.5, 0, -1.5], [0, 0.25, 0.25], [0, 0, 1]]))
    # a scale by 2: exact in every step
    src = np.array([[0, 0], [10, 0], [10, 10], [0, 10]], np.float32)
    assert np.array_equal(cvo.perspective_matrix(src, src * 2), np.diag([2.0, 2.0, 1.0]))


def test_warp_known_answers():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    eye = np.eye(3)
    assert np.array_equal(cvo.warp_perspective_replicate_u8(img, eye, (12, 9)), img)
    # integer translation by (+2, +1): dst(x, y) = src(x - 2, y - 1), border replicated
    t = np.array([[1, 0, 2], [0, 1, 1], [0, 0, 1]], float)
    out = cvo.warp_perspective_replicate_u8(img, t, (12, 9))
    assert np.array_equal(out[1:, 2:], img[:-1, :-2]) and np.array_equal(out[0, 2:], img[0, :-2]) and np.array_equal(out[1:, 0], img[:-1, 0])
    # half-pixel translation in x: weights 16384 / 16384, (a + b) * 16384 + 16384 >> 15 = (a + b + 1) >> 1: a tie rounds UP
    # (float lerp + rint would round 2.5 to 2)
    t = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], float)
    out = cvo.warp_perspective_replicate_u8(img, t, (12, 9))
    exp = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(out[:, 1:], exp.astype(np.uint8))
    two_three = np.array([[[2, 2, 2], [3, 3, 3], [3, 3, 3]]], np.uint8)
    assert cvo.warp_perspective_replicate_u8(two_three, t, (3, 1))[0, 1, 0] == 3
    # perspective matrix maps the four source corners onto the destination corners
    src = np.array([[1, 2], [9, 1], [10, 7], [0, 8]], np.float32)
    dst = np.array([[0, 0], [7, 0], [7, 5], [0, 5]], np.float32)
    M = cvo.perspective_matrix(src, dst)
    for s, d in zip(src, dst):
        q = M @ np.array([s[0], s[1], 1.0])
        assert np.allclose(q[:2] / q[2], d, atol=1e-9)


def test_product_host_operators_equal_the_oracle():
    from pytorchocr_amd.data.imaug import bgr_to_gray, resize_bilinear
    from pytorchocr_amd.utils.warp import get_part_img, get_perspective_transforms, invert_transforms
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    for dsize in ((64, 32), (17, 9), (31, 23), (5, 40), (62, 46), (47, 23)):
        assert np.array_equal(resize_bilinear(img, dsize), cvo.resize_linear_u8(img, dsize)), dsize
        assert np.array_equal(resize_bilinear(img[:, :, 0], dsize), cvo.resize_linear_u8(img[:, :, 0], dsize)), dsize
    even = rng.integers(0, 256, (24, 30, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear(even, (15, 12)), cvo.resize_linear_u8(even, (15, 12)))          # the 2x2 area route
    assert np.array_equal(bgr_to_gray(img), cvo.bgr2gray_u8(img))
    big = rng.integers(0, 256, (40, 90, 3), dtype=np.uint8)
    boxes = ([[5, 6], [40, 4], [42, 20], [7, 22]], [[10, 2], [20, 3], [19, 35], [9, 34]], [[0, 0], [59, 0], [59, 39], [0, 39]],
             [[2, 3], [86, 5], [88, 30], [1, 27]])                                                       # the last one is wider than one 64-column block
    for pts in boxes:
        assert np.array_equal(get_part_img(big, np.array(pts, np.int16)), cvo.get_part_img(big, np.array(pts, np.int16))), pts
    # the batched LU / inverse give, for every box, the bits of the one-box oracle
    src = np.array(boxes, np.float32)
    dst = np.array([[[0, 0], [30, 0], [30, 12], [0, 12]]] * len(boxes), np.float32)
    ms = get_perspective_transforms(src, dst)
    for k in range(len(boxes)):
        assert np.array_equal(ms[k], cvo.perspective_matrix(src[k], dst[k]))
        assert np.array_equal(invert_transforms(ms[k]), cvo.invert3(ms[k]))
