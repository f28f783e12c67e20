"""Entry-point counterparts (Deter / Recer / OCRer) on the GPU against the oracle pipeline on the same preprocessed input."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc_oracle, dbpost, model_oracle
from pytorchocr_amd.utils.synth import synth_state_dict, uniform01

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "pytorchocr_amd", "configs")


def _ckpt(tmp_path, contract, name, wrap=False, prefix=""):
    sd = synth_state_dict(contract[name])
    t = {prefix + k: torch.from_numpy(v) for k, v in sd.items()}
    p = os.path.join(str(tmp_path), name + ".pth")
    torch.save({"state_dict": t} if wrap else t, p)
    return p, sd


def test_deter_matches_oracle_pipeline(tmp_path, contract):
    from pytorchocr_amd.deploy.infer_det import Deter
    from pytorchocr_amd.utils.utility import sort_boxes
    ck, sd = _ckpt(tmp_path, contract, "det_r18_db", wrap=True, prefix="module.")        # DDP-style checkpoint
    det = Deter(os.path.join(CFG, "det", "det_r18_db.yml"), ck, gpu_id=0)
    img = (uniform01(200 * 320 * 3, 5).reshape(200, 320, 3) * 255).astype(np.uint8)      # BGR ndarray instead of a path
    boxes = det.run(img)
    x, shape = det._prep(img)
    assert tuple(x.shape) == (3, 736, 1184)
    maps = model_oracle.dbnet_r18_forward(sd, x[None])["maps"].numpy()[0, 0]
    with torch.no_grad():
        got_maps = det.deter(x[None].cuda())["maps"].cpu().numpy()[0, 0]
    assert np.abs(got_maps - maps).max() <= 1e-4
    exp = dbpost.boxes_from_bitmap(got_maps, dbpost.binarize(got_maps, 0.3), 0.5, 1.7, 320, 200)
    exp = sort_boxes(exp.astype(np.int16))
    assert len(boxes) == len(exp) and all(np.array_equal(a, b) for a, b in zip(boxes, exp))
    two = det.run_batch([img, img[::-1].copy()])
    assert len(two) == 2 and all(np.array_equal(a, b) for a, b in zip(two[0], boxes))


def test_recer_and_ocrer_batched_equals_per_crop(tmp_path, contract, monkeypatch):
    from pytorchocr_amd.deploy.infer_rec import Recer
    from pytorchocr_amd.deploy.run_ocr import OCRer
    ck, sd = _ckpt(tmp_path, contract, "rec_vgg_bilstm_ctc")
    rec = Recer(os.path.join(CFG, "rec", "rec_vgg_bilstm_ctc.yml"), ck)
    rng = np.random.default_rng(3)
    crops = []
    for k in range(5):
        w = 60 + 45 * k
        base = np.repeat(rng.integers(0, 255, size=(1, (w + 7) // 8, 1)), 8, axis=1)[:, :w]
        crops.append(np.clip(base + rng.integers(0, 30, size=(24 + 4 * k, w, 3)), 0, 255).astype(np.uint8))
    single = [rec.run(c) for c in crops]
    batched = rec.run_batch(crops)
    assert [t for t, _ in single] == [t for t, _ in batched]
    assert np.allclose([p for _, p in single], [p for _, p in batched], atol=0.011, equal_nan=True)
    # oracle on the same preprocessed crops
    x = torch.stack([rec._prep(c) for c in crops])
    ref = ctc_oracle.ctc_label_decode(model_oracle.crnn_forward(sd, x).numpy(), ctc_oracle.load_characters(
        os.path.join(ROOT, "pytorchocr_amd", "utils", "char_dict_6623.txt")))
    assert [t for t, _ in ref] == [t for t, _ in batched]

    dck, _ = _ckpt(tmp_path, contract, "det_r18_db")
    ocr = OCRer(os.path.join(CFG, "det", "det_r18_db.yml"), dck, os.path.join(CFG, "rec", "rec_vgg_bilstm_ctc.yml"), ck)
    img = (uniform01(240 * 400 * 3, 9).reshape(240, 400, 3) * 255).astype(np.uint8)
    fake = [np.array([[20, 30], [220, 34], [219, 70], [19, 66]], np.int16), np.array([[300, 20], [330, 20], [330, 200], [300, 200]], np.int16)]
    monkeypatch.setattr(ocr.det, "run", lambda im: fake)
    res = ocr.run(img)
    assert len(res) == 2 and all(len(r) == 3 and isinstance(r[1], str) for r in res)
    from pytorchocr_amd.utils.warp import get_part_img
    tall = get_part_img(img, fake[1])
    assert tall.shape[0] >= 1.5 * tall.shape[1]                       # second box is rotated 90 degrees before recognition
    exp1 = rec.run(np.ascontiguousarray(np.rot90(tall, 1)))
    assert res[1][1] == exp1[0]


def test_gpu_preprocess_is_bit_exact_with_the_host_operators(tmp_path, contract):
    """ptocr_preprocess_u8_f32 == DetResizeForTest + ToTensor + Normalize (numpy/torch on the host), bit for bit."""
    from pytorchocr_amd.data.gpu_preprocess import det_preprocess
    from pytorchocr_amd.deploy.infer_det import Deter
    ck, _ = _ckpt(tmp_path, contract, "det_r18_db")
    det = Deter(os.path.join(CFG, "det", "det_r18_db.yml"), ck, gpu_id=0)
    for (h, w), seed in (((200, 320), 5), ((97, 131), 6), ((736, 1280), 7), ((64, 64), 8)):
        img = (uniform01(h * w * 3, seed).reshape(h, w, 3) * 255).astype(np.uint8)
        x, shape = det._prep(img)
        rs, nm = det._gpu_ops()
        rh, rw = rs.target_size(h, w)
        x4 = det_preprocess(img, (rh, rw), nm.mean, nm.std, det.det_device, swap_rb=True).cpu()
        assert tuple(x4.shape) == (1, rh, rw, 4) and float(x4[..., 3].abs().max()) == 0
        assert torch.equal(x4[0, :, :, :3].permute(2, 0, 1), x), (h, w)
    img = (uniform01(200 * 320 * 3, 5).reshape(200, 320, 3) * 255).astype(np.uint8)
    a, b = det.run(img), det.run_gpu(img)
    assert len(a) == len(b) and all(np.array_equal(p, q) for p, q in zip(a, b))


def test_gpu_crops_and_rec_preprocess_match_host(tmp_path, contract, monkeypatch):
    from pytorchocr_amd.data.gpu_preprocess import rec_preprocess, warp_crops
    from pytorchocr_amd.deploy.infer_rec import Recer
    from pytorchocr_amd.deploy.run_ocr import OCRer
    from pytorchocr_amd.utils.warp import get_part_img
    ck, _ = _ckpt(tmp_path, contract, "rec_vgg_bilstm_ctc")
    dck, _ = _ckpt(tmp_path, contract, "det_r18_db")
    img = (uniform01(240 * 400 * 3, 9).reshape(240, 400, 3) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:240, 0:400]
    img = np.clip(img * 0.3 + ((xx // 8 * 37) % 200)[..., None], 0, 255).astype(np.uint8)
    boxes = [np.array([[20, 30], [220, 34], [219, 70], [19, 66]], np.int16), np.array([[300, 20], [330, 20], [330, 200], [300, 200]], np.int16),
             np.array([[50, 100], [150, 120], [140, 160], [40, 140]], np.int16)]
    dev = torch.device("cuda:0")
    buf, metas = warp_crops(torch.from_numpy(img).to(dev), boxes)
    host = []
    for b, (off, h, w) in zip(boxes, metas):
        c = get_part_img(img, b)
        if c.shape[0] >= 1.5 * c.shape[1]:
            c = np.rot90(c, 1)
        host.append(np.ascontiguousarray(c))
        got = buf[off:off + h * w * 3].cpu().numpy().reshape(h, w, 3)
        assert got.shape == c.shape and np.array_equal(got, c)
    rec = Recer(os.path.join(CFG, "rec", "rec_vgg_bilstm_ctc.yml"), ck)
    x4 = rec_preprocess(buf, metas, [1, 32, 320], dev).cpu()
    for i, c in enumerate(host):
        assert torch.equal(x4[i, :, :, 0], rec._prep(c)[0])
    ocr_h = OCRer(os.path.join(CFG, "det", "det_r18_db.yml"), dck, os.path.join(CFG, "rec", "rec_vgg_bilstm_ctc.yml"), ck)
    ocr_g = OCRer(os.path.join(CFG, "det", "det_r18_db.yml"), dck, os.path.join(CFG, "rec", "rec_vgg_bilstm_ctc.yml"), ck, gpu_preprocess=True)
    for o in (ocr_h, ocr_g):
        monkeypatch.setattr(o.det, "run", lambda im: boxes)
    rh, rg = ocr_h.run(img), ocr_g.run(img)
    assert [r[1] for r in rh] == [r[1] for r in rg] and len(rg) == 3


def test_gpu_preprocess_and_crops_against_the_cv2_oracle():
    """HIP pre-process kernels against oracle/cv2_oracle.py (the literal per-pixel restatement of cv2.resize / cvtColor /
    warpPerspective with hand-derived known answers, tests/test_oracle_cv2.py) -- not against the product's own host operators."""
    from oracle import cv2_oracle as cvo
    from pytorchocr_amd.data.gpu_preprocess import det_preprocess, rec_preprocess, warp_crops
    dev = torch.device("cuda:0")
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for (h, w, rh, rw), seed in (((37, 53, 64, 96), 1), ((60, 41, 32, 32), 2), ((40, 64, 40, 64), 3)):
        img = (uniform01(h * w * 3, seed).reshape(h, w, 3) * 255).astype(np.uint8)
        x4 = det_preprocess(img, (rh, rw), mean, std, dev, swap_rb=True).cpu().numpy()[0]
        rs = cvo.resize_linear_u8(np.ascontiguousarray(img[:, :, ::-1]), (rw, rh)).astype(np.float32)            # RGB
        exp = (torch.from_numpy(rs).div(255) - torch.tensor(mean)) / torch.tensor(std)
        assert np.array_equal(x4[:, :, :3], exp.numpy()), (h, w)
    img = (uniform01(48 * 80 * 3, 4).reshape(48, 80, 3) * 255).astype(np.uint8)
    boxes = [np.array([[4, 5], [60, 8], [58, 30], [3, 27]], np.int16), np.array([[66, 2], [76, 2], [76, 40], [66, 40]], np.int16)]
    buf, metas = warp_crops(torch.from_numpy(img).to(dev), boxes)
    crops = []
    for b, (off, ch, cw) in zip(boxes, metas):
        c = cvo.get_part_img(img, b)
        if c.shape[0] >= 1.5 * c.shape[1]:
            c = np.rot90(c, 1)
        crops.append(np.ascontiguousarray(c))
        assert np.array_equal(buf[off:off + ch * cw * 3].cpu().numpy().reshape(ch, cw, 3), c)
    x4 = rec_preprocess(buf, metas, [1, 32, 320], dev).cpu().numpy()
    for i, c in enumerate(crops):
        g = cvo.bgr2gray_u8(c)
        rw = min(320, int(np.ceil(32 * g.shape[1] / float(g.shape[0]))))
        r = cvo.resize_linear_u8(g, (rw, 32)).astype(np.float32)
        exp = np.zeros((32, 320), np.float32)
        exp[:, :rw] = (r / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
        assert np.array_equal(x4[i, :, :, 0], exp)


@pytest.mark.parametrize("seed", range(8 + int(os.environ.get("PTOCR_CV2_FUZZ", "0"))))        # PTOCR_CV2_FUZZ=n: n more seeds
def test_gpu_preprocess_random_sizes_against_the_cv2_oracle(seed):
    """random source / target sizes (up- and down-scaling, the exact 2x route, 1-pixel-wide targets) and random quadrilaterals through
    the HIP pre-process and crop kernels, bit for bit against oracle/cv2_oracle.py"""
    from oracle import cv2_oracle as cvo
    from pytorchocr_amd.data.gpu_preprocess import det_preprocess, rec_preprocess, warp_crops
    rng = np.random.default_rng(8000 + seed)
    dev = torch.device("cuda:0")
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    h, w = int(rng.integers(20, 90)), int(rng.integers(20, 110))
    if rng.uniform() < 0.25:
        rh, rw = 32 * max(1, h // 64), 32 * max(1, w // 64)
        h, w = 2 * rh, 2 * rw                                        # exactly half: cv2's 2x2 area route
    else:
        rh, rw = 32 * int(rng.integers(1, 4)), 32 * int(rng.integers(1, 5))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    x4 = det_preprocess(img, (rh, rw), mean, std, dev, swap_rb=True).cpu().numpy()[0]
    rs = cvo.resize_linear_u8(np.ascontiguousarray(img[:, :, ::-1]), (rw, rh)).astype(np.float32)
    exp = (torch.from_numpy(rs).div(255) - torch.tensor(mean)) / torch.tensor(std)
    assert np.array_equal(x4[:, :, :3], exp.numpy()), ("resize", h, w, rh, rw)
    boxes = []
    for _ in range(int(rng.integers(1, 5))):
        cx, cy = rng.uniform(8, w - 8), rng.uniform(8, h - 8)
        bw, bh = rng.uniform(3, max(4.0, w / 2)), rng.uniform(2, max(3.0, h / 3))
        th = rng.uniform(-0.6, 0.6)
        c, s_ = np.cos(th), np.sin(th)
        q = np.array([[-bw / 2, -bh / 2], [bw / 2, -bh / 2], [bw / 2, bh / 2], [-bw / 2, bh / 2]]) @ np.array([[c, s_], [-s_, c]]) + [cx, cy]
        q = np.clip(np.rint(q), [0, 0], [w - 1, h - 1])               # the post-process clamps its boxes to the image (db_postprocess.cpp:303-310)
        if q[:, 0].max() - q[:, 0].min() < 2 or q[:, 1].max() - q[:, 1].min() < 2:
            continue
        boxes.append(q.astype(np.int16))
    if not boxes:
        boxes.append(np.array([[2, 2], [w - 3, 3], [w - 4, h - 3], [3, h - 4]], np.int16))
    buf, metas = warp_crops(torch.from_numpy(img).to(dev), boxes)
    crops = []
    for b, (off, ch, cw) in zip(boxes, metas):
        c = cvo.get_part_img(img, b)
        if c.shape[0] >= 1.5 * c.shape[1]:
            c = np.rot90(c, 1)
        crops.append(np.ascontiguousarray(c))
        assert (ch, cw) == c.shape[:2] and np.array_equal(buf[off:off + ch * cw * 3].cpu().numpy().reshape(ch, cw, 3), c), ("crop", b.tolist())
    x4 = rec_preprocess(buf, metas, [1, 32, 320], dev).cpu().numpy()
    for i, c in enumerate(crops):
        g = cvo.bgr2gray_u8(c)
        rw_ = min(320, int(np.ceil(32 * g.shape[1] / float(g.shape[0]))))
        r = cvo.resize_linear_u8(g, (rw_, 32)).astype(np.float32)
        exp = np.zeros((32, 320), np.float32)
        exp[:, :rw_] = (r / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
        assert np.array_equal(x4[i, :, :, 0], exp), ("rec", c.shape)
