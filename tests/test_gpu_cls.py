"""Direction classifier (the cls branch of run_ocr) on the HIP engine (-m gpu): the recognition-style MobileNetV3
+ ClsHead against outputs of the reference itself and the torch-fp32 oracle, `Clser`, and the three OCRer paths with a classifier
that really turns some of the lines."""
import os

import numpy as np
import pytest
import torch

from pytorchocr_amd.utils.synth import synth_images, synth_scene_images, synth_state_dict

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorchocr_amd", "configs", "cls", "cls_mbv3small.yml")


def _weights(contract, g):
    from test_oracle_model import cls_state_dict
    return cls_state_dict(contract, g)


@pytest.fixture(scope="module")
def gold(gold_dir):
    return np.load(os.path.join(gold_dir, "cls_mbv3s_4x3x48x192.npz"))


@pytest.fixture(scope="module")
def model(contract, gold):
    from pytorchocr_amd.modeling.architectures import build_model
    from pytorchocr_amd.utils.config import load_config
    m = build_model(load_config(CFG)["Architecture"]).to("cuda:0").eval()
    sd = _weights(contract, gold)
    assert set(sd) == set(m.state_dict())                      # the reference's parameter names
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m, sd


def test_cls_model_against_reference_golden(model, gold):
    m, _ = model
    x = torch.from_numpy(synth_images(4, 3, 48, 192, seed=int(gold["seed"]))).cuda()
    m.return_all_feats = True
    try:
        with torch.no_grad():
            y = m(x)
    finally:
        m.return_all_feats = False
    f = y["backbone_out"].cpu().numpy()
    assert f.shape == gold["backbone_out"].shape
    assert np.abs(f - gold["backbone_out"]).max() <= 1e-4 * max(1.0, np.abs(gold["backbone_out"]).max())
    assert np.abs(y["head_out"].cpu().numpy() - gold["probs"]).max() <= 1e-4
    with torch.no_grad():
        p = m(x).cpu().numpy()
    assert p.shape == (4, 2) and np.abs(p - gold["probs"]).max() <= 1e-4 and np.abs(p.sum(1) - 1).max() <= 1e-6


@pytest.mark.parametrize("shape", [(1, 48, 192), (3, 48, 97), (5, 32, 100), (2, 48, 320), (130, 48, 192)])
def test_cls_model_other_shapes_against_oracle(model, shape):
    """odd widths (the 2x2 pool drops the last column), other heights, a batch larger than one launch's usual size; and the
    result of a line does not depend on its batch"""
    from oracle import model_oracle
    m, sd = model
    n, h, w = shape
    x = torch.from_numpy(synth_images(n, 3, h, w, seed=40 + n))
    with torch.no_grad():
        p = m(x.cuda()).cpu().numpy()
        p1 = m(x[:1].cuda()).cpu().numpy()
    ref = model_oracle.cls_mbv3_small_forward(sd, x[:8]).numpy()
    assert np.abs(p[:8] - ref).max() <= 1e-4
    assert np.array_equal(p[:1], p1)


def test_cls_head_module_boundary(model):
    """ClsHead.forward on the reference's tensor at that boundary (pooled NCHW features)"""
    m, sd = model
    f = torch.from_numpy(synth_images(3, 200, 2, 24, seed=3)[:, :192].copy())
    with torch.no_grad():
        p = m.head(f.cuda()).cpu().numpy()
    logits = f.mean((2, 3)).numpy() @ sd["head.fc.weight"].T + sd["head.fc.bias"]
    e = np.exp(logits - logits.max(1, keepdims=True))
    assert np.abs(p - e / e.sum(1, keepdims=True)).max() <= 1e-5


def test_clser_entry_point(model, contract, gold):
    """Clser.run / run_batch (infer_cls.py:82-101): host ClsResizeImg + model + ClsPostProcess == the oracle on the same tensor"""
    from oracle import model_oracle
    from pytorchocr_amd.deploy.infer_cls import Clser
    from pytorchocr_amd.utils.config import load_config
    c = Clser(load_config(CFG), None, 0)
    sd = model[1]
    c.clser.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    lines = [synth_scene_images(1, 40, 150, seed=1)[0], synth_scene_images(1, 64, 500, seed=2)[0], synth_scene_images(1, 48, 60, seed=3)[0]]
    got = c.run_batch(lines)
    x = torch.stack([c._prep(l) for l in lines])
    assert x.shape == (3, 3, 48, 192) and float(x[2, :, :, 60:].abs().max()) == 0.0          # right zero padding
    ref = model_oracle.cls_mbv3_small_forward(sd, x).numpy()
    for (lab, pr), r in zip(got, ref):
        assert lab == ["0", "180"][int(r.argmax())] and abs(pr - round(float(r.max()), 2)) <= 0.011
    assert c.run(lines[1]) == got[1]


def test_ocr_with_direction_classifier(model):
    """run_ocr with a classifier (run_ocr.py:192-211).  The fc bias is set into the widest gap around the median logit gap of this image's
    lines, so between a quarter and three quarters of them are called "180" and turned: the host path (numpy rotation of the crop), the one-image GPU path and the batched
    GPU path (rotation = a flag of the recognition pre-process) give the same [box, text, prob] lists, and they differ from the
    run without a classifier exactly where a line was turned."""
    from pytorchocr_amd.deploy.bench_ocr import make_ocrer
    from pytorchocr_amd.deploy.infer_cls import Clser
    from pytorchocr_amd.utils.config import load_config
    from pytorchocr_amd.utils.warp import get_part_img
    ocr = make_ocrer(0)
    img = synth_scene_images(1, 480, 640, seed=31)[0]
    plain = ocr.run_gpu(img)
    assert len(plain) > 20
    cls = Clser(load_config(CFG), None, 0)
    sd = {k: torch.from_numpy(v) for k, v in model[1].items()}
    cls.clser.load_state_dict(sd, strict=True)
    crops = []
    for box, _, _ in plain:
        part = get_part_img(img, box)
        crops.append(np.ascontiguousarray(np.rot90(part, 1) if part.shape[0] >= 1.5 * part.shape[1] else part))
    with torch.no_grad():
        p = cls.clser(torch.stack([cls._prep(c) for c in crops]).cuda()).cpu().numpy().astype(np.float64)
    gap = np.log(p[:, 1] / p[:, 0])
    order = np.sort(gap)
    q = len(order) // 4                                             # the widest gap in the middle half of the sorted logit gaps
    mid = q + 1 + int(np.argmax(np.diff(order[q:len(order) - q])))
    assert order[mid] - order[mid - 1] > 1e-4, "no clear margin between the two groups"
    sd["head.fc.bias"] = torch.tensor([0.0, -0.5 * (order[mid] + order[mid - 1])], dtype=torch.float32)
    cls.clser.load_state_dict(sd, strict=True)
    turned = gap > 0.5 * (order[mid] + order[mid - 1])
    assert turned.any() and not turned.all()
    ocr.cls = cls
    try:
        a = ocr.run_gpu(img)
        b = ocr.run_batch([img, img], rec_batch=16)
        ocr.gpu_preprocess = False
        c = ocr.run(img)
    finally:
        ocr.cls, ocr.gpu_preprocess = None, True
    assert len(a) == len(plain) == len(c)
    for k, (ra, rb0, rb1, rc, rp) in enumerate(zip(a, b[0], b[1], c, plain)):
        for r in (rb0, rb1, rc):
            assert np.array_equal(ra[0], r[0]) and ra[1] == r[1] and (ra[2] == r[2] or (np.isnan(ra[2]) and np.isnan(r[2])))
        if not turned[k]:
            assert ra[1] == rp[1]
    assert any(ra[1] != rp[1] for ra, rp, t in zip(a, plain, turned) if t)
    # what "turned" means: the recogniser saw the crop rotated by 180 degrees
    k = int(np.nonzero(turned)[0][0])
    assert ocr.rec.run_batch([np.ascontiguousarray(crops[k][::-1, ::-1])])[0][0] == a[k][1]
