"""BASELINE.json configs[4] (run_ocr: DBNet++ r18 detect -> crop -> CRNN) on the HIP engine (-m gpu).

The detector carries the hand-made brightness checkpoint (utils/synth.py: its map is a soft threshold of the image
brightness) and the images are text-like scenes, so the boxes are REAL detections (nothing monkeypatched) and the crops, the
recognition batches and the regrouping all do real work.  The CRNN has random-init weights: its texts are gibberish but
deterministic, which is all an equality test needs."""
import os

import numpy as np
import pytest
import torch

from pytorchocr_amd.utils.synth import synth_brightness_detector_state_dict, synth_images, synth_scene_images, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ocr():
    from pytorchocr_amd.deploy.bench_ocr import make_ocrer
    return make_ocrer(0)


def _same(a, b):
    assert len(a) == len(b)
    for (bx0, t0, p0), (bx1, t1, p1) in zip(a, b):
        assert np.array_equal(bx0, bx1) and t0 == t1
        assert p0 == p1 or (np.isnan(p0) and np.isnan(p1))


def test_dbpp_full_config4_size_against_oracle(contract):
    """DBNet++ r18 at the configs[4] network size 736x992 (one image): maps within 1e-4 of the torch-fp32 oracle, for the
    random synthetic weights and for the brightness checkpoint."""
    from oracle import model_oracle
    from pytorchocr_amd.modeling.architectures import build_model
    cfg = dict(model_type="det", algorithm="DB", Transform=None, Backbone=dict(name="ResNet", layers=18, pretrained=False),
               Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=True, attention_type="scale_channel_spatial"),
               Head=dict(name="DBHead", k=50))
    m = build_model(cfg).to("cuda:0").eval()
    xs = synth_images(1, 3, 736, 992, seed=5)
    for sd in (synth_state_dict(contract["detpp_r18_db"]), synth_brightness_detector_state_dict(contract["detpp_r18_db"], use_asf=True)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            y = m(torch.from_numpy(xs).cuda())["maps"].cpu().numpy()
        ref = model_oracle.dbnet_r18_forward(sd, torch.from_numpy(xs))["maps"].numpy()
        assert y.shape == (1, 1, 736, 992) and np.abs(y - ref).max() <= 1e-4


def test_run_batch_equals_per_image_run(ocr):
    """images of two sizes, one of them without any text: batched pipeline == per-image pipeline, box for box"""
    imgs = list(synth_scene_images(3, 240, 320, seed=7)) + list(synth_scene_images(2, 192, 416, seed=8))
    imgs.insert(2, np.zeros((240, 320, 3), np.uint8))
    stats = {}
    got = ocr.run_batch(imgs, rec_batch=16, stats=stats)             # small chunks: several CRNN batches per group
    assert len(got) == len(imgs) and got[2] == []
    n = 0
    for img, g in zip(imgs, got):
        _same(g, ocr.run_gpu(img))
        n += len(g)
    assert n > 30 and stats["boxes"] >= n and stats["lines"] == n
    # sub-groups of two images: the detector of sub-group i+1 is queued before the host stages of sub-group i (software pipeline)
    for a, b in zip(ocr.run_batch(imgs, rec_batch=16, det_batch=2), got):
        _same(a, b)
    # a device-resident stack gives the same as the list of arrays
    stack = torch.from_numpy(np.stack(imgs[:2])).cuda()
    for a, b in zip(ocr.run_batch(stack), got[:2]):
        _same(a, b)


def test_config4_source_size_finds_the_text_lines(ocr):
    """one 1280x960 source (network input 736x992): the detector finds the scene's bars, every box is recognised, and the
    result equals the unbatched run; the boxes also equal the post-process oracle applied to the oracle's own map"""
    from oracle import dbpost, model_oracle
    from pytorchocr_amd.data.imaug import resize_bilinear
    from pytorchocr_amd.utils.utility import sort_boxes
    img = synth_scene_images(1, 960, 1280, seed=11)[0]
    got = ocr.run_batch([img])[0]
    assert len(got) > 60
    _same(got, ocr.run_gpu(img))
    # oracle pipeline on the host: resize (host operator), normalise, torch-fp32 DB++ forward, C post-process, sort_boxes
    rs = resize_bilinear(img[:, :, ::-1], (992, 736)).astype(np.float32) / 255.0
    x = ((rs - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)).transpose(2, 0, 1)[None]
    sd = {k: v.cpu() for k, v in ocr.det.deter.state_dict().items()}
    ref = model_oracle.dbnet_r18_forward(sd, torch.from_numpy(np.ascontiguousarray(x, np.float32)))["maps"].numpy()[0, 0]
    exp = dbpost.boxes_from_bitmap(ref, dbpost.binarize(ref, 0.3), 0.5, 1.7, 1280, 960)
    exp = sort_boxes(exp.astype(np.int16)) if len(exp) else []
    assert len(exp) == len(got)
    same = sum(np.array_equal(np.asarray(e), g[0]) for e, g in zip(exp, got))
    assert same >= len(exp) - 2, "boxes differ from the oracle pipeline (%d of %d equal)" % (same, len(exp))


def test_eval_loop_on_the_gpu_pipeline(ocr):
    """pytorchocr_amd.eval (reference tools/program.py:421-473) end to end on the HIP path: detection with the brightness checkpoint
    scored against its own boxes gives hmean 1, against shifted ground truth less; recognition scored against its own texts gives
    accuracy 1 (the metric arithmetic itself is pinned in tests/test_metrics.py)."""
    from pytorchocr_amd.eval import eval as run_eval
    from pytorchocr_amd.metrics import build_metric
    dev = torch.device("cuda:0")
    det, rec = ocr.det, ocr.rec
    imgs = synth_scene_images(2, 240, 320, seed=21)
    batches = []
    for img in imgs:
        x, shape = det._prep(img)
        boxes = det.run(img)
        polys = np.stack(boxes).astype(np.float32)[None]
        batches.append([x[None], shape[None], polys, np.zeros((1, len(boxes)), bool)])
    m = run_eval(det.deter, dev, batches, det.det_post_process_class, build_metric(dict(name="DetMetric")), model_type="det")
    assert m["hmean"] == 1.0 and m["precision"] == 1.0 and m["recall"] == 1.0 and m["fps"] > 0
    far = [[b[0], b[1], b[2] + 200.0, b[3]] for b in batches]
    m2 = run_eval(det.deter, dev, far, det.det_post_process_class, build_metric(dict(name="DetMetric")), model_type="det")
    assert m2["hmean"] < 0.2
    # recognition: labels = the model's own greedy texts, encoded with CTCLabelEncode and decoded by the post-process
    from pytorchocr_amd.data.label_ops import CTCLabelEncode
    from pytorchocr_amd.utils.synth import synth_text_lines
    x = torch.from_numpy(synth_text_lines(6, 32, 320, seed=5))
    with torch.no_grad():
        texts = [t for t, _ in rec.rec_post_process_class(rec.recer(x.cuda()))]
    enc = CTCLabelEncode(max_text_length=100, character_dict_path=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "pytorchocr_amd", "utils", "char_dict_6623.txt"))
    keep = [i for i, t in enumerate(texts) if 0 < len(t) <= 100]
    labels = np.stack([enc({"label": texts[i]})["label"] for i in keep])
    rb = [x[keep], labels]
    m3 = run_eval(rec.recer, dev, [rb], rec.rec_post_process_class, build_metric(dict(name="RecMetric")), model_type="rec")
    assert len(keep) >= 3 and m3["acc"] > 0.999 and m3["norm_edit_dis"] > 0.999
