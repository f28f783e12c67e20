"""Wall time of the stages of OCRer.run_batch (synchronised after each): python tools/dbg/ocr_stages.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pytorchocr_amd.deploy.bench_ocr import make_ocrer
from pytorchocr_amd.utils.synth import synth_scene_images
from pytorchocr_amd.data.gpu_preprocess import det_preprocess_batch, rec_preprocess, warp_crops_batch
from pytorchocr_amd.data.imaug import RecResizeImg
from pytorchocr_amd.utils.utility import sort_boxes
dev = torch.device("cuda:0")
ocr = make_ocrer(0, 0, 1)
base = synth_scene_images(32, 960, 1280, seed=100)
imgs = torch.from_numpy(base).to(dev).repeat(2, 1, 1, 1).contiguous()
for _ in range(2):
    ocr.run_batch(imgs)
torch.cuda.synchronize()
T = {}
def lap(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return time.perf_counter()
with torch.no_grad():
  for _ in range(3):
    t = time.perf_counter()
    rs, nm = ocr.det._gpu_ops()
    shape = [o for o in ocr.rec.rec_ops if isinstance(o, RecResizeImg)][0].image_shape
    rh, rw = rs.target_size(960, 1280)
    x4 = det_preprocess_batch(imgs, (rh, rw), nm.mean, nm.std, swap_rb=ocr.det.det_img_mode == "RGB"); t = lap("det pre-process", t)
    maps = ocr.det.deter.forward_nhwc4(x4); t = lap("det forward", t)
    shapes = np.array([[960, 1280, rh / 960.0, rw / 1280.0]] * 64)
    res = ocr.det.det_post_process_class(maps, shapes); t = lap("det post-process", t)
    boxes = [sort_boxes(r["points"]) for r in res]; t = lap("sort_boxes", t)
    buf, metas = warp_crops_batch(imgs, boxes); t = lap("warp crops", t)
    flat = [m for per in metas for m in per]
    x_rec = rec_preprocess(buf, flat, shape, dev, flip=None); t = lap("rec pre-process", t)
    texts, pend = [], None
    for c0 in range(0, int(x_rec.shape[0]), 512):
        fut = ocr.rec.rec_post_process_class.submit(ocr.rec.recer.forward_greedy_nhwc4(x_rec[c0:c0 + 512]))
        if pend is not None: texts += pend.result()
        pend = fut
    texts += pend.result(); t = lap("CRNN + decode (%d lines)" % len(texts), t)
    probs = list(np.round(np.array([p for _, p in texts], dtype=np.float64), 2))
    words = [tt for tt, _ in texts]
    pos, out = 0, []
    for bx, per in zip(boxes, metas):
        kept = [b for b, m in zip(bx, per) if m is not None] if None in per else bx
        out.append([[b, tt, p] for b, tt, p in zip(kept, words[pos:pos + len(kept)], probs[pos:pos + len(kept)])])
        pos += len(kept)
    t = lap("regroup", t)
for k, v in T.items():
    print("%-34s %8.2f ms" % (k, v / 3))
print("sum %.2f ms" % (sum(T.values()) / 3))
