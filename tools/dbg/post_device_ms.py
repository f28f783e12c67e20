"""GPU box: device time of the stand-alone post-process call (HIP events on its stream, nothing else on the chip) on the bench's 32
text-like stress maps (synth_prob_maps, seed 7: what bench.py reports as ms_per_call_alone_stress_maps).  usage: post_device_ms.py [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 80
post = build_post_process(bench.DET_POST, {})
maps = torch.from_numpy(synth_prob_maps(32, 736, 1280, seed=7)).cuda()[:, None].contiguous()
sl = np.array([[736, 1280, 1, 1]] * 32)
post.device_ms_log = []
for _ in range(calls):
    r = post({"maps": maps}, sl)
ms = np.array(post.device_ms_log[calls // 4:])
print("stress maps: device ms per call median %.4f  min %.4f  p90 %.4f  (boxes/img %.1f)  frac of 8 TB/s on 18 B/pixel: %.4f" % (
    float(np.median(ms)), float(ms.min()), float(np.percentile(ms, 90)), sum(len(i["points"]) for i in r) / 32.0,
    18.0 * 32 * 736 * 1280 / (float(np.median(ms)) * 1e-3) / 8e12))
