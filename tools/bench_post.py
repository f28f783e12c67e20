#!/usr/bin/env python3
"""Stand-alone timing of the GPU DB post-process on text-like maps (32 x 736 x 1280, ~140 boxes per image), plus the
same stage fed with the model-like noise maps: ms per call, images/s and GB/s on the SURVEY 8d accounting (18 B/pixel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps

B, H, W = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 736, 1280
dev = torch.device("cuda:0")
post = build_post_process(dict(name="DBPostProcess", thresh=0.3, box_thresh=0.5, max_candidates=1000, unclip_ratio=1.7,
                               score_mode="poly", cpp_speedup=True, out_polygon=False), dict(use_gpu=True))
maps = torch.from_numpy(synth_prob_maps(min(4, B), H, W, seed=7)).to(dev).repeat(max(B // 4, 1), 1, 1)[:B, None].contiguous()
shape_list = np.array([[H, W, 1.0, 1.0]] * B)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(2):
    res = post({"maps": maps}, shape_list)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    res = post({"maps": maps}, shape_list)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
nb = sum(len(r["points"]) for r in res) / B
print("post-process: %.3f ms per call of %d images (%.0f boxes/image) = %.0f images/s; %.1f GB/s on 18 B/pixel"
      % (ms, B, nb, B / ms * 1e3, 18.0 * H * W * B / ms / 1e6))
