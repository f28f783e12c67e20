"""GPU box: device time of one post-process call by batch size (1, 2, 8, 32 maps of 736x1280, text-like stress maps): the chain is
latency-bound, so a small batch costs nearly what a large one costs.  usage: post_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_prob_maps
post = build_post_process(bench.DET_POST, {})
base = torch.from_numpy(synth_prob_maps(4, 736, 1280, seed=7)).cuda()
for B in (1, 2, 8, 32):
    maps = base.repeat(B // 4 + 1, 1, 1)[:B, None].contiguous()
    sl = np.array([[736, 1280, 1, 1]] * B)
    post.device_ms_log = []
    for _ in range(24):
        r = post({"maps": maps}, sl)
    print("batch %2d: device ms median %.3f (boxes/img %.1f)" % (B, float(np.median(post.device_ms_log[12:])), sum(len(i["points"]) for i in r) / float(B)))
