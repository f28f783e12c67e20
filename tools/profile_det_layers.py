#!/usr/bin/env python3
"""Per-launch timing of the DBNet-r18 forward at the bench shape (batch x 736 x 1280): which convolution costs what."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorchocr_amd.modeling import ops
from pytorchocr_amd.modeling.architectures import build_model
from pytorchocr_amd.utils.config import load_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = load_config(os.path.join(os.path.dirname(__file__), "..", "pytorchocr_amd", "configs", "det", sys.argv[2] if len(sys.argv) > 2 else "det_r18_db.yml"))
dev = torch.device("cuda:0")
model = build_model(cfg["Architecture"]).to(dev).eval()
x = torch.randn(N, 736, 1280, 4, device=dev)
x[..., 3] = 0
for _ in range(2):
    model.forward_nhwc4(x)
torch.cuda.synchronize()
ops.PROFILE, ops.PROFILE_LABELS = [], []
iters = 3
for _ in range(iters):
    model.forward_nhwc4(x)
torch.cuda.synchronize()
n = len(ops.PROFILE) // iters
tot = 0.0
for i in range(n):
    ms = sum(ops.PROFILE[k * n + i][0].elapsed_time(ops.PROFILE[k * n + i][1]) for k in range(iters)) / iters
    tot += ms
    print("%2d %-44s %7.3f ms" % (i, ops.PROFILE_LABELS[i], ms), flush=True)
print("total conv launches %d: %.3f ms" % (n, tot))
