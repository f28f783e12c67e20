#!/usr/bin/env python3
"""Random-shape sweeps of the specialised conv kernels (stem, pointwise Cin=64, fused conv0+pool, Winograd) against torch fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
bad = 0
def check(name, y, ref, shape):
    global bad
    err = (y - ref).abs().max().item(); tol = 5e-5 * max(1.0, ref.abs().max().item())
    if y.shape != ref.shape or err > tol:
        bad += 1; print("MISMATCH", name, shape, err, tol)
for it in range(60):
    N = int(rng.integers(1, 5)); H = int(rng.integers(1, 150)); W = int(rng.integers(1, 200))
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False); x = torch.randn(N, 3, H, W)
    with torch.no_grad(): ref = F.relu(conv(x))
    pc = ops.PackedConv(conv, None, dev, relu=True, cin_pad=4)
    check("stem", ops.conv2d(ops.nchw_to_nhwc(x.to(dev), 4), pc).cpu().permute(0, 3, 1, 2), ref, (N, H, W))
for it in range(60):
    N = int(rng.integers(1, 5)); H = 2 * int(rng.integers(1, 60)); W = 2 * int(rng.integers(1, 80)); cout = 32 * int(rng.integers(1, 9))
    conv = nn.Conv2d(64, cout, 1, bias=True); x = torch.randn(N, 64, H, W); coarse = torch.randn(N, cout, H // 2, W // 2)
    with torch.no_grad(): ref = F.relu(conv(x)) + F.interpolate(coarse, scale_factor=2, mode="nearest")
    pc = ops.PackedConv(conv, None, dev, relu=True)
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), pc, res=coarse.permute(0, 2, 3, 1).contiguous().to(dev), res_mode=ops.RES_ADD_UP2_POST_RELU)
    check("pw64", y.cpu().permute(0, 3, 1, 2), ref, (N, H, W, cout))
for it in range(40):
    N = int(rng.integers(1, 5)); H = int(rng.integers(2, 70)); W = int(rng.integers(2, 200)); cin = int(rng.choice([1, 3]))
    conv = nn.Conv2d(cin, 64, 3, 1, 1); x = torch.randn(N, cin, H, W)
    with torch.no_grad(): ref = F.max_pool2d(F.relu(conv(x)), 2, 2)
    pc = ops.PackedConv(conv, None, dev, relu=True, cin_pad=4)
    check("conv0pool", ops.conv3x3_relu_pool2(ops.nchw_to_nhwc(x.to(dev), 4), pc).cpu().permute(0, 3, 1, 2), ref, (N, cin, H, W))
print("fuzz done, mismatches:", bad)
