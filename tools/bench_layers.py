#!/usr/bin/env python3
"""Per-layer timing of the DBNet-r18 convolutions at BASELINE size (736x1280), TFLOP/s vs fp32 MFMA peak."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from pytorchocr_amd.modeling import ops

PEAK = 157.3
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = [  # name, Cin, H, W, Cout, k, s, p
    ("stem7x7", 4, 736, 1280, 64, 7, 2, 3),
    ("l1 3x3 64", 64, 184, 320, 64, 3, 1, 1),
    ("l2 3x3 s2", 64, 184, 320, 128, 3, 2, 1),
    ("l2 3x3 128", 128, 92, 160, 128, 3, 1, 1),
    ("l3 3x3 256", 256, 46, 80, 256, 3, 1, 1),
    ("l4 3x3 512", 512, 23, 40, 512, 3, 1, 1),
    ("in2 1x1 64->256", 64, 184, 320, 256, 1, 1, 0),
    ("out2 3x3 256->64", 256, 184, 320, 64, 3, 1, 1),
    ("in5 1x1 512->256", 512, 23, 40, 256, 1, 1, 0),
]
for name, cin, H, W, cout, k, s, p in LAYERS:
    conv = nn.Conv2d(3 if cin == 4 else cin, cout, k, s, p, bias=False)
    pc = ops.PackedConv(conv, None, dev, relu=True, cin_pad=cin)
    x = torch.randn(N, H, W, cin, device=dev)
    y = ops.conv2d(x, pc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, pc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    real_cin = 3 if cin == 4 else cin
    fl = 2.0 * N * y.shape[1] * y.shape[2] * cout * k * k * real_cin
    print("%-20s %8.3f ms  %7.1f TF/s (%4.1f%% of peak)  out %s" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / PEAK * 100, tuple(y.shape)), flush=True)
