"""GPU box: the three FPN smoothing convs of DBNet-r18 (256 -> 64, 3x3) storing their x8 / x4 / x2 nearest-upsampled result into the
concat buffer (what runs) against a plain store at their own resolution (what a gathering consumer would need)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
N, H4, W4 = 32, 184, 320
conv, bn = nn.Conv2d(256, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64).eval()
pc = ops.PackedConv(conv, bn, dev, relu=True)
fuse = torch.empty((N, H4, W4, 256), device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot_up = tot_plain = 0
for up in (8, 4, 2, 1):
    x = torch.randn(N, H4 // up, W4 // up, 256, device=dev)
    a = t(lambda: ops.conv2d(x, pc, out=fuse, out_up=up, out_coff=0, store=64))
    b = t(lambda: ops.conv2d(x, pc))
    print("level up=%d: upsampled store %.3f ms, plain store %.3f ms" % (up, a, b))
    if up > 1: tot_up += a; tot_plain += b
print("three upsampled levels: %.3f -> %.3f ms" % (tot_up, tot_plain))
