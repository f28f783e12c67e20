"""Time single 3x3/s1 layers under the kernel choice in force (PTOCR_WINO4=0 / 1 / auto): python tools/wino_layer_times.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
SHAPES = [(512, 64, 16, 160, 128), (512, 128, 8, 80, 256), (512, 256, 8, 80, 256), (512, 256, 4, 81, 512), (512, 512, 4, 81, 512),
          (32, 256, 23, 40, 64), (32, 256, 46, 80, 64), (32, 512, 23, 40, 512), (32, 64, 184, 320, 64), (1, 64, 184, 320, 64), (1, 256, 184, 320, 64),
          (1, 512, 23, 40, 512), (1, 256, 46, 80, 256), (8, 64, 184, 248, 64), (8, 256, 184, 248, 64)]
for (N, cin, H, W, cout) in SHAPES:
    pc = ops.PackedConv(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), None, dev, relu=True, cin_pad=cin)
    x = torch.randn(N, H, W, cin, device=dev)
    ops.PROFILE_LABELS, ops.PROFILE = [], []
    ops.conv2d(x, pc)
    lab = ops.PROFILE_LABELS[0].split()[0]
    ops.PROFILE_LABELS = ops.PROFILE = None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d(x, pc)
    e1.record(); torch.cuda.synchronize()
    print("%-9s %4dx%3dx%3dx%3d->%3d  %.3f ms" % (lab, N, H, W, cin, cout, e0.elapsed_time(e1) / 10), flush=True)
