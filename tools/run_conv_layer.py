#!/usr/bin/env python3
"""Run one 3x3 layer a few times (for rocprofv3 --pmc passes): run_conv_layer.py N Cin H W Cout [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from pytorchocr_amd.modeling import ops

N, cin, H, W, cout = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
pc = ops.PackedConv(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), None, dev, relu=True)
x = torch.randn(N, H, W, cin, device=dev)
for _ in range(iters):
    ops.conv2d(x, pc)
torch.cuda.synchronize()
print("done")
