"""GPU box: one 3x3 convolution through the F(4x4) Winograd kernel, fp32 and split-operand form, against torch fp32 (CPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch import nn
import torch.nn.functional as F
from pytorchocr_amd.modeling import ops
torch.manual_seed(0)
N, Cin, H, W, Cout = 2, 64, 32, 64, 64
x = torch.randn(N, Cin, H, W)
xd = x.permute(0, 2, 3, 1).contiguous().cuda()
def run(conv, split):
    ops.WINO_SPLIT = split
    pc = ops.PackedConv(conv, None, torch.device("cuda:0"), relu=False, cin_pad=Cin)
    return ops.conv2d(xd, pc).cpu().permute(0, 3, 1, 2)
conv = nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
with torch.no_grad():
    conv.weight.zero_()
    for c in range(Cin):
        conv.weight[c, c, 1, 1] = 1.0
    ref = conv(x)
y = run(conv, True)
err = (y - ref).abs().amax(dim=(0, 2, 3))
print("identity: per-output-channel max err:", ["%.2g" % v for v in err.tolist()])
# one input channel at a time into output channel 5 and 37
for c in range(8):
    with torch.no_grad():
        conv.weight.zero_()
        conv.weight[5, c, 1, 1] = 1.0
        conv.weight[37, c, 0, 2] = 0.5
        ref = conv(x)
    y = run(conv, True)
    e = (y - ref).abs()
    print("cin %d -> cout 5 err %.3g, cout 37 err %.3g, elsewhere %.3g" % (c, e[:, 5].max(), e[:, 37].max(), e[:, [i for i in range(64) if i not in (5, 37)]].max()))
