import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
from pytorchocr_amd.modeling import bf16_path as bp
dev = torch.device("cuda:0")
c3 = bp._C3(nn.Conv2d(96, 24, 3, 1, 1, bias=False), nn.BatchNorm2d(24).eval(), dev, 1)
for (n, h, w, up) in [(32, 184, 320, 1), (32, 23, 40, 8)]:
    x = torch.randn(n, h, w, 96, device=dev).to(torch.bfloat16)
    out = torch.empty((n, h * up, w * up, 96), dtype=torch.bfloat16, device=dev)
    f = lambda: bp.conv3x3(x, c3, out=out, up=up, coff=72, cstore=24)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print("conv3x3 bf16 %dx%dx%d up%d: %.1f us" % (n, h, w, up, e0.elapsed_time(e1) * 100), flush=True)
