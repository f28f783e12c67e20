import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
def bench(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, cin, H, W, cout) in [(32, 128, 92, 160, 256), (32, 256, 46, 80, 256), (32, 512, 23, 40, 256), (32, 64, 184, 320, 256)]:
    pc = ops.PackedConv(nn.Conv2d(cin, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout).eval(), dev, relu=True)
    x = torch.randn(N, H, W, cin, device=dev)
    res = torch.randn(N, H // 2, W // 2, cout, device=dev)
    t0 = bench(lambda: ops.conv2d(x, pc))
    t1 = bench(lambda: ops.conv2d(x, pc, res=res, res_mode=ops.RES_ADD_UP2_POST_RELU)) if H % 2 == 0 else float("nan")
    gf = 2.0 * N * H * W * cin * cout / 1e9
    mb = (x.numel() + N * H * W * cout) * 4 / 1e6
    print("%dx%dx%dx%d->%d  plain %.3f ms (%.0f TF/s, %.2f TB/s)   +up2 residual %.3f ms" % (N, H, W, cin, cout, t0, gf / t0, mb / t0 / 1e3, t1), flush=True)
