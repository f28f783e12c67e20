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
pc = ops.PackedConv(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64).eval(), dev, relu=True, cin_pad=4)
x = torch.randn(32, 3, 736, 1280, device=dev)
x4 = ops.nchw_to_nhwc(x, 4)
print("stem+pool nchw  %.3f ms" % bench(lambda: ops.stem_relu_pool(None, x, pc)))
print("stem+pool nhwc4 %.3f ms" % bench(lambda: ops.stem_relu_pool(x4, None, pc)))
print("stem nchw       %.3f ms" % bench(lambda: ops.stem_from_nchw(x, pc)))
print("stem nhwc4      %.3f ms" % bench(lambda: ops.conv2d(x4, pc)))
y = ops.conv2d(x4, pc)
print("maxpool         %.3f ms" % bench(lambda: ops.maxpool2d(y, 3, 2, 1)))
