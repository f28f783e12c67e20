"""s_memtime probes of the F(4x4,3x3) Winograd kernel: head / main loop / epilogue cycles per workgroup (100 MHz counter on
gfx950 s_memtime is in its own clock domain: the numbers are only compared with each other and with tools/wino_timing.py)."""
import os, sys, ctypes as C
os.environ["PTOCR_WINO4"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from torch import nn
from pytorchocr_amd.modeling import ops
from pytorchocr_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for (N, cin, H, W, cout) in [(32, 64, 184, 320, 64), (32, 256, 184, 320, 64), (32, 512, 23, 40, 512)]:
    pc = ops.PackedConv(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), None, dev, relu=True)
    x = torch.randn(N, H, W, cin, device=dev)
    ops.conv2d(x, pc); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d(x, pc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nblk = 200000
    buf = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    L.ptocr_wino4_set_timing_buffer(C.c_void_p(buf.data_ptr()))
    ops.conv2d(x, pc); torch.cuda.synchronize()
    L.ptocr_wino4_set_timing_buffer(C.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 3] != 0]
    d = np.diff(t, axis=1).astype(np.float64)
    print("shape", (N, cin, H, W, cout), "blocks", len(t), "%.3f ms" % ms)
    print("  head %.0f  main %.0f  epilogue %.0f  total %.0f cycles (median)" % (np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(t[:, 3] - t[:, 0])))
    print("  per chunk %.0f cycles" % (np.median(d[:, 1]) / (cin // 4)))
    span = (t[:, 3].max() - t[:, 0].min())
    print("  kernel span %.0f cycles; sum of block totals / 256 CUs = %.0f" % (span, (t[:, 3] - t[:, 0]).sum() / 256), flush=True)
