import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from torch import nn
from pytorchocr_amd.modeling import ops
from pytorchocr_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for (N, cin, H, W, cout) in [(512, 512, 4, 80, 512), (512, 256, 8, 80, 256), (32, 512, 23, 40, 512)]:
    pc = ops.PackedConv(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), None, dev, relu=True)
    x = torch.randn(N, H, W, cin, device=dev)
    ops.conv2d(x, pc); torch.cuda.synchronize()
    buf = torch.zeros(200000 * 4, dtype=torch.int64, device=dev)
    L.ptocr_wino_set_timing_buffer(C.c_void_p(buf.data_ptr()))
    ops.conv2d(x, pc); torch.cuda.synchronize()
    L.ptocr_wino_set_timing_buffer(C.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 4); t = t[t[:, 3] != 0]
    d = np.diff(t, axis=1).astype(np.float64)
    print((N, cin, H, W, cout), "blocks", len(t), "head %.0f main %.0f epi %.0f per chunk %.0f" % (np.median(d[:,0]), np.median(d[:,1]), np.median(d[:,2]), np.median(d[:,1])/(cin//4)))
