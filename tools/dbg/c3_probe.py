"""phase timestamps of the eight-wave 3x3 bf16 kernel (build with PTOCR_EXTRA_HIPCC_FLAGS=-DC3_DBG=16)"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
from pytorchocr_amd.modeling import bf16_path as bp
from pytorchocr_amd import _lib
dev = torch.device("cuda:0")
c3 = bp._C3(nn.Conv2d(96, 24, 3, 1, 1, bias=False), nn.BatchNorm2d(24).eval(), dev, 1)
x = torch.randn(32, 184, 320, 96, device=dev).to(torch.bfloat16)
out = torch.empty((32, 184, 320, 96), dtype=torch.bfloat16, device=dev)
for _ in range(3):
    bp.conv3x3(x, c3, out=out, up=1, coff=72, cstore=24)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 160)()
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libptocr_hip.so"))
print("rc", lib.ptocr_c3_dbg_read(buf))
names = ["top", "gload issued", "mfma done", "sync1", "part written+sync2", "epilogue+sync3", "stores issued", "sync4", "lstore done"]
for it in range(16):
    t = [buf[it * 10 + i] for i in range(9)]
    if t[0] == 0: continue
    print("tile %2d: " % it + "  ".join("%s +%d" % (names[i], t[i] - t[i - 1]) for i in range(1, 9)) + "  | total to next top: %s" % (buf[(it + 1) * 10] - t[0] if it < 15 and buf[(it + 1) * 10] else "-"))
