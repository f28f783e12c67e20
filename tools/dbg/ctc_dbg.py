import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorchocr_amd import _lib
from pytorchocr_amd.modeling import ops
torch.manual_seed(0)
M, K, Cc, Np = 5, 64, 200, 256
w = torch.zeros(Np, K); w[:Cc] = torch.randn(Cc, K) * 0.3
b = torch.zeros(Np); b[:Cc] = torch.randn(Cc)
x = torch.randn(M, K)
xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
idx = torch.empty(M, dtype=torch.int32, device="cuda"); prob = torch.empty(M, device="cuda")
work = torch.zeros((M, Np // 64, 4), device="cuda")
_lib.check(_lib.lib().ptocr_linear_ctc_greedy_f32(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), M, K, Np, Cc, _lib.ptr(work), _lib.ptr(idx), _lib.ptr(prob), _lib.cur_stream()))
torch.cuda.synchronize()
lg = x @ w.t() + b
print("work max", work[..., 0].cpu())
print("work idx", work[..., 1].cpu().view(torch.int32))
print("work sum", work[..., 2].cpu())
for t in range(4):
    sl = lg[:, 64 * t:64 * (t + 1)].clone()
    if 64 * t + 64 > Cc: sl[:, Cc - 64 * t:] = -1e30
    print("ref tile", t, sl.max(1).values, sl.argmax(1) + 64 * t)
print("idx", idx.cpu(), "ref", lg[:, :Cc].argmax(1))
print("prob", prob.cpu(), torch.softmax(lg[:, :Cc], 1).max(1).values)
