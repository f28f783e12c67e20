import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import torch.nn.functional as F
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
bad = 0
for it in range(150):
    N = int(rng.integers(1, 10)); H = int(rng.integers(1, 70)); W = int(rng.integers(1, 90))
    cin = int(rng.choice([32, 64, 96, 128])); cout = int(rng.choice([8, 24, 64, 96, 128, 192]))
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=bool(rng.integers(0, 2)))
    x = torch.randn(N, cin, H, W)
    relu = bool(rng.integers(0, 2))
    with torch.no_grad():
        ref = conv(x)
        res = torch.randn_like(ref) if rng.integers(0, 2) else None
        r2 = ref + res if res is not None else ref
        r2 = F.relu(r2) if relu else r2
    pc = ops.PackedConv(conv, None, dev, relu=relu)
    assert pc.wino_u is not None, (cin, cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = {}
    if res is not None:
        rp = torch.zeros(N, H, W, pc.c_tensor); rp[..., :cout] = res.permute(0, 2, 3, 1)
        kw = dict(res=rp.to(dev), res_mode=ops.RES_ADD_PRE_RELU)
    y = ops.conv2d(xd, pc, **kw).cpu()[..., :cout].permute(0, 3, 1, 2)
    err = (y - r2).abs().max().item()
    tol = 5e-5 * max(1.0, r2.abs().max().item())
    if err > tol:
        bad += 1
        print("MISMATCH", (N, cin, H, W, cout), relu, res is not None, err, tol)
print("fuzz done, mismatches:", bad)
