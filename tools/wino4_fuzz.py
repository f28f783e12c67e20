"""F(4x4,3x3) Winograd kernel against torch's direct convolution on random shapes (run on the GPU box; PTOCR_WINO4=1 is forced)."""
import os, sys
os.environ["PTOCR_WINO4"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import torch.nn.functional as F
from torch import nn
from pytorchocr_amd.modeling import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
shapes = [(4, 8, 80), (8, 4, 81), (3, 8, 33), (5, 4, 30), (2, 8, 64), (1, 16, 32), (1, 24, 20), (2, 16, 16), (1, 32, 16), (3, 23, 40), (2, 46, 80), (1, 5, 7), (5, 1, 1), (1, 33, 65)]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    if it < len(shapes):
        N, H, W = shapes[it]
    else:
        N = int(rng.integers(1, 7)); H = int(rng.integers(1, 70)); W = int(rng.integers(1, 90))
    cin = int(rng.choice([16, 32, 64, 96, 128])); cout = int(rng.choice([8, 24, 64, 96, 128, 192]))
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=bool(rng.integers(0, 2)))
    x = torch.randn(N, cin, H, W)
    relu = bool(rng.integers(0, 2))
    up = int(rng.choice([1, 1, 1, 2, 4]))
    with torch.no_grad():
        ref = conv(x)
        res = torch.randn_like(ref) if (rng.integers(0, 2) and up == 1) else None
        r2 = ref + res if res is not None else ref
        r2 = F.relu(r2) if relu else r2
        if up > 1:
            r2 = F.interpolate(r2, scale_factor=up, mode="nearest")
    pc = ops.PackedConv(conv, None, dev, relu=relu, cin_pad=cin)
    assert pc.wino4_ok, (cin, cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = {}
    if res is not None:
        rp = torch.zeros(N, H, W, pc.c_tensor); rp[..., :cout] = res.permute(0, 2, 3, 1)
        kw = dict(res=rp.to(dev), res_mode=ops.RES_ADD_PRE_RELU)
    if up > 1:
        kw["out_up"] = up
    y = ops.conv2d(xd, pc, **kw).cpu()[..., :cout].permute(0, 3, 1, 2)
    err = (y - r2).abs().max().item()
    tol = 2e-4 * max(1.0, r2.abs().max().item())
    if y.shape != r2.shape or not (err <= tol):
        bad += 1
        print("MISMATCH", (N, cin, H, W, cout), relu, res is not None, up, err, tol, flush=True)
    elif it < 16:
        print("ok", (N, cin, H, W, cout), "err %.2e" % err, flush=True)
print("fuzz done, mismatches:", bad)
