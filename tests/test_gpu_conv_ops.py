"""HIP conv / pool / layout kernels against plain torch fp32 CPU ops on the same seeded inputs (-m gpu)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * 2 - 1


CASES = [
    # N, Cin, H, W, Cout, k, stride, pad
    (2, 64, 20, 36, 64, 3, 1, 1),
    (1, 64, 23, 40, 128, 3, 2, 1),
    (2, 128, 12, 20, 128, 3, 1, 1),
    (1, 64, 16, 24, 128, 1, 2, 0),
    (1, 512, 5, 7, 256, 1, 1, 0),
    (3, 256, 9, 11, 64, 3, 1, 1),
    (1, 256, 6, 10, 512, 3, 2, 1),
    (1, 512, 2, 40, 512, 2, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bn_relu_matches_torch(case):
    from pytorchocr_amd.modeling import ops
    N, Cin, H, W, Cout, k, s, p = case
    dev = _dev()
    conv = nn.Conv2d(Cin, Cout, k, s, p, bias=True)
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, k, k, seed=1) * (3.0 / (Cin * k * k)) ** 0.5)
        conv.bias.copy_(_rand(Cout, seed=2) * 0.1)
        bn.weight.copy_(_rand(Cout, seed=3) * 0.4 + 1); bn.bias.copy_(_rand(Cout, seed=4) * 0.2)
        bn.running_mean.copy_(_rand(Cout, seed=5) * 0.2); bn.running_var.copy_(_rand(Cout, seed=6) * 0.5 + 1)
    x = _rand(N, Cin, H, W, seed=7)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
    pc = ops.PackedConv(conv, bn, dev, relu=True)
    y = ops.conv2d(_nhwc(x).to(dev), pc)
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 70, 130), (3, 33, 47), (2, 160, 608)])
def test_stem_7x7_cin3(shape):
    """the stem kernel (K without the padding channel, persistent tiles) and, via PTOCR_STEM_KERNEL semantics, the generic
    kernel: both against torch fp32; odd sizes exercise partial tiles and the zero border"""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    N, H, W = shape
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(64, 3, 7, 7, seed=1) * 0.15)
        bn.running_mean.copy_(_rand(64, seed=5) * 0.2); bn.running_var.copy_(_rand(64, seed=6) * 0.5 + 1)
    x = _rand(N, 3, H, W, seed=7)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
    pc = ops.PackedConv(conv, bn, dev, relu=True, cin_pad=4)
    assert pc.stem_w is not None
    x4 = ops.nchw_to_nhwc(x.to(dev), 4)
    assert torch.equal(x4.cpu()[..., :3], _nhwc(x)) and float(x4[..., 3].abs().max()) == 0.0
    x4[..., 3] = 7.0                                     # the stem kernel must ignore the padding channel
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    for use in (True, False):
        ops.USE_STEM_KERNEL = use
        try:
            if not use:
                x4[..., 3] = 0.0                             # the generic kernel multiplies it by zero weights
            y = ops.conv2d(x4, pc)
        finally:
            ops.USE_STEM_KERNEL = True
        got = y.cpu().permute(0, 3, 1, 2)
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= tol, use
    # the same kernel reading the NCHW model input directly: bit-identical to the NHWC4 form
    y_nchw = ops.stem_from_nchw(x.to(dev), pc)
    x4[..., 3] = 0.0
    assert torch.equal(y_nchw, ops.conv2d(x4, pc))


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 70, 130), (3, 33, 47), (2, 160, 608), (1, 736, 1280), (5, 9, 250), (1, 129, 61)])
def test_stem_with_fused_max_pool(shape):
    """conv 7x7/s2 + BN + ReLU + MaxPool2d(3, 2, 1) in one kernel == the stem kernel followed by the pool kernel, bit for bit (strips
    of 30 columns with a carried row: sizes that are no multiple of anything, one-strip and many-strip images, 1 and 23 tile rows),
    from the NHWC4 tensor and from the NCHW input, and within tolerance of torch"""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    N, H, W = shape
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(64, 3, 7, 7, seed=1) * 0.15)
        bn.running_mean.copy_(_rand(64, seed=5) * 0.2); bn.running_var.copy_(_rand(64, seed=6) * 0.5 + 1)
    x = _rand(N, 3, H, W, seed=11)
    pc = ops.PackedConv(conv, bn, dev, relu=True, cin_pad=4)
    xd = x.to(dev)
    x4 = ops.nchw_to_nhwc(xd, 4)
    two = ops.maxpool2d(ops.conv2d(x4, pc), 3, 2, 1)
    one = ops.stem_relu_pool(x4, None, pc)
    assert one is not None and one.shape == two.shape and torch.equal(one, two)
    assert torch.equal(ops.stem_relu_pool(None, xd, pc), two)
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(bn(conv(x))), 3, 2, 1)
    got = one.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert ops.stem_relu_pool(x4, None, ops.PackedConv(conv, bn, dev, relu=False, cin_pad=4)) is None      # needs the ReLU (0 as pool padding)


def test_residual_and_upsample_epilogues():
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    conv = nn.Conv2d(64, 64, 3, 1, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(_rand(64, 64, 3, 3, seed=1) * 0.07)
    x = _rand(2, 64, 12, 20, seed=2)
    res = _rand(2, 64, 12, 20, seed=3)
    small = _rand(2, 64, 6, 10, seed=4)
    with torch.no_grad():
        ref1 = F.relu(conv(x) + res)
        ref2 = F.relu(conv(x)) + F.interpolate(small, scale_factor=2, mode="nearest")
        ref3 = F.interpolate(F.relu(conv(x)), scale_factor=4, mode="nearest")
    pc = ops.PackedConv(conv, None, dev, relu=True)
    xd = _nhwc(x).to(dev)
    y1 = ops.conv2d(xd, pc, res=_nhwc(res).to(dev), res_mode=ops.RES_ADD_PRE_RELU)
    y2 = ops.conv2d(xd, pc, res=_nhwc(small).to(dev), res_mode=ops.RES_ADD_UP2_POST_RELU)
    big = torch.full((2, 48, 80, 256), -7.0, device=dev)
    ops.conv2d(xd, pc, out=big, out_up=4, out_coff=128)
    for got, ref in ((y1, ref1), (y2, ref2), (big[..., 128:192], ref3)):
        assert (got.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5
    assert float((big[..., :128] + 7).abs().max()) == 0 and float((big[..., 192:] + 7).abs().max()) == 0


def test_convtranspose_and_head_tail():
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    t3 = nn.ConvTranspose2d(64, 64, 2, 2)
    bn = nn.BatchNorm2d(64).eval()
    t6 = nn.ConvTranspose2d(64, 1, 2, 2)
    with torch.no_grad():
        t3.weight.copy_(_rand(64, 64, 2, 2, seed=1) * 0.2); t3.bias.copy_(_rand(64, seed=2) * 0.1)
        bn.running_mean.copy_(_rand(64, seed=5) * 0.2); bn.running_var.copy_(_rand(64, seed=6) * 0.5 + 1)
        t6.weight.copy_(_rand(64, 1, 2, 2, seed=3) * 0.3); t6.bias.copy_(_rand(1, seed=4))
    x = _rand(2, 64, 9, 13, seed=7)
    with torch.no_grad():
        mid = F.relu(bn(t3(x)))
        ref = torch.sigmoid(t6(mid))
    pt = ops.PackedConvT2x2(t3, bn, dev, relu=True)
    y = ops.conv2d(_nhwc(x).to(dev), pt)
    assert (y.cpu().permute(0, 3, 1, 2) - mid).abs().max().item() <= 2e-5
    w4 = t6.weight.detach()[:, 0].permute(1, 2, 0).reshape(4, -1).contiguous().to(dev)
    maps = ops.convt2x2_sigmoid(y, w4, float(t6.bias.detach()[0]))
    assert maps.shape == ref.shape
    assert (maps.cpu() - ref).abs().max().item() <= 1e-5


def test_maxpool_variants():
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    x = _rand(2, 64, 17, 23, seed=1)
    for k, s, p in ((3, 2, 1), (2, 2, 0), ((2, 2), (2, 1), (0, 1))):
        ref = F.max_pool2d(x, k, s, p)
        got = ops.maxpool2d(_nhwc(x).to(dev), k, s, p).cpu().permute(0, 3, 1, 2)
        assert torch.equal(got, ref)


def test_layout_roundtrip():
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    x = _rand(2, 70, 9, 13, seed=1).to(dev)
    y = ops.nchw_to_nhwc(x, 72)
    assert torch.equal(y[..., :70].permute(0, 3, 1, 2), x)
    z = ops.nhwc_to_nchw(y)
    assert torch.equal(z[:, :70], x)


def test_padded_channels_hardswish_and_masked_store():
    """MobileNetV3-style conv: Cin 72 (tensor 96), Cout 24 (tensor 32), Hardswish, residual with its own channel stride."""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    conv = nn.Conv2d(72, 24, 1, bias=False)
    bn = nn.BatchNorm2d(24, eps=1e-3).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(24, 72, 1, 1, seed=1) * 0.3)
        bn.running_mean.copy_(_rand(24, seed=5) * 0.2); bn.running_var.copy_(_rand(24, seed=6) * 0.5 + 1)
    x = _rand(2, 72, 11, 13, seed=7) * 3
    res = _rand(2, 24, 11, 13, seed=8)
    with torch.no_grad():
        ref_hs = F.hardswish(bn(conv(x)))
        ref_res = bn(conv(x)) + res
    xp = torch.zeros(2, 11, 13, 96); xp[..., :72] = _nhwc(x)
    rp = torch.zeros(2, 11, 13, 32); rp[..., :24] = _nhwc(res)
    y = ops.conv2d(xp.to(dev), ops.PackedConv(conv, bn, dev, relu=ops.ACT_HSWISH)).cpu()
    assert y.shape == (2, 11, 13, 32) and float(y[..., 24:].abs().max()) == 0.0
    assert (y[..., :24].permute(0, 3, 1, 2) - ref_hs).abs().max().item() <= 2e-5
    y = ops.conv2d(xp.to(dev), ops.PackedConv(conv, bn, dev, relu=ops.ACT_NONE), res=rp.to(dev), res_mode=ops.RES_ADD_PRE_RELU).cpu()
    assert (y[..., :24].permute(0, 3, 1, 2) - ref_res).abs().max().item() <= 2e-5
    # exact concat slice: 24 columns at offset 48 of a 96-channel tensor, neighbours untouched
    big = torch.full((2, 11, 13, 96), 5.0, device=dev)
    ops.conv2d(xp.to(dev), ops.PackedConv(conv, bn, dev, relu=ops.ACT_HSWISH), out=big, out_coff=48, store=24)
    big = big.cpu()
    assert (big[..., 48:72].permute(0, 3, 1, 2) - ref_hs).abs().max().item() <= 2e-5
    assert float((big[..., :48] - 5).abs().max()) == 0 and float((big[..., 72:] - 5).abs().max()) == 0


@pytest.mark.parametrize("k,s,act", [(3, 1, 1), (3, 2, 2), (5, 1, 2), (5, 2, 1)])
def test_depthwise_conv(k, s, act):
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    Cc = 88
    conv = nn.Conv2d(Cc, Cc, k, s, (k - 1) // 2, groups=Cc, bias=False)
    bn = nn.BatchNorm2d(Cc, eps=1e-3).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cc, 1, k, k, seed=1) * 0.4)
        bn.running_mean.copy_(_rand(Cc, seed=5) * 0.2); bn.running_var.copy_(_rand(Cc, seed=6) * 0.5 + 1)
    x = _rand(2, Cc, 17, 23, seed=2) * 2
    with torch.no_grad():
        ref = bn(conv(x))
        ref = F.relu(ref) if act == 1 else F.hardswish(ref)
    xp = torch.zeros(2, 17, 23, 96); xp[..., :Cc] = _nhwc(x)
    y = ops.dwconv(xp.to(dev), ops.PackedDW(conv, bn, dev, act)).cpu()
    assert float(y[..., Cc:].abs().max()) == 0.0
    assert (y[..., :Cc].permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5


def test_squeeze_excitation():
    from pytorchocr_amd.modeling import ops
    from pytorchocr_amd.modeling.backbones.det_mobilenet_v3 import SqueezeExcitation
    dev = _dev()
    se = SqueezeExcitation(240)
    with torch.no_grad():
        for prm in se.parameters():
            prm.copy_(_rand(*prm.shape, seed=prm.numel()) * 0.3)
    x = _rand(3, 240, 50, 70, seed=3)
    with torch.no_grad():
        sc = F.hardsigmoid(se.fc2(F.relu(se.fc1(F.adaptive_avg_pool2d(x, 1)))))
        ref = sc * x
    xp = torch.zeros(3, 50, 70, 256); xp[..., :240] = _nhwc(x)
    y = ops.se_scale_(xp.to(dev), ops.PackedSE(se, dev)).cpu()
    assert float(y[..., 240:].abs().max()) == 0.0
    assert (y[..., :240].permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("case", [(2, 64, 20, 36, 64), (1, 256, 23, 40, 64), (1, 128, 16, 32, 128), (1, 64, 33, 47, 192),
                                  (5, 32, 150, 170, 64), (6, 64, 8, 48, 64), (7, 32, 4, 40, 64), (3, 32, 6, 21, 128), (8, 32, 23, 40, 64), (5, 32, 9, 24, 64),
                                  (1, 32, 32, 16, 64), (2, 48, 16, 16, 64), (9, 32, 8, 16, 64), (6, 16, 4, 81, 128), (3, 64, 46, 80, 64)])
@pytest.mark.parametrize("mode", ["1", "0"])
def test_winograd_3x3_matches_torch(case, mode, monkeypatch):
    """Both Winograd kernels -- F(4x4,3x3) (mode "1": what the cost model picks for every layer of the three networks) and
    F(2x2,3x3) (mode "0": the fallback) -- against torch fp32, incl. odd sizes, residual, concat slice, fused upsample store, and every
    patch geometry of either kernel (single-image patches, the 30-of-32-slot 6x5 patch, the 2- / 4-image patches of short maps)."""
    from pytorchocr_amd.modeling import ops
    assert ops.USE_WINOGRAD
    monkeypatch.setattr(ops, "WINO4_MODE", mode)
    N, Cin, H, W, Cout = case
    dev = _dev()
    conv = nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, 3, 3, seed=1) * (3.0 / (Cin * 9)) ** 0.5)
        bn.weight.copy_(_rand(Cout, seed=3) * 0.4 + 1); bn.bias.copy_(_rand(Cout, seed=4) * 0.2)
        bn.running_mean.copy_(_rand(Cout, seed=5) * 0.2); bn.running_var.copy_(_rand(Cout, seed=6) * 0.5 + 1)
    x = _rand(N, Cin, H, W, seed=7)
    res = _rand(N, Cout, H, W, seed=8)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
        ref_res = F.relu(bn(conv(x)) + res)
    pc = ops.PackedConv(conv, bn, dev, relu=True, cin_pad=Cin)
    assert pc.wino_u is not None and pc.wino4_ok
    xd = _nhwc(x).to(dev)
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    monkeypatch.setattr(ops, "PROFILE", [])
    monkeypatch.setattr(ops, "PROFILE_LABELS", [])
    y = ops.conv2d(xd, pc).cpu().permute(0, 3, 1, 2)
    assert ops.PROFILE_LABELS[0].startswith("wino43x3" if mode == "1" else "wino3x3 ")      # the kernel under test did run
    assert (y - ref).abs().max().item() <= tol
    y = ops.conv2d(xd, pc, res=_nhwc(res).to(dev), res_mode=ops.RES_ADD_PRE_RELU).cpu().permute(0, 3, 1, 2)
    assert (y - ref_res).abs().max().item() <= tol
    big = torch.full((N, H, W, Cout + 128), 3.0, device=dev)
    ops.conv2d(xd, pc, out=big, out_coff=64, store=Cout)
    big = big.cpu()
    assert (big[..., 64:64 + Cout].permute(0, 3, 1, 2) - ref).abs().max().item() <= tol
    assert float((big[..., :64] - 3).abs().max()) == 0 and float((big[..., 64 + Cout:] - 3).abs().max()) == 0
    # nearest upsample fused into the store (the FPN's p3/p4/p5 branches write straight into the concat buffer)
    up = 2 if H * W > 2000 else 4
    big = torch.full((N, H * up, W * up, Cout + 64), 3.0, device=dev)
    ops.conv2d(xd, pc, out=big, out_up=up, out_coff=64, store=Cout)
    big = big.cpu()
    ref_up = F.interpolate(ref, scale_factor=up, mode="nearest")
    assert (big[..., 64:].permute(0, 3, 1, 2) - ref_up).abs().max().item() <= tol
    assert float((big[..., :64] - 3).abs().max()) == 0


@pytest.mark.parametrize("case", [(8, 64, 16, 160, 128), (2, 64, 20, 36, 64), (5, 32, 150, 170, 64), (6, 64, 8, 48, 64), (7, 32, 4, 40, 64), (3, 32, 6, 22, 128),
                                  (1, 32, 32, 16, 64), (9, 32, 8, 16, 64), (6, 16, 4, 82, 128), (3, 64, 46, 80, 64), (2, 48, 2, 2, 64)])
def test_winograd_with_fused_relu_and_2x2_pool(case, monkeypatch):
    """ptocr_conv3x3_wino4_pool2_f32 (round 4: MaxPool2d(2, 2) taken inside the F(4x4) kernel's epilogue -- CRNN conv1 + pooling1) against
    the two launches it replaces: max is exact, so the pooled tensor must be IDENTICAL to maxpool2d(conv2d(x)) -- every patch geometry,
    maps with partial patches on both edges, the CRNN's own shape; and against torch within the Winograd tolerance"""
    from pytorchocr_amd.modeling import ops
    monkeypatch.setattr(ops, "WINO4_MODE", "1")
    N, Cin, H, W, Cout = case
    dev = _dev()
    conv = nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, 3, 3, seed=11) * (3.0 / (Cin * 9)) ** 0.5)
        bn.weight.copy_(_rand(Cout, seed=13) * 0.4 + 1); bn.bias.copy_(_rand(Cout, seed=14) * 0.2)
        bn.running_mean.copy_(_rand(Cout, seed=15) * 0.2); bn.running_var.copy_(_rand(Cout, seed=16) * 0.5 + 1)
    x = _rand(N, Cin, H, W, seed=17)
    pc = ops.PackedConv(conv, bn, dev, relu=True, cin_pad=Cin)
    xd = _nhwc(x).to(dev)
    monkeypatch.setattr(ops, "PROFILE", [])
    monkeypatch.setattr(ops, "PROFILE_LABELS", [])
    one = ops.conv2d_relu_pool2(xd, pc)
    assert len(ops.PROFILE_LABELS) == 1 and ops.PROFILE_LABELS[0].endswith("pool2")          # one launch, the fused one
    monkeypatch.setattr(ops, "CONV_POOL_FUSE", False)
    two = ops.conv2d_relu_pool2(xd, pc)
    torch.cuda.synchronize()
    assert one.shape == two.shape == (N, H // 2, W // 2, Cout) and torch.equal(one, two)
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(bn(conv(x))), 2, 2)
    assert (one.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    # odd sizes fall back to the two launches
    monkeypatch.setattr(ops, "CONV_POOL_FUSE", True)
    xo = _nhwc(_rand(1, Cin, 5, 7, seed=18)).to(dev)
    yo = ops.conv2d_relu_pool2(xo, pc)
    assert yo.shape == (1, 2, 3, Cout) and torch.equal(yo, ops.maxpool2d(ops.conv2d(xo, pc), 2, 2, 0))


@pytest.mark.parametrize("case", [(2, 20, 36, 256), (1, 7, 9, 96), (3, 46, 80, 32), (1, 184, 320, 256),
                                  (2, 20, 36, 256, 128), (1, 7, 9, 128, 128), (3, 92, 160, 256, 128), (40, 2, 6, 128, 128)])
def test_pointwise_k64_kernel(case):
    """1x1 kernels with LDS-resident weights, Cin = 64 (FPN lateral in2) and Cin = 128 (in3: half of the output channels per
    workgroup): plain, with the nearest-x2 top-down add after the ReLU, into a concat slice; pixel counts that are not multiples of
    the 128-pixel tile, fewer tiles than workgroups; and the generic kernel on the same layer"""
    from pytorchocr_amd.modeling import ops
    N, H, W, Cout = case[:4]
    Cin = case[4] if len(case) > 4 else 64
    dev = _dev()
    conv = nn.Conv2d(Cin, Cout, 1, 1, 0, bias=False)
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, 1, 1, seed=1) * 0.2)
        bn.weight.copy_(_rand(Cout, seed=3) * 0.4 + 1); bn.bias.copy_(_rand(Cout, seed=4) * 0.2)
        bn.running_mean.copy_(_rand(Cout, seed=5) * 0.2); bn.running_var.copy_(_rand(Cout, seed=6) * 0.5 + 1)
    x = _rand(N, Cin, H, W, seed=7)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
    pc = ops.PackedConv(conv, bn, dev, relu=True)
    assert pc.pw_w is not None
    xd = _nhwc(x).to(dev)
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    for use in (True, False):
        ops.USE_PW64_KERNEL = use
        ops.PW_MIN_PIXELS = 0                                # the size heuristic would send these small cases to the generic kernel
        try:
            y = ops.conv2d(xd, pc).cpu().permute(0, 3, 1, 2)
            assert (y - ref).abs().max().item() <= tol, use
            if H % 2 == 0 and W % 2 == 0:
                coarse = _rand(N, Cout, H // 2, W // 2, seed=9)
                ref_up = ref + F.interpolate(coarse, scale_factor=2, mode="nearest")
                y = ops.conv2d(xd, pc, res=_nhwc(coarse).to(dev), res_mode=ops.RES_ADD_UP2_POST_RELU).cpu().permute(0, 3, 1, 2)
                assert (y - ref_up).abs().max().item() <= tol, use
        finally:
            ops.USE_PW64_KERNEL = True
    ops.PW_MIN_PIXELS = 0
    big = torch.full((N, H, W, Cout + 96), 3.0, device=dev)
    ops.conv2d(xd, pc, out=big, out_coff=32, store=Cout)
    big = big.cpu()
    assert (big[..., 32:32 + Cout].permute(0, 3, 1, 2) - ref).abs().max().item() <= tol
    assert float((big[..., :32] - 3).abs().max()) == 0 and float((big[..., 32 + Cout:] - 3).abs().max()) == 0
    ops.PW_MIN_PIXELS = 4096


@pytest.mark.parametrize("cin,cout,act", [(16, 72, "hs"), (24, 96, "relu"), (40, 16, "none")])
def test_pointwise_small_k_padded_channels(cin, cout, act):
    """the same kernel on MobileNetV3-like layers: input / output channels zero-padded to 32-multiples, Hardswish"""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    conv = nn.Conv2d(cin, cout, 1, bias=False)
    bn = nn.BatchNorm2d(cout, eps=1e-3).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(cout, cin, 1, 1, seed=1) * 0.4)
        bn.running_mean.copy_(_rand(cout, seed=5) * 0.2); bn.running_var.copy_(_rand(cout, seed=6) * 0.5 + 1)
    x = _rand(2, cin, 21, 37, seed=7)
    with torch.no_grad():
        ref = bn(conv(x))
        ref = F.hardswish(ref) if act == "hs" else (F.relu(ref) if act == "relu" else ref)
    pc = ops.PackedConv(conv, bn, dev, relu={"hs": ops.ACT_HSWISH, "relu": True, "none": False}[act])
    assert pc.pw_w is not None and pc.pw_w.shape[0] in (32, 64)
    xp = torch.zeros(2, 21, 37, pc.cin); xp[..., :cin] = _nhwc(x)
    ops.PW_MIN_PIXELS = 0
    try:
        y = ops.conv2d(xp.to(dev), pc).cpu()
    finally:
        ops.PW_MIN_PIXELS = 4096
    assert y.shape[3] == pc.c_tensor and (pc.c_tensor == cout or float(y[..., cout:].abs().max()) == 0.0)
    assert (y[..., :cout].permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cin,shape", [(1, (3, 32, 320)), (3, (2, 32, 100)), (1, (2, 9, 15))])
def test_small_cin_conv_relu_pool_fused(cin, shape):
    """CRNN conv0 + relu0 + pooling0 in one kernel vs torch fp32 (and vs the unfused generic kernels), odd sizes included"""
    from pytorchocr_amd.modeling import ops
    N, H, W = shape
    dev = _dev()
    conv = nn.Conv2d(cin, 64, 3, 1, 1)
    with torch.no_grad():
        conv.weight.copy_(_rand(64, cin, 3, 3, seed=1) * 0.4); conv.bias.copy_(_rand(64, seed=2) * 0.3)
    x = _rand(N, cin, H, W, seed=7)
    with torch.no_grad():
        ref = F.max_pool2d(F.relu(conv(x)), 2, 2)
    pc = ops.PackedConv(conv, None, dev, relu=True, cin_pad=4)
    assert pc.small_w is not None
    x4 = ops.nchw_to_nhwc(x.to(dev), 4)
    tol = 2e-5 * max(1.0, ref.abs().max().item())
    for use in (True, False):
        ops.USE_SMALL_CONV_KERNEL = use
        try:
            y = ops.conv3x3_relu_pool2(x4, pc).cpu().permute(0, 3, 1, 2)
        finally:
            ops.USE_SMALL_CONV_KERNEL = True
        assert y.shape == ref.shape and (y - ref).abs().max().item() <= tol, use


def test_winograd_narrow_layer_padded_cout():
    """3x3 96 -> 24 (MobileNetV3's FPN smoothing / head convs): Winograd weights zero-padded to 64 output channels, only the real
    ones (or the tensor's 32-channel padding, as zeros) are stored; concat slices at 24-channel offsets"""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    conv = nn.Conv2d(96, 24, 3, 1, 1, bias=False)
    bn = nn.BatchNorm2d(24).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(24, 96, 3, 3, seed=1) * 0.06)
        bn.running_mean.copy_(_rand(24, seed=5) * 0.2); bn.running_var.copy_(_rand(24, seed=6) * 0.5 + 1)
    x = _rand(2, 96, 22, 38, seed=7)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
    pc = ops.PackedConv(conv, bn, dev, relu=True)
    assert pc.wino_u is not None and pc.wino_cout == 64 and pc.c_tensor == 32
    xd = _nhwc(x).to(dev)
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    y = ops.conv2d(xd, pc).cpu()
    assert y.shape[3] == 32 and float(y[..., 24:].abs().max()) == 0.0
    assert (y[..., :24].permute(0, 3, 1, 2) - ref).abs().max().item() <= tol
    fuse = torch.full((2, 44, 76, 96), 3.0, device=dev)
    ops.conv2d(xd, pc, out=fuse, out_up=2, out_coff=48, store=24)
    fuse = fuse.cpu()
    assert (fuse[..., 48:72].permute(0, 3, 1, 2) - F.interpolate(ref, scale_factor=2, mode="nearest")).abs().max().item() <= tol
    assert float((fuse[..., :48] - 3).abs().max()) == 0 and float((fuse[..., 72:] - 3).abs().max()) == 0


def test_conv_on_tensors_beyond_2gib_runs_as_half_batches():
    """a 3x3 conv whose 256-channel input exceeds 2 GiB (the concat buffer of a 64-image batch in run_ocr): split into half batches,
    the fast kernel runs (not the first-generation fallback) and every image equals its single-image result"""
    from pytorchocr_amd.modeling import ops
    dev = _dev()
    conv = nn.Conv2d(256, 64, 3, 1, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(_rand(64, 256, 3, 3, seed=1) * 0.02)
    pc = ops.PackedConv(conv, None, dev, relu=True)
    N, H, W = 5, 736, 640                                       # 5 x 736 x 640 x 256 x 4 B = 2.4 GB
    x = torch.empty((N, H, W, 256), dtype=torch.float32, device=dev)
    for i in range(N):
        x[i] = torch.from_numpy(np.ascontiguousarray(_nhwc(_rand(1, 256, 64, W, seed=20 + i))[0])).to(dev).repeat(H // 64 + 1, 1, 1)[:H]
    assert x.numel() * 4 >= 2 ** 31
    ops.PROFILE, ops.PROFILE_LABELS = [], []
    try:
        y = ops.conv2d(x, pc)
        labels = list(ops.PROFILE_LABELS)
    finally:
        ops.PROFILE = ops.PROFILE_LABELS = None
    assert len(labels) >= 2 and all(l.startswith("wino") for l in labels), labels
    for i in (0, 2, 4):
        assert torch.equal(y[i:i + 1], ops.conv2d(x[i:i + 1].contiguous(), pc))


@pytest.mark.parametrize("seed", range(6 + int(os.environ.get("PTOCR_CONV_FUZZ", "0"))))      # PTOCR_CONV_FUZZ=n: n more seeds
def test_conv_random_shapes_against_torch(seed):
    """random layer shapes through ops.conv2d against torch fp32: 3x3 / s1 (whichever Winograd form the cost model picks, with the
    residual and the upsampled concat-slice store), 3x3 / s2 and 1x1 on the generic and the LDS-resident-weights kernels"""
    from pytorchocr_amd.modeling import ops
    rng = np.random.default_rng(7000 + seed)
    dev = _dev()
    kind = int(rng.integers(0, 3))                                   # 0: 3x3 s1, 1: 3x3 s2, 2: 1x1
    N = int(rng.integers(1, 6))
    H, W = int(rng.integers(3, 70)), int(rng.integers(3, 90))
    Cin = int(rng.choice([16, 32, 48, 64, 96, 128, 256]))
    Cout = int(rng.choice([24, 32, 64, 96, 128, 192, 256]))
    k, stride, pad = (3, 1, 1) if kind == 0 else ((3, 2, 1) if kind == 1 else (1, 1, 0))
    conv = nn.Conv2d(Cin, Cout, k, stride, pad, bias=False)
    bn = nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, k, k, seed=11 + seed) * (3.0 / (Cin * k * k)) ** 0.5)
        bn.weight.copy_(_rand(Cout, seed=3) * 0.4 + 1); bn.bias.copy_(_rand(Cout, seed=4) * 0.2)
        bn.running_mean.copy_(_rand(Cout, seed=5) * 0.2); bn.running_var.copy_(_rand(Cout, seed=6) * 0.5 + 1)
    relu = bool(rng.integers(0, 2))
    x = _rand(N, Cin, H, W, seed=70 + seed)
    with torch.no_grad():
        pre = bn(conv(x))
        ref = F.relu(pre) if relu else pre
    pc = ops.PackedConv(conv, bn, dev, relu=relu)                   # input channels padded as the models pad them (zeros)
    xp = torch.zeros(N, H, W, pc.cin)
    xp[..., :Cin] = _nhwc(x)
    xd = xp.to(dev)
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    y = ops.conv2d(xd, pc).cpu()
    assert y.shape[3] >= Cout and float(y[..., Cout:].abs().max() if y.shape[3] > Cout else 0.0) == 0.0
    assert (y[..., :Cout].permute(0, 3, 1, 2) - ref).abs().max().item() <= tol, (kind, N, H, W, Cin, Cout)
    if kind == 0 and relu:
        res = _rand(N, Cout, H, W, seed=90 + seed)
        rp = torch.zeros(N, H, W, y.shape[3]); rp[..., :Cout] = _nhwc(res)
        y = ops.conv2d(xd, pc, res=rp.to(dev), res_mode=ops.RES_ADD_PRE_RELU).cpu()
        assert (y[..., :Cout].permute(0, 3, 1, 2) - F.relu(pre + res)).abs().max().item() <= tol, ("res", N, H, W, Cin, Cout)
        if Cout % 4 == 0:
            up = int(rng.choice([1, 2, 4]))
            big = torch.full((N, H * up, W * up, Cout + 64), 3.0, device=dev)
            ops.conv2d(xd, pc, out=big, out_up=up, out_coff=32, store=Cout)
            big = big.cpu()
            exp = ref.repeat_interleave(up, 2).repeat_interleave(up, 3)
            assert (big[..., 32:32 + Cout].permute(0, 3, 1, 2) - exp).abs().max().item() <= tol, ("up", up, N, H, W, Cin, Cout)
            assert float((big[..., :32] - 3).abs().max()) == 0 and float((big[..., 32 + Cout:] - 3).abs().max()) == 0


def test_first_cut_winograd_kernel_still_agrees(monkeypatch):
    """PTOCR_WINO4R=0 keeps conv_wino4_kernel (the first cut of the F(4x4,3x3) kernel) selectable: its weights are packed only then
    (round 6), and it must still agree with the re-cut and with torch -- plain, residual and fused-pool forms."""
    from pytorchocr_amd.modeling import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    N, Cin, Cout, H, W = 3, 64, 64, 40, 72
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(Cout).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout) * 0.4 + 1); bn.bias.copy_(torch.randn(Cout) * 0.2)
        bn.running_mean.copy_(torch.randn(Cout) * 0.2); bn.running_var.copy_(torch.rand(Cout) * 0.5 + 1)
    x, res = torch.randn(N, Cin, H, W), torch.randn(N, Cout, H, W)
    with torch.no_grad():
        ref = F.relu(bn(conv(x)))
        ref_res = F.relu(bn(conv(x)) + res)
        ref_pool = F.max_pool2d(ref, 2, 2)
    xd, rd = x.permute(0, 2, 3, 1).contiguous().to(dev), res.permute(0, 2, 3, 1).contiguous().to(dev)
    monkeypatch.setattr(ops, "WINO4_MODE", "1")
    outs = {}
    for recut in (True, False):
        monkeypatch.setattr(ops, "WINO4R", recut)
        pc = ops.PackedConv(conv, bn, dev, relu=True, cin_pad=Cin)
        assert pc.wino4_ok and (pc.wino4r_u is not None) == recut and (pc.wino4_u is not None) == (not recut)
        monkeypatch.setattr(ops, "PROFILE", [])
        monkeypatch.setattr(ops, "PROFILE_LABELS", [])
        outs[recut] = (ops.conv2d(xd, pc), ops.conv2d(xd, pc, res=rd, res_mode=ops.RES_ADD_PRE_RELU), ops.conv2d_relu_pool2(xd, pc))
        assert all(l.startswith("wino43x3") for l in ops.PROFILE_LABELS[:2])
    tol = 3e-5 * max(1.0, ref_res.abs().max().item())
    for got, want in zip(outs[False], (ref, ref_res, ref_pool)):
        assert (got.cpu().permute(0, 3, 1, 2) - want).abs().max().item() <= tol
    for a, b in zip(outs[True], outs[False]):                      # the two cuts share transforms and summation order per tile
        assert (a - b).abs().max().item() <= tol
