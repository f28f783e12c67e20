"""bf16 inference path of the MobileNetV3 DB detector (BASELINE.json configs[3]) on the HIP engine (-m gpu).

Tolerance.  north_star's 1e-4 is an fp32 statement; bf16 carries 8 significant bits (relative rounding 2^-9 per stored
activation and weight), through ~45 layers.  What the bf16 path is held to, against the REFERENCE's own fp32 outputs
(tests/golden) and the fp32 oracle:  max |p_bf16 - p_fp32| <= 3e-2 on the probability maps, mean <= 3e-3, and at most 0.5 % of the
pixels on the other side of the 0.3 binarisation threshold (those are the pixels that can move a box vertex; the post-process
itself is bit-exact on whatever map it is given).  The per-kernel tests compare each bf16 kernel with a float64 evaluation of the
same bf16-rounded operands (tight: only the accumulation order and the output rounding differ)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pytorchocr_amd.utils.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu

MBV3S = dict(model_type="det", algorithm="DB", Transform=None,
             Backbone=dict(name="MobileNetV3", model_name="small", scale=1.0, pretrained=False),
             Neck=dict(name="FPN", out_channels=96, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50))


def _model(contract):
    from pytorchocr_amd.modeling.architectures import build_model
    m = build_model(dict(MBV3S))
    sd = synth_state_dict(contract["det_mbv3s_db"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval(), sd


def _bf(t):
    return t.to(torch.bfloat16)


def test_pwconv_kernel_matches_float64_of_the_same_operands():
    from pytorchocr_amd.modeling import bf16_path as bp
    torch.manual_seed(0)
    for (n, h, w, cin, cout, act, res_mode, use_scale) in ((2, 5, 7, 16, 72, 1, 0, False), (1, 6, 8, 96, 576, 2, 0, False),
                                                            (3, 4, 6, 240, 40, 0, 1, True), (2, 8, 12, 48, 96, 1, 2, False)):
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False)
        bn = torch.nn.BatchNorm2d(cout).eval()
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.6, 1.4); bn.bias.data.uniform_(-0.2, 0.2)
        pw = bp._Pw(conv, bn, torch.device("cuda:0"), act)
        x = _bf(torch.randn(n, h, w, pw.cin)); x[..., cin:] = 0
        res = scale = None
        if res_mode == 1:
            res = _bf(torch.randn(n, h, w, pw.cstore))
        if res_mode == 2:
            res = _bf(torch.randn(n, h // 2, w // 2, pw.cstore))
        if res is not None:
            res[..., cout:] = 0                               # padding channels of every activation tensor hold zeros
        if use_scale:
            scale = torch.rand(n, pw.cin)
        y = bp.pwconv(x.cuda(), pw, res=res.cuda() if res is not None else None, res_mode=res_mode,
                      scale=scale.cuda() if scale is not None else None).float().cpu()
        xd = x.double()
        if use_scale:
            xd = _bf((x.float() * scale[:, None, None, :])).double()            # the gate is applied to the input, rounded to bf16 once
        ref = xd.reshape(-1, pw.cin) @ pw.w.cpu().double().t()[:, :pw.cstore] + pw.b.cpu().double()[:pw.cstore]
        ref = ref.reshape(n, h, w, pw.cstore)
        if res_mode == 1:
            ref = ref + res.double()
        ref = F.relu(ref) if act == 1 else (F.hardswish(ref) if act == 2 else ref)
        if res_mode == 2:
            ref = ref + res.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
        err = (y.double() - ref).abs()
        assert float((err / (ref.abs() + 1.0)).max()) <= 6e-3, (cin, cout, float(err.max()))      # one bf16 rounding of the output
        assert float(y[..., cout:].abs().max()) == 0.0 if pw.cstore > cout else True                 # padding channels stay zero


def test_conv3x3_and_dwconv_kernels():
    from pytorchocr_amd.modeling import bf16_path as bp
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(96, 24, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(24).eval(); bn.running_var.uniform_(0.5, 1.5); bn.running_mean.uniform_(-0.2, 0.2)
    c3 = bp._C3(conv, bn, torch.device("cuda:0"), 1)
    x = _bf(torch.randn(2, 9, 11, 96))
    out = torch.zeros(2, 18, 22, 96, dtype=torch.bfloat16, device="cuda:0")
    bp.conv3x3(x.cuda(), c3, out=out, up=2, coff=24, cstore=24)
    w = c3.w.cpu().double().reshape(32, 3, 3, 96)[:24].permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w, c3.b.cpu().double()[:24], 1, 1)).permute(0, 2, 3, 1)
    ref = ref.repeat_interleave(2, 1).repeat_interleave(2, 2)
    got = out.float().cpu()
    assert float(((got[..., 24:48].double() - ref).abs() / (ref.abs() + 1)).max()) <= 6e-3
    assert float(got[..., :24].abs().max()) == 0 and float(got[..., 48:].abs().max()) == 0      # other slices of the concat untouched
    for k, stride, c in ((3, 1, 16), (5, 2, 96), (5, 1, 240), (5, 1, 576), (3, 2, 72)):     # 576: two channel groups per block row; 72: nine octets
        dw = torch.nn.Conv2d(c, c, k, stride, (k - 1) // 2, groups=c, bias=False)
        bn = torch.nn.BatchNorm2d(c).eval(); bn.running_var.uniform_(0.5, 1.5)
        pd = bp._Dw(dw, bn, torch.device("cuda:0"), 2)
        x = _bf(torch.randn(3, 37, 41, pd.c))
        y, partial, nblk = bp.dwconv(x.cuda(), pd, True)
        wd = pd.w.cpu().double().t().reshape(pd.c, 1, k, k)
        ref = F.hardswish(F.conv2d(x.double().permute(0, 3, 1, 2), wd, pd.b.cpu().double(), stride, (k - 1) // 2, groups=pd.c)).permute(0, 2, 3, 1)
        assert float(((y.float().cpu().double() - ref).abs() / (ref.abs() + 1)).max()) <= 6e-3
        pooled = partial.sum(1).cpu().double()
        assert float((pooled - ref.sum((1, 2))).abs().max()) <= 1e-2 * ref.shape[1] * ref.shape[2] * 1e-2 + 1e-2


def test_conv3x3_many_tiles_images_and_concat_slice():
    """the persistent 3x3 kernel on a map of several tiles per image (partial tiles on both edges), several images, written into the last
    slice of a concat buffer whose other slices must stay untouched; and the 32-channel output form the head uses"""
    from pytorchocr_amd.modeling import bf16_path as bp
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(96, 24, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(24).eval(); bn.running_var.uniform_(0.5, 1.5); bn.running_mean.uniform_(-0.2, 0.2)
    c3 = bp._C3(conv, bn, torch.device("cuda:0"), 1)
    n, h, w = 5, 37, 75                                          # 5 x 3 tiles of 8 x 32 per image
    x = _bf(torch.randn(n, h, w, 96))
    wd = c3.w.cpu().double().reshape(32, 3, 3, 96)[:24].permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), wd, c3.b.cpu().double()[:24], 1, 1)).permute(0, 2, 3, 1)
    out = torch.full((n, h, w, 96), 7.0, dtype=torch.bfloat16, device="cuda:0")
    bp.conv3x3(x.cuda(), c3, out=out, up=1, coff=72, cstore=24)
    got = out.float().cpu()
    assert float(((got[..., 72:].double() - ref).abs() / (ref.abs() + 1)).max()) <= 6e-3
    assert float((got[..., :72] - 7.0).abs().max()) == 0
    y = bp.conv3x3(x.cuda(), c3).float().cpu()                  # own tensor, 32 channels wide
    assert y.shape == (n, h, w, 32) and float(y[..., 24:].abs().max()) == 0
    assert float(((y[..., :24].double() - ref).abs() / (ref.abs() + 1)).max()) <= 6e-3


@pytest.mark.parametrize("hw", [(38, 64), (37, 62)])            # W % 4 == 0: two pixels per thread on 16-byte loads; otherwise one per thread
def test_stem_kernel_both_forms(hw):
    import ctypes as C
    from pytorchocr_amd import _lib
    from pytorchocr_amd.modeling import bf16_path as bp
    torch.manual_seed(6)
    h, w = hw
    x = torch.randn(3, 3, h, w)
    wt = torch.randn(16, 3, 3, 3) * 0.3
    b = torch.randn(16) * 0.1
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    t = torch.empty((3, ho, wo, 16), dtype=torch.bfloat16, device="cuda:0")
    xd, wd, bd = x.cuda(), wt.permute(1, 2, 3, 0).reshape(27, 16).contiguous().cuda(), b.cuda()
    _lib.check(_lib.lib().ptocr_stem3x3s2_bf16(bp._ptr(xd), bp._ptr(wd), bp._ptr(bd), bp._ptr(t), 3, h, w, 2, _lib.cur_stream()), "stem")
    ref = F.hardswish(F.conv2d(x.double(), wt.double(), b.double(), 2, 1)).permute(0, 2, 3, 1)
    assert float(((t.float().cpu().double() - ref).abs() / (ref.abs() + 1)).max()) <= 6e-3


def test_se_gate_kernels_match_float64():
    """hardsigmoid(fc2(relu(fc1(mean)))) from the per-block channel sums: the one-block-per-image kernel and the eight-blocks-per-image
    pair used for wide layers, against float64"""
    from pytorchocr_amd import _lib
    from pytorchocr_amd.modeling import bf16_path as bp
    torch.manual_seed(3)
    for (n, c, s_, nblk, hw) in ((3, 576, 144, 23, 920), (2, 288, 72, 12, 3680), (5, 1024, 256, 7, 49), (1, 96, 24, 24, 3680)):
        partial = torch.randn(n, nblk, c) * 3
        w1, b1, w2, b2 = torch.randn(s_, c) * 0.1, torch.randn(s_) * 0.1, torch.randn(c, s_) * 0.1, torch.randn(c) * 0.1
        mean = partial.double().sum(1) / hw
        ref = F.hardsigmoid(F.relu(mean @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
        d = [t.cuda().contiguous() for t in (partial, w1.t(), b1, w2.t(), b2)]
        for split in (False, True):
            scale = torch.empty(n, c, device="cuda:0")
            if split:
                hid = torch.empty(n, s_, device="cuda:0")
                _lib.check(_lib.lib().ptocr_se_fc_split_f32(*[bp._ptr(t) for t in d], bp._ptr(hid), bp._ptr(scale), n, hw, c, s_, nblk, _lib.cur_stream()), "split")
            else:
                _lib.check(_lib.lib().ptocr_se_fc_t_f32(*[bp._ptr(t) for t in d], bp._ptr(scale), n, hw, c, s_, nblk, _lib.cur_stream()), "single")
            assert float((scale.cpu().double() - ref).abs().max()) <= 2e-5, (c, split)


def test_bf16_maps_against_reference_golden_and_oracle(gold_dir, contract):
    from oracle import model_oracle
    m, sd = _model(contract)
    g = np.load(os.path.join(gold_dir, "det_mbv3s_db_1x3x64x96.npz"))
    x = torch.from_numpy(synth_images(1, 3, 64, 96, seed=int(g["seed"]))).cuda()
    with torch.no_grad():
        p32 = m(x)["maps"].cpu().numpy()
        m.set_compute_dtype("bf16")
        p16 = m(x)["maps"].cpu().numpy()
    assert p16.dtype == np.float32 and p16.shape == g["maps"].shape
    assert np.abs(p32 - g["maps"]).max() <= 1e-4                    # the fp32 path still meets its own bar
    d = np.abs(p16 - g["maps"])
    assert d.max() <= 3e-2 and d.mean() <= 3e-3, (d.max(), d.mean())
    xs = synth_images(2, 3, 224, 320, seed=31)
    with torch.no_grad():
        p16 = m(torch.from_numpy(xs).cuda())["maps"].cpu().numpy()
    ref = model_oracle.dbnet_forward(sd, torch.from_numpy(xs))["maps"].numpy()
    d = np.abs(p16 - ref)
    flips = ((p16 > 0.3) != (ref > 0.3)).mean()
    assert d.max() <= 3e-2 and d.mean() <= 3e-3 and flips <= 5e-3, (d.max(), d.mean(), flips)
    m.set_compute_dtype("f32")
    with torch.no_grad():
        assert np.abs(m(torch.from_numpy(xs).cuda())["maps"].cpu().numpy() - ref).max() <= 1e-4


def test_bf16_config3_size_properties(contract):
    """736x1280 (BASELINE configs[3] geometry): one image against the fp32 oracle, then batch 32 against itself -- the maps of an
    image do not depend on the batch it travels in, and two runs give the same bits"""
    from oracle import model_oracle
    m, sd = _model(contract)
    m.set_compute_dtype("bf16")
    base = synth_images(2, 3, 736, 1280, seed=9)
    with torch.no_grad():
        one = m(torch.from_numpy(base[:1]).cuda())["maps"]
        ref = model_oracle.dbnet_forward(sd, torch.from_numpy(base[:1]))["maps"].numpy()
        d = np.abs(one.cpu().numpy() - ref)
        flips = ((one.cpu().numpy() > 0.3) != (ref > 0.3)).mean()
        assert d.max() <= 3e-2 and d.mean() <= 3e-3 and flips <= 5e-3, (d.max(), d.mean(), flips)
        xb = torch.from_numpy(base).cuda().repeat(16, 1, 1, 1)
        a = m(xb)["maps"]
        b = m(xb)["maps"]
    assert a.shape == (32, 1, 736, 1280) and torch.equal(a, b)
    assert torch.equal(a[0], one[0]) and torch.equal(a[2], a[0]) and torch.equal(a[31], a[1])
    with pytest.raises(NotImplementedError):
        from pytorchocr_amd.modeling.architectures import build_model
        build_model(dict(model_type="det", algorithm="DB", Transform=None, Backbone=dict(name="ResNet", layers=18, pretrained=False),
                         Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False), Head=dict(name="DBHead", k=50))
                    ).to("cuda:0").eval().set_compute_dtype("bf16")(torch.zeros(1, 3, 64, 64, device="cuda:0"))
