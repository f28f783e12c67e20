"""Host-side logic that needs no GPU: state_dict contract, BN folding / weight packing, config handling, decode."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import ctc_oracle
from pytorchocr_amd.modeling import ops
from pytorchocr_amd.modeling.architectures import build_model
from pytorchocr_amd.postprocess import build_post_process
from pytorchocr_amd.utils.synth import synth_state_dict

DET = dict(model_type="det", algorithm="DB", Transform=None,
           Backbone=dict(name="ResNet", layers=18, pretrained=True, ckpt_path=".../model_zoo/resnet18-5c106cde.pth"),
           Neck=dict(name="FPN", out_channels=256, mode="DB", use_asf=False, attention_type="scale_channel_spatial"),
           Head=dict(name="DBHead", k=50))
REC = dict(model_type="rec", algorithm="CRNN", in_channels=1, Transform=None,
           Backbone=dict(name="VGG", model_name="v1", scale=1.0, pretrained=False, ckpt_path=None),
           Neck=dict(name="SequenceEncoder", encoder_type="rnn", hidden_size=256),
           Head=dict(name="CTCHead", out_channels=6624))


@pytest.mark.parametrize("cfg,name", [(DET, "det_r18_db"), (REC, "rec_vgg_bilstm_ctc")])
def test_state_dict_contract_and_strict_load(contract, cfg, name):
    m = build_model(cfg)                      # stock yml: pretrained=True with a placeholder path must NOT fetch a URL
    ref = contract[name]
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k][0] and str(v.dtype) == ref[k][1], k
    w = {("module." + k): torch.from_numpy(v) for k, v in synth_state_dict(ref).items()}
    m.load_state_dict({k[len("module."):]: v for k, v in w.items()}, strict=True)
    assert cfg["Backbone"].get("in_channels") is None          # build_model deep-copies its config (reference __init__.py:10)


def test_bn_fold_and_pack_reproduce_conv_bn():
    conv = nn.Conv2d(8, 64, 3, 2, 1, bias=True)
    bn = nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2); bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    pc = ops.PackedConv(conv, bn, torch.device("cpu"), relu=False, cin_pad=32)
    assert pc.w.shape == (64, 288) and pc.cin == 32
    wk = pc.w.reshape(64, 3, 3, 32)[..., :8].permute(0, 3, 1, 2)
    x = torch.randn(2, 8, 9, 11)
    with torch.no_grad():
        ref = bn(conv(x))
        got = F.conv2d(x, wk, pc.b, 2, 1)
    assert (ref - got).abs().max() < 1e-5
    assert float(pc.w.reshape(64, 3, 3, 32)[..., 8:].abs().max()) == 0.0


def test_convtranspose_pack_layout():
    t = nn.ConvTranspose2d(64, 64, 2, 2)
    pt = ops.PackedConvT2x2(t, None, torch.device("cpu"), relu=False)
    x = torch.randn(1, 64, 3, 5)
    with torch.no_grad():
        ref = t(x)
        cols = torch.einsum("nchw,kc->nkhw", x, pt.w) + pt.b.view(1, -1, 1, 1)     # [1, 4*64, 3, 5]
    got = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            got[:, :, a::2, b::2] = cols[:, (a * 2 + b) * 64:(a * 2 + b + 1) * 64]
    assert (ref - got).abs().max() < 1e-5


def test_unsupported_options_raise():
    with pytest.raises(NotImplementedError):
        build_post_process(dict(name="DBPostProcess", cpp_speedup=False), {})
    with pytest.raises(NotImplementedError):
        build_post_process(dict(name="DBPostProcess", cpp_speedup=True, use_dilation=True), {})
    with pytest.raises(NotImplementedError):
        build_model(dict(DET, Neck=dict(DET["Neck"], use_asf=True, attention_type="scale_spatial")))
    assert hasattr(build_model(dict(DET, Neck=dict(DET["Neck"], use_asf=True))).neck, "concat_attention")
    with pytest.raises(NotImplementedError):
        build_model(dict(DET, Backbone=dict(name="ResNet", layers=50)))
    with pytest.raises(AssertionError):
        build_model(dict(DET, Backbone=dict(name="MobileNetV3", model_name="small")))
    m = build_model(DET)
    with pytest.raises(RuntimeError):
        m.eval()(torch.zeros(1, 3, 32, 32))                    # CPU tensor: no fallback


def test_ctc_decode_host_part_matches_oracle():
    post = build_post_process(dict(name="CTCLabelDecode"), dict(character_dict_path=None, use_space_char=False, seed=1))
    chars = ctc_oracle.load_characters(None)
    assert post.character == chars
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 5, size=(40, 30))
    idx[3] = 0
    prob = rng.random((40, 30)).astype(np.float32)
    got = post.decode(idx, prob, is_remove_duplicate=True)
    exp = ctc_oracle.decode(idx, prob, chars)
    for (t1, c1), (t2, c2) in zip(got, exp):
        assert t1 == t2 and ((math.isnan(c1) and math.isnan(c2)) or abs(c1 - c2) < 1e-7)
    assert got[3][0] == "" and math.isnan(got[3][1])
