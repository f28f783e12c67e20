"""MobileNetV3 (small / large, any width) detection backbone on the HIP engine.

Mirror of reference `MobileNetV3` (pytocr/modeling/backbones/det_mobilenet_v3.py:154-279; block `InvertedResidual`
:106-151, `SqueezeExcitation` :67-85, `ConvBNActivation` :38-61, table `_mobilenet_v3_conf` :282-325): same constructor
arguments, parameter names and four stage outputs.  1x1 expand / project convs run on the MFMA conv kernel (channels
zero-padded to multiples of 32, Hardswish / ReLU / residual add fused in the epilogue), depthwise 3x3 / 5x5 convs and
Squeeze-Excitation are dedicated memory-bound kernels.  BatchNorm eps is 1e-3 as in the reference (:202).
"""
import logging
import os
from functools import partial

import torch
from torch import nn

from .. import ops


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _cba(cin, cout, k, stride, groups, norm_layer, act_layer):
    pad = (k - 1) // 2
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=False), norm_layer(cout), act_layer(inplace=True)
                         if act_layer is not nn.Identity else nn.Identity())


class SqueezeExcitation(nn.Module):
    def __init__(self, input_channels, squeeze_factor=4):
        super().__init__()
        squeeze_channels = _make_divisible(input_channels // squeeze_factor, 8)
        self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1)


class _Cnf:
    def __init__(self, cin, k, exp, cout, use_se, act, stride, width_mult):
        adj = lambda c: _make_divisible(c * width_mult, 8)
        self.input_channels, self.kernel, self.expanded_channels, self.out_channels = adj(cin), k, adj(exp), adj(cout)
        self.use_se, self.use_hs, self.stride = use_se, act == "HS", stride


def _conf(arch, width_mult, use_se):
    c = partial(_Cnf, width_mult=width_mult)
    if arch == "large":
        t = [(16, 3, 16, 16, False, "RE", 1), (16, 3, 64, 24, False, "RE", 2), (24, 3, 72, 24, False, "RE", 1),
             (24, 5, 72, 40, use_se, "RE", 2), (40, 5, 120, 40, use_se, "RE", 1), (40, 5, 120, 40, use_se, "RE", 1),
             (40, 3, 240, 80, False, "HS", 2), (80, 3, 200, 80, False, "HS", 1), (80, 3, 184, 80, False, "HS", 1),
             (80, 3, 184, 80, False, "HS", 1), (80, 3, 480, 112, use_se, "HS", 1), (112, 3, 672, 112, use_se, "HS", 1),
             (112, 5, 672, 160, True, "HS", 2), (160, 5, 960, 160, True, "HS", 1), (160, 5, 960, 160, True, "HS", 1)]
    elif arch == "small":
        t = [(16, 3, 16, 16, use_se, "RE", 2), (16, 3, 72, 24, False, "RE", 2), (24, 3, 88, 24, False, "RE", 1),
             (24, 5, 96, 40, use_se, "HS", 2), (40, 5, 240, 40, use_se, "HS", 1), (40, 5, 240, 40, use_se, "HS", 1),
             (40, 5, 120, 48, use_se, "HS", 1), (48, 5, 144, 48, use_se, "HS", 1), (48, 5, 288, 96, True, "HS", 2),
             (96, 5, 576, 96, True, "HS", 1), (96, 5, 576, 96, True, "HS", 1)]
    else:
        raise ValueError("Unsupported model type {}".format(arch))
    return [c(*row) for row in t]


class InvertedResidual(nn.Module):
    def __init__(self, cnf, norm_layer):
        super().__init__()
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.Hardswish if cnf.use_hs else nn.ReLU
        self.act_code = ops.ACT_HSWISH if cnf.use_hs else ops.ACT_RELU
        self.conv1 = _cba(cnf.input_channels, cnf.expanded_channels, 1, 1, 1, norm_layer, act) \
            if cnf.expanded_channels != cnf.input_channels else None
        self.conv2 = _cba(cnf.expanded_channels, cnf.expanded_channels, cnf.kernel, cnf.stride, cnf.expanded_channels, norm_layer, act)
        self.se = SqueezeExcitation(cnf.expanded_channels) if cnf.use_se else None
        self.conv3 = _cba(cnf.expanded_channels, cnf.out_channels, 1, 1, 1, norm_layer, nn.Identity)

    def pack(self, dev):
        p = {"dw": ops.PackedDW(self.conv2[0], self.conv2[1], dev, self.act_code),
             "pw": ops.PackedConv(self.conv3[0], self.conv3[1], dev, relu=ops.ACT_NONE), "res": self.use_res_connect}
        if self.conv1 is not None:
            p["ex"] = ops.PackedConv(self.conv1[0], self.conv1[1], dev, relu=self.act_code)
        if self.se is not None:
            p["se"] = ops.PackedSE(self.se, dev)
        return p

    @staticmethod
    def run(p, x):
        out = ops.conv2d(x, p["ex"]) if "ex" in p else x
        out = ops.dwconv(out, p["dw"])
        if "se" in p:
            out = ops.se_scale_(out, p["se"])
        if p["res"]:
            return ops.conv2d(out, p["pw"], res=x, res_mode=ops.RES_ADD_PRE_RELU)      # project + identity, no activation
        return ops.conv2d(out, p["pw"])


class MobileNetV3(ops.PackedModule):
    def __init__(self, in_channels=3, model_name="large", width_mult=1.0, use_se=True, dilation=False, reduced_tail=False,
                 pretrained=False, ckpt_path=None, scale=None, **kwargs):
        super().__init__()
        if scale is not None:                       # BASELINE.json writes "mbv3small_x1.0" as Backbone.scale
            width_mult = scale
        assert width_mult in [0.35, 0.5, 0.75, 1.0, 1.25], "supported scale are [0.35, 0.5, 0.75, 1.0, 1.25] but input width_mult is {}".format(width_mult)
        if dilation or reduced_tail:
            raise NotImplementedError("pytorchocr_amd MobileNetV3: dilation / reduced_tail are not on the hot path")
        setting = _conf(model_name, width_mult, use_se)
        norm_layer = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01)
        first = setting[0].input_channels
        self.conv1 = _cba(in_channels, first, 3, 2, 1, norm_layer, nn.Hardswish)
        self.stages = nn.ModuleList()
        self.out_channels = []
        layers, i, start_idx = [], 0, (2 if model_name == "large" else 0)
        for cnf in setting:
            if cnf.stride == 2 and i > start_idx:
                self.stages.append(nn.Sequential(*layers))
                self.out_channels.append(cnf.input_channels)
                layers = []
            layers.append(InvertedResidual(cnf, norm_layer))
            i += 1
        last_in = setting[-1].out_channels
        layers.append(_cba(last_in, 6 * last_in, 1, 1, 1, norm_layer, nn.Hardswish))
        self.stages.append(nn.Sequential(*layers))
        self.out_channels.append(6 * last_in)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        if pretrained:
            if ckpt_path and os.path.exists(ckpt_path):
                logging.getLogger("root").info("load imagenet weights from %s", ckpt_path)
                self.load_state_dict(torch.load(ckpt_path, map_location="cpu"), strict=False)
            else:
                logging.getLogger("root").warning("pretrained backbone checkpoint %r not found; keeping random init "
                                                  "(no network fetch in pytorchocr_amd)", ckpt_path)

    def _pack(self, dev):
        p = {"stem": ops.PackedConv(self.conv1[0], self.conv1[1], dev, relu=ops.ACT_HSWISH), "stages": []}
        for stage in self.stages:
            blocks = []
            for m in stage:
                if isinstance(m, InvertedResidual):
                    blocks.append(("ir", m.pack(dev)))
                else:
                    blocks.append(("cba", ops.PackedConv(m[0], m[1], dev, relu=ops.ACT_HSWISH)))
            p["stages"].append(blocks)
        return p

    def forward_nhwc(self, x4):
        self._check_eval()
        p = self.packed()
        x = ops.conv2d(x4, p["stem"])
        outs = []
        for blocks in p["stages"]:
            for kind, bp in blocks:
                x = InvertedResidual.run(bp, x) if kind == "ir" else ops.conv2d(x, bp)
            outs.append(x)
        return outs

    def forward(self, x):
        feats = self.forward_nhwc(ops.nchw_to_nhwc(x, 4))
        return [ops.nhwc_to_nchw(f)[:, :c] for f, c in zip(feats, self.out_channels)]
