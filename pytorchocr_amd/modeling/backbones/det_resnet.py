"""ResNet-18/34 (BasicBlock) and ResNet-50/101/152 (Bottleneck, stride on the 3x3: "ResNet V1.5") detection backbones on the HIP
conv engine, 7x7 stem or the three-conv 3x3 stem (mode_3x3).

Mirror of the reference class `ResNet` (pytocr/modeling/backbones/det_resnet.py:143-312): same constructor
arguments, same parameter/buffer names (state_dict contract), same four outputs
(C2..C5).  nn.Conv2d / nn.BatchNorm2d objects are kept only as parameter containers; forward runs
BN-folded fp32 MFMA convolutions with fused ReLU / residual-add epilogues on NHWC activations.
"""
import logging
import os

import torch
from torch import nn

from .. import ops


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def pack(self, dev):
        p = {"c1": ops.PackedConv(self.conv1, self.bn1, dev, relu=True),
             "c2": ops.PackedConv(self.conv2, self.bn2, dev, relu=True)}
        if self.downsample is not None:
            p["ds"] = ops.PackedConv(self.downsample[0], self.downsample[1], dev, relu=False)
        return p

    @staticmethod
    def run(p, x):
        out = ops.conv2d(x, p["c1"])
        idt = ops.conv2d(x, p["ds"]) if "ds" in p else x
        # conv2 + bn2, += identity, ReLU  (reference det_resnet.py:72-80) in one epilogue
        return ops.conv2d(out, p["c2"], res=idt, res_mode=ops.RES_ADD_PRE_RELU)


class Bottleneck(nn.Module):
    """reference det_resnet.py:85-140 (groups = 1, base width 64, no dilation): 1x1 -> 3x3 (stride here) -> 1x1 (x4), residual, ReLU"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def pack(self, dev):
        p = {"c1": ops.PackedConv(self.conv1, self.bn1, dev, relu=True), "c2": ops.PackedConv(self.conv2, self.bn2, dev, relu=True),
             "c3": ops.PackedConv(self.conv3, self.bn3, dev, relu=True)}
        if self.downsample is not None:
            p["ds"] = ops.PackedConv(self.downsample[0], self.downsample[1], dev, relu=False)
        return p

    @staticmethod
    def run(p, x):
        out = ops.conv2d(ops.conv2d(x, p["c1"]), p["c2"])
        idt = ops.conv2d(x, p["ds"]) if "ds" in p else x
        # conv3 + bn3, += identity, ReLU (det_resnet.py:128-138) in one epilogue
        return ops.conv2d(out, p["c3"], res=idt, res_mode=ops.RES_ADD_PRE_RELU)


class ResNet(ops.PackedModule):
    def __init__(self, in_channels=3, layers=50, mode_3x3=False, pretrained=False, ckpt_path=None, **kwargs):
        super().__init__()
        table = {18: ([2, 2, 2, 2], BasicBlock), 34: ([3, 4, 6, 3], BasicBlock), 50: ([3, 4, 6, 3], Bottleneck),
                 101: ([3, 4, 23, 3], Bottleneck), 152: ([3, 8, 36, 3], Bottleneck)}
        if layers not in table:
            raise ValueError("ResNet layers must be one of %s, got %r" % (sorted(table), layers))
        # every option against ITS OWN default (reference det_resnet.py:143-150): a grouped, wide or dilated ResNet is not built, and a
        # value that belongs to another key (groups=64, width_per_group=1) must not slip through as "some default"
        defaults = {"groups": (None, 1), "width_per_group": (None, 64), "norm_layer": (None,),
                    "replace_stride_with_dilation": (None, [False, False, False], (False, False, False)), "zero_init_residual": (None, False)}
        for k, allowed in defaults.items():
            if k in kwargs and not any(kwargs[k] is v or kwargs[k] == v for v in allowed):
                raise NotImplementedError("pytorchocr_amd ResNet: %s=%r is not built (plain ResNet with the reference's default initialisation only)" % (k, kwargs[k]))
        depth, self.block = table[layers]
        self.mode_3x3 = bool(mode_3x3)
        if not self.mode_3x3:                                  # 7x7 kernel
            self.inplanes = 64
            self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
        else:                                                  # three 3x3 convs (det_resnet.py:196-206)
            self.inplanes = 128
            self.conv1_1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
            self.bn1_1 = nn.BatchNorm2d(64)
            self.conv1_2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False)
            self.bn1_2 = nn.BatchNorm2d(64)
            self.conv1_3 = nn.Conv2d(64, 128, 3, 1, 1, bias=False)
            self.bn1_3 = nn.BatchNorm2d(128)
        e = self.block.expansion
        self.layer1 = self._make_layer(64, depth[0], 1)
        self.layer2 = self._make_layer(128, depth[1], 2)
        self.layer3 = self._make_layer(256, depth[2], 2)
        self.layer4 = self._make_layer(512, depth[3], 2)
        self.out_channels = [64 * e, 128 * e, 256 * e, 512 * e]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if pretrained:
            # local files only: the reference falls back to a URL fetch (det_resnet.py:246-254); never here
            if ckpt_path and os.path.exists(ckpt_path):
                logging.getLogger("root").info("load imagenet weights from %s", ckpt_path)
                self.load_state_dict(torch.load(ckpt_path, map_location="cpu"), strict=False)
            else:
                logging.getLogger("root").warning("pretrained backbone checkpoint %r not found; keeping random init "
                                                  "(no network fetch in pytorchocr_amd)", ckpt_path)

    def _make_layer(self, planes, blocks, stride):
        block, downsample = self.block, None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _pack(self, dev):
        if self.mode_3x3:
            p = {"stem3": [ops.PackedConv(self.conv1_1, self.bn1_1, dev, relu=True, cin_pad=4), ops.PackedConv(self.conv1_2, self.bn1_2, dev, relu=True),
                           ops.PackedConv(self.conv1_3, self.bn1_3, dev, relu=True)]}
        else:
            p = {"stem": ops.PackedConv(self.conv1, self.bn1, dev, relu=True, cin_pad=4)}
        for li in (1, 2, 3, 4):
            p["layer%d" % li] = [blk.pack(dev) for blk in getattr(self, "layer%d" % li)]
        return p

    def forward_nhwc(self, x4, x_nchw=None):
        """x4: f32[N,H,W,4] (RGB + zero channel) -> [C2, C3, C4, C5] NHWC; or the model's NCHW input as `x_nchw` (x4 None)."""
        self._check_eval()
        p = self.packed()
        if self.mode_3x3:
            x = x4 if x_nchw is None else ops.nchw_to_nhwc(x_nchw, 4)
            for pc in p["stem3"]:
                x = ops.conv2d(x, pc)
            x = ops.maxpool2d(x, 3, 2, 1)
        else:
            x = ops.stem_relu_pool(x4, x_nchw, p["stem"])         # stem + ReLU + max pool in one kernel where it applies
            if x is None:
                x = ops.conv2d(x4, p["stem"]) if x_nchw is None else ops.stem_from_nchw(x_nchw, p["stem"])
                x = ops.maxpool2d(x, 3, 2, 1)
        outs = []
        for li in (1, 2, 3, 4):
            for bp in p["layer%d" % li]:
                x = self.block.run(bp, x)
            outs.append(x)
        return outs

    def forward(self, x):
        """NCHW in, list of NCHW feature maps out (reference contract, det_resnet.py:282-312)."""
        feats = self.forward_nhwc(None, x_nchw=x)
        return [ops.nhwc_to_nchw(f) for f in feats]

    def forward_from_nchw(self, x):
        """NCHW model input -> NHWC feature maps (the stem reads the planes itself: no boundary layout pass)"""
        return self.forward_nhwc(None, x_nchw=x)
