"""DB++ Adaptive Scale Fusion on the HIP engine: attention_type "scale_channel_spatial" (the yml default), "scale_spatial" and
"scale_channel" (pytocr/modeling/necks/asf.py:9-29, 78-107).

Mirror of reference `ScaleFeatureSelection` / `ScaleChannelSpatialAttention` (pytocr/modeling/necks/asf.py:32-75,
110-162): 3x3 conv 256->64 WITH bias (MFMA), channel gate (avgpool -> 1x1 -> ReLU -> 1x1 -> sigmoid) ADDED to x,
spatial gate (mean over C -> 3x3(1->1) -> ReLU -> 1x1 -> sigmoid) ADDED again, 1x1 64->4 + sigmoid, and
out = concat_i(score_i * feat_i).  Parameter names are the reference's.  Because `fuse` already is the
concatenation of the four (upsampled) features, the re-weighting happens in place on its channel slices.
"""
import ctypes as C

import torch
from torch import nn

from .. import ops
from ... import _lib


class ScaleChannelSpatialAttention(nn.Module):
    def __init__(self, in_channels, mid_channels, num_features):
        super().__init__()
        self.channel_wise = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, mid_channels, 1, bias=False), nn.ReLU(),
                                          nn.Conv2d(mid_channels, in_channels, 1, bias=False), nn.Sigmoid())
        self.spatial_wise = nn.Sequential(nn.Conv2d(1, 1, 3, padding=1, bias=False), nn.ReLU(), nn.Conv2d(1, 1, 1, bias=False), nn.Sigmoid())
        self.attention_wise = nn.Sequential(nn.Conv2d(in_channels, num_features, 1, bias=False), nn.Sigmoid())


class ScaleSpatialAttention(nn.Module):
    def __init__(self, in_channels, num_features):
        super().__init__()
        self.spatial_wise = nn.Sequential(nn.Conv2d(1, 1, 3, padding=1, bias=False), nn.ReLU(), nn.Conv2d(1, 1, 1, bias=False), nn.Sigmoid())
        self.attention_wise = nn.Sequential(nn.Conv2d(in_channels, num_features, 1, bias=False), nn.Sigmoid())


class ScaleChannelAttention(nn.Module):
    def __init__(self, in_channels, mid_channels, num_features):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(in_channels, mid_channels, 1, bias=False)
        self.bn = nn.BatchNorm2d(mid_channels)
        self.fc2 = nn.Conv2d(mid_channels, num_features, 1, bias=False)


class ScaleFeatureSelection(nn.Module):
    def __init__(self, in_channels, inter_channels, out_features_num=4, attention_type="scale_spatial"):
        super().__init__()
        if attention_type not in ("scale_channel_spatial", "scale_spatial", "scale_channel"):
            # the reference builds no attention for another name and fails at the first forward (asf.py:124-133, 148)
            raise ValueError("ASF attention_type must be scale_spatial, scale_channel_spatial or scale_channel, got %r" % (attention_type,))
        if (in_channels, inter_channels, out_features_num) != (256, 64, 4):
            raise NotImplementedError("pytorchocr_amd ASF kernels are specialised for 256 -> 64 channels, 4 levels")
        self.conv = nn.Conv2d(in_channels, inter_channels, 3, padding=1)
        self.type = attention_type
        if attention_type == "scale_spatial":
            self.enhanced_attention = ScaleSpatialAttention(inter_channels, out_features_num)
        elif attention_type == "scale_channel_spatial":
            self.enhanced_attention = ScaleChannelSpatialAttention(inter_channels, inter_channels // 4, out_features_num)
        else:
            self.enhanced_attention = ScaleChannelAttention(inter_channels, inter_channels // 2, out_features_num)
        self.out_features_num = out_features_num

    def pack(self, dev):
        a = self.enhanced_attention
        f = lambda t: t.detach().float().cpu().contiguous()
        p = {"conv": ops.PackedConv(self.conv, None, dev, relu=False), "type": self.type}
        if self.type == "scale_channel":
            w1, b1 = ops.fold_bn(a.fc1.weight, None, a.bn)              # [32, 64, 1, 1], [32] in float64
            p.update(w1=w1.reshape(32, 64).float().contiguous().to(dev), b1=b1.float().contiguous().to(dev), w2=f(a.fc2.weight).reshape(4, 32).to(dev))
            return p
        if self.type == "scale_channel_spatial":
            p.update(cw1=f(a.channel_wise[1].weight).reshape(16, 64).to(dev), cw2=f(a.channel_wise[3].weight).reshape(64, 16).to(dev))
        p.update(sp3=f(a.spatial_wise[0].weight).reshape(9).to(dev), sp1=float(a.spatial_wise[2].weight.detach().reshape(-1)[0]),
                 att=f(a.attention_wise[0].weight).reshape(4, 64).to(dev))
        return p

    @staticmethod
    def run(p, fuse):
        N, H, W, _ = fuse.shape
        y = ops.conv2d(fuse, p["conv"])
        L = _lib.lib()
        L.ptocr_asf_work_floats.restype = C.c_long
        work = torch.empty(L.ptocr_asf_work_floats(N, H, W), dtype=torch.float32, device=fuse.device)
        if p["type"] == "scale_channel":
            _lib.check(L.ptocr_asf_scale_channel_f32(_lib.ptr(y), _lib.ptr(fuse), _lib.ptr(p["w1"]), _lib.ptr(p["b1"]), _lib.ptr(p["w2"]), _lib.ptr(work),
                                                     N, H, W, _lib.cur_stream()), "ptocr_asf_scale_channel_f32")
        elif p["type"] == "scale_spatial":
            _lib.check(L.ptocr_asf_scale_spatial_f32(_lib.ptr(y), _lib.ptr(fuse), _lib.ptr(p["sp3"]), C.c_float(p["sp1"]), _lib.ptr(p["att"]),
                                                     _lib.ptr(work), N, H, W, _lib.cur_stream()), "ptocr_asf_scale_spatial_f32")
        else:
            _lib.check(L.ptocr_asf_scale_channel_spatial_f32(_lib.ptr(y), _lib.ptr(fuse), _lib.ptr(p["cw1"]), _lib.ptr(p["cw2"]),
                                                             _lib.ptr(p["sp3"]), C.c_float(p["sp1"]), _lib.ptr(p["att"]), _lib.ptr(work),
                                                             N, H, W, _lib.cur_stream()), "ptocr_asf_scale_channel_spatial_f32")
        return fuse

    @staticmethod
    def run_pyramid(p, pyr):
        """The same on an ops.Pyramid (round 6): the 3x3 conv reads the four planes in place (ptocr_conv3x3_wino4r_pyramid_f32), the
        re-weighting reads them again and WRITES the concat once -- the upsampled copies the in-place form re-weights never exist.
        Returns the re-weighted f32[N,H,W,256]; bit-identical to run(p, pyr.materialize())."""
        N, H, W = pyr.N, pyr.H, pyr.W
        y = ops.conv3x3_pyramid(pyr, p["conv"])
        L = _lib.lib()
        L.ptocr_asf_work_floats.restype = C.c_long
        work = torch.empty(L.ptocr_asf_work_floats(N, H, W), dtype=torch.float32, device=pyr.buf.device)
        out = torch.empty((N, H, W, 256), dtype=torch.float32, device=pyr.buf.device)
        off, sh = (C.c_longlong * 4)(*pyr.offs), (C.c_int * 4)(*pyr.shifts)
        kind = {"scale_channel_spatial": 0, "scale_spatial": 1, "scale_channel": 2}[p["type"]]
        null = C.c_void_p(0)
        if kind == 2:
            wa, wb, w3, w1x1, watt = _lib.ptr(p["w1"]), _lib.ptr(p["b1"]), null, 0.0, _lib.ptr(p["w2"])
        else:
            wa, wb = (_lib.ptr(p["cw1"]), _lib.ptr(p["cw2"])) if kind == 0 else (null, null)
            w3, w1x1, watt = _lib.ptr(p["sp3"]), p["sp1"], _lib.ptr(p["att"])
        _lib.check(L.ptocr_asf_pyramid_f32(kind, _lib.ptr(y), _lib.ptr(pyr.buf), off, sh, C.c_longlong(pyr.buf.numel()), _lib.ptr(out), wa, wb, w3,
                                           C.c_float(w1x1), watt, _lib.ptr(work), N, H, W, _lib.cur_stream()), "ptocr_asf_pyramid_f32")
        return out
