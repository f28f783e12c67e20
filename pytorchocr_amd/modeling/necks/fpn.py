"""DB-style FPN neck on the HIP conv engine.

Mirror of reference `FPN` (pytocr/modeling/necks/fpn.py:8-134), mode "DB": 4 lateral 1x1 conv+BN+ReLU,
top-down nearest-x2 upsample-add, 4 smoothing 3x3 conv+BN+ReLU to C/4, nearest x8/x4/x2, concat (p5,p4,p3,p2).
Fusions: the upsample-add runs in the lateral conv's epilogue (add AFTER its ReLU, fpn.py:133-134), and the
smoothing convs store their nearest-upsampled result straight into their channel slice of the `fuse` tensor,
so interpolate + cat never exist as separate passes.
"""
import torch
from torch import nn

from .. import ops


def _cbr(cin, cout, k, pad):
    return nn.Sequential(nn.Conv2d(cin, cout, k, 1, pad, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class FPN(ops.PackedModule):
    def __init__(self, in_channels, out_channels=256, mode=None, use_asf=False, attention_type="scale_spatial", **kwargs):
        super().__init__()
        if mode != "DB":
            raise NotImplementedError("pytorchocr_amd FPN: only mode='DB' is on the hot path")
        if out_channels % 32 != 0 or (out_channels // 4) % 4 != 0:
            raise NotImplementedError("pytorchocr_amd FPN: out_channels must be a multiple of 32 (got %d)" % out_channels)
        self.mode, self.use_asf = mode, use_asf
        self.in5 = _cbr(in_channels[-1], out_channels, 1, 0)
        self.in4 = _cbr(in_channels[-2], out_channels, 1, 0)
        self.in3 = _cbr(in_channels[-3], out_channels, 1, 0)
        self.in2 = _cbr(in_channels[-4], out_channels, 1, 0)
        sm = out_channels // 4
        self.out_channels = out_channels
        self.out5 = _cbr(out_channels, sm, 3, 1)
        self.out4 = _cbr(out_channels, sm, 3, 1)
        self.out3 = _cbr(out_channels, sm, 3, 1)
        self.out2 = _cbr(out_channels, sm, 3, 1)
        if self.use_asf:    # DB++
            from .asf import ScaleFeatureSelection
            self.concat_attention = ScaleFeatureSelection(out_channels, sm, attention_type=attention_type)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.normal_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1.)
                m.bias.data.fill_(1e-4)

    def _pack(self, dev):
        p = {k: ops.PackedConv(getattr(self, k)[0], getattr(self, k)[1], dev, relu=True)
             for k in ("in5", "in4", "in3", "in2", "out5", "out4", "out3", "out2")}
        if self.use_asf:
            p["asf"] = self.concat_attention.pack(dev)
        return p

    def forward_nhwc(self, feats, pyramid_for=None):
        """pyramid_for: the PackedConv that will consume the result (DBHead's first conv); when it can read a pyramid the result is an
        ops.Pyramid instead of the concat tensor"""
        self._check_eval()
        p = self.packed()
        c2, c3, c4, c5 = feats
        in5 = ops.conv2d(c5, p["in5"])
        out4 = ops.conv2d(c4, p["in4"], res=in5, res_mode=ops.RES_ADD_UP2_POST_RELU)
        out3 = ops.conv2d(c3, p["in3"], res=out4, res_mode=ops.RES_ADD_UP2_POST_RELU)
        out2 = ops.conv2d(c2, p["in2"], res=out3, res_mode=ops.RES_ADD_UP2_POST_RELU)
        N, H4, W4, _ = c2.shape
        sm = self.out_channels // 4
        # Round 5: the four smoothing convs store at their OWN resolution into one allocation and the consumer reads the pyramid in place:
        # the x8 / x4 / x2 upsampled copies (3 x 482 MB per 32 images at 736x1280) are never written.  The consumer is the head's first conv
        # (DB) or -- round 6 -- the ASF (DB++): its 3x3 conv reads the planes, its re-weighting reads them again and writes the concat once.
        consumer = p["asf"]["conv"] if self.use_asf else pyramid_for
        if consumer is not None and sm == 64 and ops.pyramid_conv_ok(consumer, N, H4, W4):
            pyr = ops.Pyramid(N, H4, W4, (3, 2, 1, 0), c2.device)
            ops.conv2d(in5, p["out5"], out=pyr.plane(0), store=sm)
            ops.conv2d(out4, p["out4"], out=pyr.plane(1), store=sm)
            ops.conv2d(out3, p["out3"], out=pyr.plane(2), store=sm)
            ops.conv2d(out2, p["out2"], out=pyr.plane(3), store=sm)
            if self.use_asf:
                return self.concat_attention.run_pyramid(p["asf"], pyr)
            return pyr
        fuse = torch.empty((N, H4, W4, self.out_channels), dtype=torch.float32, device=c2.device)
        ops.conv2d(in5, p["out5"], out=fuse, out_up=8, out_coff=0, store=sm)
        ops.conv2d(out4, p["out4"], out=fuse, out_up=4, out_coff=sm, store=sm)
        ops.conv2d(out3, p["out3"], out=fuse, out_up=2, out_coff=2 * sm, store=sm)
        ops.conv2d(out2, p["out2"], out=fuse, out_up=1, out_coff=3 * sm, store=sm)
        if self.use_asf:
            self.concat_attention.run(p["asf"], fuse)          # re-weights the four 64-channel slices in place
        return fuse

    def forward(self, x):
        feats = [ops.nchw_to_nhwc(f, f.shape[1]) for f in x]
        return ops.nhwc_to_nchw(self.forward_nhwc(feats))
