"""CRNN VGG conv stacks (v1, v2) on the HIP conv engine.

Mirror of reference `VGG` (pytocr/modeling/backbones/rec_vgg.py:8-120).  model_name "v1": 7 convs (all with bias; BN after
conv2/4/6), 4 max pools with the asymmetric (2,2)/(2,1)/(0,1) windows, H: 32 -> 1.  model_name "v2" (:37-44, :62-76): a 5x5 / stride 2
first conv to 24 (32) channels, then every layer as depthwise k x k (bias, BN after layers 2/4/6, ReLU) + pointwise 1x1 (bias, the same
BN rule, ReLU), pools 1..3 of v1 (the stride-2 first conv stands in for pooling0) -- depthwise layers on ptocr_dwconv2_f32, the
rest on the conv engine.  Same module / parameter names as the reference (state_dict contract).
"""
from torch import nn

from .. import ops


class VGG(ops.PackedModule):
    def __init__(self, in_channels=3, model_name="v1", scale=1.0, leaky_relu=False, pretrained=False, ckpt_path=None, **kwargs):
        super().__init__()
        if model_name not in ("v1", "v2"):
            raise ValueError("supported vgg model are ['v1', 'v2'] but input model_name is %r" % (model_name,))
        if scale not in (0.5, 1.0):
            raise ValueError("supported scale are [0.5, 1.0] but input scale is %r" % (scale,))
        if leaky_relu:
            raise NotImplementedError("pytorchocr_amd VGG: leaky_relu=True is not on the hot path (the yml's use ReLU)")
        self.model_name = model_name
        cnn = nn.Sequential()
        if model_name == "v1":
            nm = [64, 128, 256, 256, 512, 512, 512] if scale == 1.0 else [32, 64, 128, 128, 256, 256, 512]
            ks = [3, 3, 3, 3, 3, 3, 2]
            ps = [1, 1, 1, 1, 1, 1, 0]
            for i in range(7):
                cnn.add_module("conv%d" % i, nn.Conv2d(in_channels if i == 0 else nm[i - 1], nm[i], ks[i], 1, ps[i]))
                if i in (2, 4, 6):
                    cnn.add_module("batchnorm%d" % i, nn.BatchNorm2d(nm[i]))
                cnn.add_module("relu%d" % i, nn.ReLU(True))
                if i in (0, 1):
                    cnn.add_module("pooling%d" % i, nn.MaxPool2d(2, 2))
                elif i in (3, 5):
                    cnn.add_module("pooling%d" % (2 if i == 3 else 3), nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        else:
            nm = [24, 128, 256, 256, 512, 512, 512] if scale == 1.0 else [32, 64, 128, 128, 256, 256, 256]
            ks = [5, 3, 3, 3, 3, 3, 2]
            ps = [2, 1, 1, 1, 1, 1, 0]
            cnn.add_module("conv_0", nn.Conv2d(in_channels, nm[0], 5, 2, 2))
            cnn.add_module("relu_0", nn.ReLU(True))
            for i in range(1, 7):
                n_in, bn = nm[i - 1], i in (2, 4, 6)
                cnn.add_module("conv%d" % i, nn.Conv2d(n_in, n_in, ks[i], 1, ps[i], groups=n_in))
                if bn:
                    cnn.add_module("batchnorm%d" % i, nn.BatchNorm2d(n_in))
                cnn.add_module("relu%d" % i, nn.ReLU(True))
                cnn.add_module("convproject%d" % i, nn.Conv2d(n_in, nm[i], 1, 1, 0))
                if bn:
                    cnn.add_module("batchnormproject%d" % i, nn.BatchNorm2d(nm[i]))
                cnn.add_module("reluproject%d" % i, nn.ReLU(True))
                if i == 1:
                    cnn.add_module("pooling1", nn.MaxPool2d(2, 2))
                elif i in (3, 5):
                    cnn.add_module("pooling%d" % (2 if i == 3 else 3), nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        self.cnn = cnn
        self.in_channels = in_channels
        self.out_channels = nm[-1]

    def _pack(self, dev):
        c = self.cnn
        if self.model_name == "v1":
            bn = {2: c.batchnorm2, 4: c.batchnorm4, 6: c.batchnorm6}
            return [ops.PackedConv(getattr(c, "conv%d" % i), bn.get(i), dev, relu=True, cin_pad=4 if i == 0 else None)
                    for i in range(7)]
        p = [ops.PackedConv(c.conv_0, None, dev, relu=True, cin_pad=4)]
        for i in range(1, 7):
            bn = i in (2, 4, 6)
            p.append((ops.PackedDW(getattr(c, "conv%d" % i), getattr(c, "batchnorm%d" % i) if bn else None, dev, ops.ACT_RELU),
                      ops.PackedConv(getattr(c, "convproject%d" % i), getattr(c, "batchnormproject%d" % i) if bn else None, dev, relu=True)))
        return p

    def forward_nhwc(self, x4):
        """x4 f32[B,32,W,4] -> f32[B,1,T,512]"""
        self._check_eval()
        p = self.packed()
        if self.model_name == "v2":
            x = ops.conv2d(x4, p[0])                                           # 5x5 / s2: [B,16,W/2,32 (24 real)]
            for i in range(1, 7):
                x = ops.conv2d(ops.dwconv(x, p[i][0]), p[i][1])
                if i == 1:
                    x = ops.maxpool2d(x, 2, 2, 0)
                elif i in (3, 5):
                    x = ops.maxpool2d(x, (2, 2), (2, 1), (0, 1))
            return x
        x = ops.conv3x3_relu_pool2(x4, p[0])                                   # conv0 + relu0 + pooling0, fused
        x = ops.conv2d_relu_pool2(x, p[1])                                     # conv1 + relu1 + pooling1, fused when the layer runs on the F(4x4) kernel
        x = ops.conv2d(x, p[2])
        x = ops.conv2d(x, p[3]); x = ops.maxpool2d(x, (2, 2), (2, 1), (0, 1))
        x = ops.conv2d(x, p[4])
        x = ops.conv2d(x, p[5]); x = ops.maxpool2d(x, (2, 2), (2, 1), (0, 1))
        return ops.conv2d(x, p[6])

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x, 4)))
