"""CRNN VGG conv stack (v1) on the HIP conv engine.

Mirror of reference `VGG` (pytocr/modeling/backbones/rec_vgg.py:8-120), model_name "v1": 7 convs (all with
bias; BN after conv2/4/6), 4 max pools with the asymmetric (2,2)/(2,1)/(0,1) windows, H: 32 -> 1.
"""
from torch import nn

from .. import ops


class VGG(ops.PackedModule):
    def __init__(self, in_channels=3, model_name="v1", scale=1.0, leaky_relu=False, pretrained=False, ckpt_path=None, **kwargs):
        super().__init__()
        if model_name != "v1" or leaky_relu:
            raise NotImplementedError("pytorchocr_amd VGG: only model_name='v1' with ReLU is on the hot path")
        assert scale in (0.5, 1.0)
        nm = [64, 128, 256, 256, 512, 512, 512] if scale == 1.0 else [32, 64, 128, 128, 256, 256, 512]
        if nm[0] % 64:
            raise NotImplementedError("pytorchocr_amd VGG: scale=0.5 channel counts are not multiples of 64")
        ks = [3, 3, 3, 3, 3, 3, 2]
        ps = [1, 1, 1, 1, 1, 1, 0]
        cnn = nn.Sequential()
        for i in range(7):
            cnn.add_module("conv%d" % i, nn.Conv2d(in_channels if i == 0 else nm[i - 1], nm[i], ks[i], 1, ps[i]))
            if i in (2, 4, 6):
                cnn.add_module("batchnorm%d" % i, nn.BatchNorm2d(nm[i]))
            cnn.add_module("relu%d" % i, nn.ReLU(True))
            if i in (0, 1):
                cnn.add_module("pooling%d" % i, nn.MaxPool2d(2, 2))
            elif i in (3, 5):
                cnn.add_module("pooling%d" % (2 if i == 3 else 3), nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        self.cnn = cnn
        self.in_channels = in_channels
        self.out_channels = nm[-1]

    def _pack(self, dev):
        c = self.cnn
        bn = {2: c.batchnorm2, 4: c.batchnorm4, 6: c.batchnorm6}
        return [ops.PackedConv(getattr(c, "conv%d" % i), bn.get(i), dev, relu=True, cin_pad=4 if i == 0 else None)
                for i in range(7)]

    def forward_nhwc(self, x4):
        """x4 f32[B,32,W,4] -> f32[B,1,T,512]"""
        self._check_eval()
        p = self.packed()
        x = ops.conv3x3_relu_pool2(x4, p[0])                                   # conv0 + relu0 + pooling0, fused
        x = ops.conv2d(x, p[1]); x = ops.maxpool2d(x, 2, 2, 0)
        x = ops.conv2d(x, p[2])
        x = ops.conv2d(x, p[3]); x = ops.maxpool2d(x, (2, 2), (2, 1), (0, 1))
        x = ops.conv2d(x, p[4])
        x = ops.conv2d(x, p[5]); x = ops.maxpool2d(x, (2, 2), (2, 1), (0, 1))
        return ops.conv2d(x, p[6])

    def forward(self, x):
        return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x, 4)))
