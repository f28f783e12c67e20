"""ClsHead: mirror of reference pytocr/modeling/heads/cls_head.py:5-29 (global average pool -> Linear -> softmax in eval).

On the device the backbone's AvgPool2d(2, 2), this head's AdaptiveAvgPool2d(1), the Linear and the softmax are ONE kernel
(`ptocr_cls_head_f32`): a mean of equal 2x2 block means is the mean over the blocks' pixels."""
import torch
from torch import nn

from .. import ops


class ClsHead(nn.Module):
    def __init__(self, in_channels, class_dim, **kwargs):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(in_channels, class_dim)
        self._packed = None

    def _pack(self, cp, dev):
        sig = (self.fc.weight._version, self.fc.weight.data_ptr(), self.fc.bias._version, cp, str(dev))
        if self._packed is None or self._packed[0] != sig:
            w = torch.zeros((self.fc.out_features, cp), dtype=torch.float32, device=dev)
            w[:, :self.fc.in_features] = self.fc.weight.detach().float()
            self._packed = (sig, w.contiguous(), self.fc.bias.detach().float().contiguous().to(dev))
        return self._packed[1], self._packed[2]

    def forward_nhwc(self, feat):
        """feat: the backbone's un-pooled `features` output f32[N,H,W,Cp] -> softmax f32[N, class_dim]"""
        if self.training:
            raise NotImplementedError("pytorchocr_amd implements the inference (eval) hot path only; call .eval()")
        w, b = self._pack(int(feat.shape[3]), feat.device)
        return ops.cls_head(feat, w, b)

    def forward(self, x, **kwargs):
        """x: NCHW features AFTER the backbone's pool (the reference's tensor at this boundary); a 1x1-block mean of it"""
        if not x.is_cuda:
            raise RuntimeError("pytorchocr_amd ClsHead.forward: input is on %s; the HIP path has no CPU fallback" % x.device)
        n, c, h, w_ = x.shape
        f = ops.nchw_to_nhwc(x.float().contiguous(), (c + 3) // 4 * 4)
        # the fused kernel pools 2x2 blocks first; doubling the map restates "already pooled" exactly (every value twice per axis)
        f = f.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
        return self.forward_nhwc(f)
