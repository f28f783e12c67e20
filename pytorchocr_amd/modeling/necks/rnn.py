"""CRNN sequence encoder on the HIP engine.

Mirror of reference pytocr/modeling/necks/rnn.py: `Im2Seq` (:4-15), `BidirectionalLSTM` (:18-36, nn.LSTM
bidirectional + optional Linear), `EncoderWithRNN` (:38-48, BiLSTM(in->H, Linear 2H->H) + BiLSTM(H->H)),
`SequenceEncoder` (:65-90).  Parameter names are the reference's (nn.LSTM / nn.Linear objects are kept as
parameter containers).  The sequence stays batch-major on the device (row = b*T + t), so Im2Seq is a view.
"""
import torch
from torch import nn

from .. import ops


class Im2Seq(nn.Module):
    def __init__(self, in_channels, **kwargs):
        super().__init__()
        self.out_channels = in_channels


class BidirectionalLSTM(nn.Module):
    def __init__(self, n_in, n_hidden, n_out=None):
        super().__init__()
        if n_hidden != 256 or n_in % 32:
            raise NotImplementedError("pytorchocr_amd BiLSTM kernel: hidden_size must be 256 and n_in a multiple of 32")
        self.rnn = nn.LSTM(n_in, n_hidden, bidirectional=True)
        self.out_channels = n_hidden * 2
        if n_out is not None:
            self.embedding = nn.Linear(n_hidden * 2, n_out)
            self.out_channels = n_out
        self.n_out = n_out

    def pack(self, dev):
        r = self.rnn
        f = lambda t: t.detach().float().cpu()
        w_ih = torch.cat([f(r.weight_ih_l0), f(r.weight_ih_l0_reverse)], 0).contiguous().to(dev)       # [8H, In]
        bias = torch.cat([f(r.bias_ih_l0) + f(r.bias_hh_l0), f(r.bias_ih_l0_reverse) + f(r.bias_hh_l0_reverse)]).contiguous().to(dev)
        w_hh = torch.stack([f(r.weight_hh_l0), f(r.weight_hh_l0_reverse)], 0).contiguous().to(dev)    # [2, 4H, H]
        p = {"w_ih": w_ih, "b": bias, "w_hh": w_hh}
        if self.n_out is not None:
            p["emb_w"] = f(self.embedding.weight).contiguous().to(dev)
            p["emb_b"] = f(self.embedding.bias).contiguous().to(dev)
        return p

    @staticmethod
    def run(p, x, B, T):
        """x f32[B*T, In] -> f32[B*T, out]"""
        xproj = ops.linear(x, p["w_ih"], p["b"])                 # [B*T, 8H] == [B][T][2][4H]
        out = ops.lstm_bidir(xproj, p["w_hh"], B, T)             # [B*T, 2H]
        if "emb_w" in p:
            out = ops.linear(out, p["emb_w"], p["emb_b"])
        return out


class EncoderWithRNN(nn.Module):
    def __init__(self, in_channels, hidden_size):
        super().__init__()
        self.out_channels = hidden_size * 2
        self.rnn = nn.Sequential(BidirectionalLSTM(in_channels, hidden_size, hidden_size),
                                 BidirectionalLSTM(hidden_size, hidden_size))


class SequenceEncoder(ops.PackedModule):
    def __init__(self, in_channels, encoder_type, hidden_size=256, **kwargs):
        super().__init__()
        self.encoder_reshape = Im2Seq(in_channels)
        self.out_channels = in_channels
        if encoder_type == "reshape":
            self.only_reshape = True
        elif encoder_type == "rnn":
            self.encoder = EncoderWithRNN(in_channels, hidden_size)
            self.out_channels = self.encoder.out_channels
            self.only_reshape = False
        else:
            raise NotImplementedError("pytorchocr_amd SequenceEncoder: encoder_type %r is not on the hot path" % encoder_type)

    def _pack(self, dev):
        if self.only_reshape:
            return []
        return [m.pack(dev) for m in self.encoder.rnn]

    def forward_seq(self, feat):
        """feat f32[B,1,T,C] NHWC (the VGG output, H == 1) -> (f32[B*T, C'], B, T), rows b*T + t."""
        self._check_eval()
        B, Hh, T, Cc = feat.shape
        assert Hh == 1, "the height of backbone output featuremap must be 1"
        x = feat.reshape(B * T, Cc)
        for p in self.packed():
            x = BidirectionalLSTM.run(p, x, B, T)
        return x, B, T

    def forward(self, x):
        """NCHW [B,C,1,W] -> [T,B,C'] (reference contract)."""
        x4 = ops.nchw_to_nhwc(x, x.shape[1])
        y, B, T = self.forward_seq(x4)
        return y.reshape(B, T, -1).permute(1, 0, 2).contiguous()
