"""CTC head on the HIP engine.  Mirror of reference `CTCHead` (pytocr/modeling/heads/rec_ctc_head.py:5-36):
Linear(in -> out_channels) over all T*B rows (MFMA GEMM), softmax(dim=2) in eval.  The fused decode path
(`greedy`) never materialises the softmax tensor: arg-max and max-probability come straight from the logits."""
import torch
from torch import nn

from .. import ops


class CTCHead(ops.PackedModule):
    def __init__(self, in_channels, out_channels, return_feats=False, **kwargs):
        super().__init__()
        self.fc = nn.Linear(in_channels, out_channels)
        self.out_channels = out_channels
        self.return_feats = return_feats

    def _pack(self, dev):
        C, K = self.fc.weight.shape
        Cp = (C + 127) // 128 * 128               # 128: the fused FC + arg-max kernel runs 128-column tiles
        w = torch.zeros(Cp, K)
        w[:C] = self.fc.weight.detach().float().cpu()
        b = torch.zeros(Cp)
        b[:C] = self.fc.bias.detach().float().cpu()
        return {"w": w.contiguous().to(dev), "b": b.contiguous().to(dev), "C": C, "Cp": Cp}

    def logits(self, x):
        """x f32[B*T, K] -> logits f32[B*T, Cp] (first C columns valid)"""
        self._check_eval()
        p = self.packed()
        return ops.linear(x, p["w"], p["b"]), p["C"]

    def forward_seq(self, seq):
        """(x[B*T,K], B, T) -> softmax f32[T,B,C] (reference contract, rec_ctc_head.py:32-36)"""
        x, B, T = seq
        lg, C = self.logits(x)
        pr = ops.softmax_rows(lg, C)                              # [B*T, C]
        return pr.reshape(B, T, C).permute(1, 0, 2).contiguous()

    def greedy(self, seq):
        """(x[B*T,K], B, T) -> (idx int32[B,T], prob f32[B,T]) without the softmax tensor"""
        x, B, T = seq
        if ops.FUSE_CTC:
            self._check_eval()
            p = self.packed()
            idx, prob = ops.linear_ctc_greedy(x, p["w"], p["b"], p["C"])     # arg-max / sum-exp in the FC's epilogue
        else:
            lg, C = self.logits(x)
            idx, prob = ops.ctc_greedy(lg, C, is_prob=False)
        return idx.reshape(B, T), prob.reshape(B, T)

    def forward(self, x, **kwargs):
        T, B, K = x.shape
        xb = x.permute(1, 0, 2).contiguous().reshape(B * T, K)
        return self.forward_seq((xb, B, T))
