"""bf16 inference path of the MobileNetV3 detector (BASELINE.json configs[3]: DBNet mbv3-small x1.0, bf16).

`BaseModel.set_compute_dtype("bf16")` (the counterpart of calling `.half()` / autocast on the reference module) routes a
det model made of MobileNetV3 + FPN(DB) + DBHead through `Mbv3DbBf16`: the same parameters (reference state_dict contract),
BatchNorm folded in float64, weights rounded to bf16 once, activations bf16 NHWC in HBM, fp32 accumulation, fp32 probability
maps out.  Layers of the reference graph and the kernels that run them (csrc/bf16_ops.hip):

  stem ConvBNActivation 3x3/s2 (det_mobilenet_v3.py:205-207)        ptocr_stem3x3s2_bf16 (reads the NCHW fp32 input itself)
  InvertedResidual (:106-151): 1x1 expand + act                      ptocr_pwconv_bf16
      depthwise k x k + act, SE average pool                         ptocr_dwconv_bf16 (pool partial sums in the same pass)
      SE gate fc1 / ReLU / fc2 / hardsigmoid (:76-85)                ptocr_se_fc_f32
      SE multiply + 1x1 project (+ identity)                         ptocr_pwconv_bf16 (gate applied to its input fragments)
  last 1x1 conv 96 -> 576 + Hardswish                                ptocr_pwconv_bf16
  FPN laterals + top-down add (fpn.py:102-113), smoothing 3x3
      + nearest upsample + concat (:115-131)                          ptocr_pwconv_bf16 (res_mode 2), ptocr_conv3x3_bf16 (out_up, coff)
  DBHead binarize (det_db_head.py:9-17)                              ptocr_conv3x3_bf16, ptocr_db_head_tail_bf16

Tolerance: bf16 keeps 8 significant bits, so the 1e-4 bar of the fp32 path cannot hold; tests/test_gpu_bf16.py states and checks
what does (max |p_bf16 - p_fp32| on the probability maps, and the share of pixels whose side of the 0.3 threshold changes)."""
import ctypes as C
import os

import torch

from .. import _lib
from . import ops


# bench.py sets this to 0 to have every launch wrapper add the bytes its kernel moves (input + output activations, weights)
TRAFFIC = None


def _count(*tensors):
    global TRAFFIC
    if TRAFFIC is not None:
        TRAFFIC += sum(t.numel() * t.element_size() for t in tensors if t is not None)


def _r16(c):
    return (c + 15) // 16 * 16


def _r32(c):
    return (c + 31) // 32 * 32


def _bf(t, dev):
    return t.to(torch.bfloat16).contiguous().to(dev)


def _f(t, dev):
    return t.float().contiguous().to(dev)


class _Pw:
    """1x1 conv (+BN): weights bf16[Cout_pad][Cin_pad], bias f32[Cout_pad]"""

    def __init__(self, conv, bn, dev, act):
        w, b = ops.fold_bn(conv.weight, conv.bias, bn)
        cout, cin = w.shape[0], w.shape[1]
        self.cin, self.cstore, self.cpad = _r16(cin), _r16(cout), _r32(cout)
        wp = torch.zeros(self.cpad, self.cin, dtype=torch.float64)
        wp[:cout, :cin] = w.reshape(cout, cin)
        bp = torch.zeros(self.cpad, dtype=torch.float64)
        bp[:cout] = b
        self.w, self.b, self.act = _bf(wp, dev), _f(bp, dev), ops._act_code(act)


class _C3:
    """3x3/s1/p1 conv (+BN) with <= 32 outputs: weights bf16[32][9 * Cin_pad]"""

    def __init__(self, conv, bn, dev, act):
        w, b = ops.fold_bn(conv.weight, conv.bias, bn)
        cout, cin = w.shape[0], w.shape[1]
        if cout > 32:
            raise NotImplementedError("bf16 path: 3x3 convs with more than 32 output channels are not built (got %d)" % cout)
        self.cin, self.cout = _r16(cin), cout
        wp = torch.zeros(32, 9, self.cin, dtype=torch.float64)
        wp[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin)
        bp = torch.zeros(32, dtype=torch.float64)
        bp[:cout] = b
        self.w, self.b, self.act = _bf(wp.reshape(32, 9 * self.cin), dev), _f(bp, dev), ops._act_code(act)


class _Dw:
    def __init__(self, conv, bn, dev, act):
        w, b = ops.fold_bn(conv.weight, conv.bias, bn)              # [C, 1, k, k]
        c, k = w.shape[0], w.shape[2]
        self.c, self.k, self.stride = _r16(c), k, conv.stride[0]
        wp = torch.zeros(k * k, self.c, dtype=torch.float64)
        wp[:, :c] = w[:, 0].reshape(c, k * k).t()
        bp = torch.zeros(self.c, dtype=torch.float64)
        bp[:c] = b
        self.w, self.b, self.act = _f(wp, dev), _f(bp, dev), ops._act_code(act)


class _Se:
    def __init__(self, se, dev, cpad):
        w1 = se.fc1.weight.detach().double().cpu()[:, :, 0, 0]      # [S, C]
        w2 = se.fc2.weight.detach().double().cpu()[:, :, 0, 0]      # [C, S]
        s, c = w1.shape
        w1p = torch.zeros(s, cpad, dtype=torch.float64); w1p[:, :c] = w1
        w2p = torch.zeros(cpad, s, dtype=torch.float64); w2p[:c] = w2
        b2p = torch.full((cpad,), -3.0, dtype=torch.float64); b2p[:c] = se.fc2.bias.detach().double().cpu()
        self.w1, self.b1, self.w2, self.b2, self.s, self.c = _f(w1p, dev), _f(se.fc1.bias.detach().cpu(), dev), _f(w2p, dev), _f(b2p, dev), s, cpad
        self.w1t, self.w2t = _f(w1p.t().contiguous(), dev), _f(w2p.t().contiguous(), dev)      # [C][S], [S][C]: the gate kernel's coalesced form


SE_SPLIT = os.environ.get("PTOCR_SE_SPLIT", "0") != "0"          # 1: the wide gates (C >= 256) as the round-4 pair of split launches; default (round 6): one 1024-thread block per image for every gate
SE_SPLIT_MIN_C = int(os.environ.get("PTOCR_SE_SPLIT_MIN_C", "256"))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def pwconv(x, pw, res=None, res_mode=0, scale=None, out=None, coff=0):
    n, h, w_, cin = x.shape
    assert cin == pw.cin, (cin, pw.cin)
    if out is None:
        out = torch.empty((n, h, w_, pw.cstore), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().ptocr_pwconv_bf16(_ptr(x), _ptr(pw.w), _ptr(pw.b), _ptr(res), _ptr(scale), _ptr(out), n, h, w_, cin, pw.cpad,
                                            pw.cstore, pw.act, res_mode, res.shape[3] if res is not None else 0, out.shape[3], coff,
                                            _lib.cur_stream()), "ptocr_pwconv_bf16")
    _count(x, pw.w, res, scale)
    if TRAFFIC is not None:
        _count(out[..., :pw.cstore])
    return out


def conv3x3(x, c3, out=None, up=1, coff=0, cstore=None):
    n, h, w_, cin = x.shape
    assert cin == c3.cin
    cs = cstore if cstore is not None else _r16(c3.cout)
    if out is None:
        out = torch.empty((n, h * up, w_ * up, cs), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().ptocr_conv3x3_bf16(_ptr(x), _ptr(c3.w), _ptr(c3.b), _ptr(out), n, h, w_, cin, cs, c3.act, up, out.shape[3], coff,
                                             _lib.cur_stream()), "ptocr_conv3x3_bf16")
    _count(x, c3.w)
    if TRAFFIC is not None:
        _count(out[..., :cs])
    return out


FUSE_PLANES = os.environ.get("PTOCR_BF16_FUSE_PLANES", "1") != "0"   # 0: the FPN output as one [N,H,W,96] concat buffer (rounds 2-3)


def conv3x3_planes(planes, c3):
    """the head's 3x3 conv on the FPN output kept as four planes bf16[4][N,H,W,24] (channel order of the concat)"""
    k, n, h, w_, c = planes.shape
    assert k == 4 and c == 24 and c3.cin == 96 and planes.is_contiguous()
    cs = _r16(c3.cout)
    out = torch.empty((n, h, w_, cs), dtype=torch.bfloat16, device=planes.device)
    _lib.check(_lib.lib().ptocr_conv3x3_planes_bf16(_ptr(planes), _ptr(c3.w), _ptr(c3.b), _ptr(out), n, h, w_, cs, c3.act, cs, _lib.cur_stream()),
               "ptocr_conv3x3_planes_bf16")
    _count(planes, c3.w, out)
    return out


LAT_FUSE = os.environ.get("PTOCR_BF16_LAT_FUSE", "1") != "0"      # 0: lateral in2 and smoothing conv out2 as two launches (rounds 2-3)


def conv3x3_lat(x2, lat, td, c3, out, up, coff, cstore):
    """smoothing conv of relu(bn(conv1x1(x2))) + nearest_x2(td) in one launch: the 96-channel lateral output is never written"""
    n, h, w_, cin = x2.shape
    assert cin == 16 and lat.cin == 16 and lat.cpad == 96 and c3.cin == 96 and td.shape[1] * 2 == h and td.shape[2] * 2 == w_
    _lib.check(_lib.lib().ptocr_conv3x3_lat_bf16(_ptr(x2), _ptr(lat.w), _ptr(lat.b), _ptr(td), td.shape[3], _ptr(c3.w), _ptr(c3.b), _ptr(out),
                                                 n, h, w_, cstore, c3.act, up, out.shape[3], coff, _lib.cur_stream()), "ptocr_conv3x3_lat_bf16")
    _count(x2, lat.w, td, c3.w)
    if TRAFFIC is not None:
        _count(out[..., coff:coff + cstore])
    return out


def dwconv(x, dw, want_pool):
    n, h, w_, c = x.shape
    assert c == dw.c
    pad = (dw.k - 1) // 2
    ho, wo = (h + 2 * pad - dw.k) // dw.stride + 1, (w_ + 2 * pad - dw.k) // dw.stride + 1
    y = torch.empty((n, ho, wo, c), dtype=torch.bfloat16, device=x.device)
    partial, nblk = None, 0
    if want_pool:
        nblk = int(_lib.lib().ptocr_dwconv_bf16_nblk(n, h, w_, dw.k, dw.stride))
        partial = torch.empty((n, nblk, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_dwconv_bf16(_ptr(x), _ptr(dw.w), _ptr(dw.b), _ptr(y), _ptr(partial), n, h, w_, c, dw.k, dw.stride, dw.act,
                                            _lib.cur_stream()), "ptocr_dwconv_bf16")
    _count(x, y, dw.w, partial)
    return y, partial, nblk


EXDW_FUSE = os.environ.get("PTOCR_BF16_EXDW_FUSE", "1") != "0"    # 0: expansion and depthwise conv of a 16-channel stride-2 block as two launches


def _exdw_ok(blk):
    ex, dw = blk.get("ex"), blk["dw"]
    return (EXDW_FUSE and ex is not None and "se" not in blk and ex.cin == 16 and dw.k == 3 and dw.stride == 2 and dw.c <= 96
            and ex.cstore == dw.c and ex.act == dw.act and ex.act in (ops.ACT_RELU, ops.ACT_HSWISH))


def expand_dw(x, ex, dw):
    """1x1 expansion + depthwise 3x3 / stride 2 in one launch: the expanded tensor is never written"""
    n, h, w_, cin = x.shape
    assert cin == 16 and ex.cin == 16
    y = torch.empty((n, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1, dw.c), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().ptocr_expand_dw3x3s2_bf16(_ptr(x), _ptr(ex.w), _ptr(ex.b), _ptr(dw.w), _ptr(dw.b), _ptr(y), n, h, w_, dw.c, ex.act, dw.act,
                                                    _lib.cur_stream()), "ptocr_expand_dw3x3s2_bf16")
    _count(x, ex.w, dw.w, y)
    return y


def se_gate(partial, nblk, se, hw):
    n = partial.shape[0]
    scale = torch.empty((n, se.c), dtype=torch.float32, device=partial.device)
    if se.c >= SE_SPLIT_MIN_C and SE_SPLIT:                                 # a function of the layer only: an image's gate never depends on its batch
        hidden = torch.empty((n, se.s), dtype=torch.float32, device=partial.device)
        _lib.check(_lib.lib().ptocr_se_fc_split_f32(_ptr(partial), _ptr(se.w1t), _ptr(se.b1), _ptr(se.w2t), _ptr(se.b2), _ptr(hidden), _ptr(scale),
                                                    n, hw, se.c, se.s, nblk, _lib.cur_stream()), "ptocr_se_fc_split_f32")
        return scale
    _lib.check(_lib.lib().ptocr_se_fc_t_f32(_ptr(partial), _ptr(se.w1t), _ptr(se.b1), _ptr(se.w2t), _ptr(se.b2), _ptr(scale), n, hw, se.c, se.s, nblk,
                                            _lib.cur_stream()), "ptocr_se_fc_t_f32")
    return scale


class Mbv3DbBf16:
    """Packed bf16 form of a det BaseModel with a MobileNetV3 backbone, FPN(mode DB, no ASF) neck and DBHead."""

    def __init__(self, model, dev):
        from .backbones.det_mobilenet_v3 import InvertedResidual, MobileNetV3
        from .heads.det_db_head import DBHead
        from .necks.fpn import FPN
        bb, neck, head = model.backbone, getattr(model, "neck", None), model.head
        if not (isinstance(bb, MobileNetV3) and isinstance(neck, FPN) and isinstance(head, DBHead)) or neck.use_asf:
            raise NotImplementedError("the bf16 path is built for DBNet with a MobileNetV3 backbone + FPN (mode DB, no ASF) + DBHead "
                                      "(BASELINE configs[3]); other architectures run in fp32")
        c1 = bb.conv1[0]
        if c1.weight.shape[0] != 16 or c1.weight.shape[1] != 3:
            raise NotImplementedError("bf16 stem kernel: 3 -> 16 channels (width_mult 1.0)")
        w, b = ops.fold_bn(c1.weight, c1.bias, bb.conv1[1])
        self.stem_w = _f(w.permute(1, 2, 3, 0).reshape(27, 16), dev)
        self.stem_b = _f(b, dev)
        self.out_channels = list(bb.out_channels)
        self.stages = []
        for stage in bb.stages:
            blocks = []
            for m in stage:
                if isinstance(m, InvertedResidual):
                    blk = {"dw": _Dw(m.conv2[0], m.conv2[1], dev, m.act_code), "pw": _Pw(m.conv3[0], m.conv3[1], dev, ops.ACT_NONE),
                           "res": m.use_res_connect}
                    if m.conv1 is not None:
                        blk["ex"] = _Pw(m.conv1[0], m.conv1[1], dev, m.act_code)
                    if m.se is not None:
                        blk["se"] = _Se(m.se, dev, blk["dw"].c)
                    blocks.append(blk)
                else:
                    blocks.append({"cba": _Pw(m[0], m[1], dev, ops.ACT_HSWISH)})
            self.stages.append(blocks)
        self.lat = {k: _Pw(getattr(neck, k)[0], getattr(neck, k)[1], dev, ops.ACT_RELU) for k in ("in5", "in4", "in3", "in2")}
        self.smooth = {k: _C3(getattr(neck, k)[0], getattr(neck, k)[1], dev, ops.ACT_RELU) for k in ("out5", "out4", "out3", "out2")}
        self.fuse_c, self.sm = neck.out_channels, neck.out_channels // 4
        hb = head.binarize
        self.head_c0 = _C3(hb[0], hb[1], dev, ops.ACT_RELU)
        c4 = hb[3].weight.shape[0]
        if c4 != 24:
            raise NotImplementedError("bf16 head tail kernel: 24 channels (FPN out_channels 96); got %d" % c4)
        w3, b3 = ops.fold_bn(hb[3].weight, hb[3].bias, hb[4], cout_dim=1)            # [Cin, Cout, 2, 2]
        self.t_w1 = _f(w3.permute(2, 3, 0, 1).reshape(4, c4, c4), dev)               # [(a*2+b)][ci][co]
        self.t_b1 = _f(b3, dev)
        w6 = hb[6].weight.detach().double().cpu()                                    # [C4, 1, 2, 2]
        self.t_w2 = _f(w6[:, 0].permute(1, 2, 0).reshape(4, c4), dev)                # [(a*2+b)][co]
        self.t_b2 = float(hb[6].bias.detach().cpu()[0])
        self.c4 = c4

    def forward(self, x, want_feats=False):
        """x f32[N,3,H,W] (H, W multiples of 32) on the device -> {"maps": f32[N,1,H,W]}; with want_feats also the reference's
        return_all_feats entries (base_model.py:56-73) as fp32 NCHW copies of the bf16 tensors"""
        x = x.contiguous().float()
        n, _, h, w_ = x.shape
        t = torch.empty((n, (h - 1) // 2 + 1, (w_ - 1) // 2 + 1, 16), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().ptocr_stem3x3s2_bf16(_ptr(x), _ptr(self.stem_w), _ptr(self.stem_b), _ptr(t), n, h, w_, ops.ACT_HSWISH,
                                                   _lib.cur_stream()), "ptocr_stem3x3s2_bf16")
        _count(x, t)
        feats = []
        for blocks in self.stages:
            for blk in blocks:
                if "cba" in blk:
                    t = pwconv(t, blk["cba"])
                    continue
                if _exdw_ok(blk):
                    d, partial, nblk = expand_dw(t, blk["ex"], blk["dw"]), None, 0
                else:
                    e = pwconv(t, blk["ex"]) if "ex" in blk else t
                    d, partial, nblk = dwconv(e, blk["dw"], "se" in blk)
                scale = se_gate(partial, nblk, blk["se"], d.shape[1] * d.shape[2]) if "se" in blk else None
                t = pwconv(d, blk["pw"], res=t if blk["res"] else None, res_mode=1 if blk["res"] else 0, scale=scale)
            feats.append(t)
        c2, c3, c4, c5 = feats
        in5 = pwconv(c5, self.lat["in5"])
        out4 = pwconv(c4, self.lat["in4"], res=in5, res_mode=2)
        out3 = pwconv(c3, self.lat["in3"], res=out4, res_mode=2)
        fuse_lat = (LAT_FUSE and not want_feats and c2.shape[3] == 16 and self.lat["in2"].cpad == 96 and self.fuse_c == 96
                    and self.lat["in2"].act == ops.ACT_RELU and c2.shape[1] % 2 == 0 and c2.shape[2] % 2 == 0)
        out2 = None if fuse_lat else pwconv(c2, self.lat["in2"], res=out3, res_mode=2)
        h4, w4 = c2.shape[1], c2.shape[2]
        sm = self.sm
        # The concat of fpn.py:96-100 as four planes [4][N,H4,W4,24] when the head conv can gather them (its persistent kernel: 96 input
        # channels): a smoothing conv then writes whole cache lines instead of 48-byte slices of 192-byte pixels.
        planes = FUSE_PLANES and self.fuse_c == 96 and sm == 24 and self.head_c0.cin == 96
        if planes:
            fuse = torch.empty((4, n, h4, w4, sm), dtype=torch.bfloat16, device=x.device)
            dst = [(fuse[k], 0) for k in range(4)]
        else:
            fuse = torch.empty((n, h4, w4, self.fuse_c), dtype=torch.bfloat16, device=x.device)
            dst = [(fuse, k * sm) for k in range(4)]
        conv3x3(in5, self.smooth["out5"], out=dst[0][0], up=8, coff=dst[0][1], cstore=sm)
        conv3x3(out4, self.smooth["out4"], out=dst[1][0], up=4, coff=dst[1][1], cstore=sm)
        conv3x3(out3, self.smooth["out3"], out=dst[2][0], up=2, coff=dst[2][1], cstore=sm)
        if fuse_lat:            # the largest lateral inside its smoothing conv: 0.7 GB less traffic per forward of 32 images
            conv3x3_lat(c2, self.lat["in2"], out3, self.smooth["out2"], out=dst[3][0], up=1, coff=dst[3][1], cstore=sm)
        else:
            conv3x3(out2, self.smooth["out2"], out=dst[3][0], up=1, coff=dst[3][1], cstore=sm)
        hx = conv3x3_planes(fuse, self.head_c0) if planes else conv3x3(fuse, self.head_c0)      # [N, H4, W4, 32], channels 24..31 zero
        maps = torch.empty((n, 1, 4 * h4, 4 * w4), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ptocr_db_head_tail_bf16(_ptr(hx), _ptr(self.t_w1), _ptr(self.t_b1), _ptr(self.t_w2), C.c_float(self.t_b2), _ptr(maps),
                                                      n, h4, w4, self.c4, hx.shape[3], _lib.cur_stream()), "ptocr_db_head_tail_bf16")
        _count(hx, maps)
        if want_feats:
            return {"backbone_out": [f.float().permute(0, 3, 1, 2)[:, :c].contiguous() for f, c in zip(feats, self.out_channels)],
                    "neck_out": (torch.cat(list(fuse), dim=3) if planes else fuse).float().permute(0, 3, 1, 2)[:, :self.fuse_c].contiguous(),
                    "maps": maps}
        return {"maps": maps}
