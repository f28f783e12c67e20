"""Host-side wrappers of the C ABI (include/ptocr_hip.h) used by the model mirror classes.

Activations are fp32 NHWC torch tensors on the MI355X; torch only provides device memory and streams."""
import ctypes as C

import torch

from .. import _lib
from .._lib import ConvDesc, RES_NONE, RES_ADD_PRE_RELU, RES_ADD_UP2_POST_RELU  # noqa: F401


# bench.py sets this to a list to time every conv launch with HIP events on the launch stream
PROFILE = None
PROFILE_LABELS = None                          # optional list: one text label per PROFILE entry (tools/profile_det_layers.py)
# 3x3 / stride 1 layers run as Winograd F(2x2,3x3) unless PTOCR_WINOGRAD=0 (then the direct implicit GEMM runs them)
import os as _os
USE_WINOGRAD = _os.environ.get("PTOCR_WINOGRAD", "1") != "0"
WINO_SPLIT = _os.environ.get("PTOCR_WINO_SPLIT", "0") == "1"       # experiment: F(4x4) Winograd with two-piece bf16 operands (NOT the fp32 path)
WINO4_MODE = _os.environ.get("PTOCR_WINO4", "auto")        # "0": F(2x2) only, "1": F(4x4) wherever it applies, else by cost
WINO4R = _os.environ.get("PTOCR_WINO4R", "1") != "0"        # F(4x4) layers on the round-5 kernel (conv_wino4r.hip); 0: conv_wino4.hip
WINO_COST = [2560, 14000, 2990, 23500]                     # cycles: F(2x2) per chunk / fixed, F(4x4) per chunk / fixed
# the 7x7 / stride 2 RGB stem runs in its own kernel unless PTOCR_STEM_KERNEL=0 (then the generic implicit GEMM runs it)
USE_PYRAMID = _os.environ.get("PTOCR_FPN_PYRAMID", "1") != "0"    # DB FPN output as a four-plane pyramid read in place by the head conv (no upsampled copies); 0: the concat tensor
USE_STEM_KERNEL = _os.environ.get("PTOCR_STEM_KERNEL", "1") != "0"
# ... and takes its 3x3 / stride 2 max pool along (one kernel) unless PTOCR_STEM_POOL=0
USE_STEM_POOL = _os.environ.get("PTOCR_STEM_POOL", "1") != "0"
# 1x1 convolutions with 64 input channels run in the LDS-resident-weights kernel unless PTOCR_PW64_KERNEL=0
USE_PW64_KERNEL = _os.environ.get("PTOCR_PW64_KERNEL", "1") != "0"
# CRNN's conv0 + ReLU + 2x2 pool run fused on the VALU unless PTOCR_SMALL_CONV_KERNEL=0
PW_MIN_PIXELS = 4096       # per image (NOT per batch: kernel choice must not depend on the batch size, results are compared
                           # bit-exactly across batch sizes); smaller maps go to the generic kernel
USE_SMALL_CONV_KERNEL = _os.environ.get("PTOCR_SMALL_CONV_KERNEL", "1") != "0"


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s -- the HIP path needs a cuda (ROCm) device, there is no CPU fallback"
                           % (what, t.device))


def fold_bn(w, conv_bias, bn, cout_dim=0):
    """Fold eval-mode BatchNorm into conv weight/bias (float64 on the host).  bn: nn.BatchNorm2d or None."""
    w = w.detach().double().cpu()
    cout = w.shape[cout_dim]
    b = conv_bias.detach().double().cpu() if conv_bias is not None else torch.zeros(cout, dtype=torch.float64)
    if bn is not None:
        scale = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        shape = [1] * w.dim()
        shape[cout_dim] = cout
        w = w * scale.view(shape)
        b = (b - bn.running_mean.detach().double().cpu()) * scale + bn.bias.detach().double().cpu()
    return w, b


def _rup(v, m):
    return (v + m - 1) // m * m


ACT_NONE, ACT_RELU, ACT_HSWISH = 0, 1, 2


def _act_code(relu):
    return int(relu) if not isinstance(relu, bool) else (ACT_RELU if relu else ACT_NONE)


class PackedConv:
    """Weights of one convolution in the kernel's layout: f32[Cout_k][Kpad], K = (kh*KW+kw)*Cin_pad + ci.

    Channel counts that are not multiples of 32 (MobileNetV3: 16, 24, 40, 72, ...) are handled by zero padding: the
    input tensor carries `cin` = roundup(Cin, 32) channels (4 for 1/3-channel images), the output tensor carries
    `c_tensor` = roundup(Cout, 32) channels, the weight matrix has roundup(c_tensor, 64) rows.  Padded rows/columns and
    their bias are zero, and every activation used here maps 0 to 0, so padding channels stay exactly zero."""

    def __init__(self, conv, bn, device, relu, cin_pad=None):
        w, b = fold_bn(conv.weight, conv.bias, bn)
        cout, cin, kh, kw = w.shape
        cin_pad = cin_pad or (4 if cin in (1, 3) else _rup(cin, 32))
        c_tensor = _rup(cout, 32)
        cout_k = _rup(c_tensor, 64)
        wk = torch.zeros(cout_k, kh, kw, cin_pad, dtype=torch.float64)
        wk[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        k = kh * kw * cin_pad
        kpad = (k + 31) // 32 * 32
        wp = torch.zeros(cout_k, kpad, dtype=torch.float64)
        wp[:, :k] = wk.reshape(cout_k, k)
        bp = torch.zeros(cout_k, dtype=torch.float64)
        bp[:cout] = b
        self.w = wp.float().contiguous().to(device)
        self.b = bp.float().contiguous().to(device)
        self.cin, self.cout, self.kh, self.kw = cin_pad, cout_k, kh, kw
        self.cout_real, self.c_tensor = cout, c_tensor
        self.stride = conv.stride[0]
        self.pad_h, self.pad_w = conv.padding
        self.relu = _act_code(relu)
        self.convt = False
        # 3x3 / s1 / p1 layers with <= 4 input channels and 64 outputs followed by ReLU (CRNN conv0): w[(ci*3+ky)*3+kx][cout]
        self.small_w = None
        if (kh, kw) == (3, 3) and self.stride == 1 and (self.pad_h, self.pad_w) == (1, 1) and cin <= 4 and cout == 64 \
                and self.relu == ACT_RELU:
            self.small_w = w.permute(1, 2, 3, 0).reshape(cin * 9, 64).contiguous().float().to(device)
            self.small_b = b.float().contiguous().to(device)
            self.small_cin = cin
        # pointwise layers with <= 64 input channels (the FPN lateral in2, MobileNetV3's wide early layers): W[k][cout], k-major,
        # channels zero-padded to the tensors' widths, for the LDS-resident-weights kernel
        self.pw_w = None
        if (kh, kw) == (1, 1) and self.stride == 1 and (self.pad_h, self.pad_w) == (0, 0) and (cin_pad in (32, 64) or (cin_pad == 128 and c_tensor % 128 == 0)) and c_tensor <= 256:
            pw = torch.zeros(cin_pad, c_tensor, dtype=torch.float64)
            pw[:cin, :cout] = w.reshape(cout, cin).t()
            pb = torch.zeros(c_tensor, dtype=torch.float64)
            pb[:cout] = b
            self.pw_w = pw.float().contiguous().to(device)
            self.pw_b = pb.float().contiguous().to(device)
        # ResNet stem (7x7 / s2 / p3, RGB -> 64): K axis without the padding channel, w[ky][kx*3 + c][cout], one zero row per ky
        self.stem_w = None
        if (kh, kw) == (7, 7) and self.stride == 2 and (self.pad_h, self.pad_w) == (3, 3) and cin == 3 and cout == 64 \
                and cin_pad == 4 and self.relu in (ACT_NONE, ACT_RELU):
            sw = torch.zeros(7, 22, 64, dtype=torch.float64)
            sw[:, :21, :] = w.permute(2, 3, 1, 0).reshape(7, 21, 64)              # [ky][kx][c][cout] -> [ky][kx*3 + c][cout]
            self.stem_w = sw.float().contiguous().to(device)
            self.stem_b = b.float().contiguous().to(device)
        # Winograd F(2x2,3x3) form of the same weights for 3x3 / s1 / p1 layers: U = G g G^T, packed [Cout/64][Cin/4][16][64][4]
        self.wino_u = self.wino4_u = None
        self.wino4_ok = False                                   # an F(4x4,3x3) form of the weights exists (wino4r_u, or wino4_u for the first cut)
        if (kh, kw) == (3, 3) and self.stride == 1 and (self.pad_h, self.pad_w) == (1, 1) and cin % 16 == 0 and cout % 4 == 0 \
                and cin_pad == cin and self.relu in (ACT_NONE, ACT_RELU):
            G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
            cw = _rup(cout, 64)                                                    # narrower layers: zero weights up to 64
            U = torch.zeros(cw, cin, 4, 4, dtype=torch.float64)
            U[:cout] = torch.einsum("ar,ocrs,bs->ocab", G, w, G)                   # [Cout, Cin, 4, 4]
            U = U.reshape(cw // 64, 64, cin // 4, 4, 16).permute(0, 2, 4, 1, 3)    # [ct, chunk, xi, cout, c4]
            self.wino_u = U.contiguous().float().to(device)
            bw = torch.zeros(cw, dtype=torch.float64)
            bw[:cout] = b
            self.wino_b = bw.float().contiguous().to(device)
            self.wino_cout = cw
            # F(4x4,3x3) form for the large maps (conv_wino4.hip): U = G6 g G6^T, packed [Cout/64][Cin/4][12 waves][3 xi][64 lanes][4]:
            # wave w owns xi = 3w + e; lane (n = lane & 31, h = lane >> 5) holds {nb, t} -> U[xi][4 chunk + 2h + t][64 ct + 32 nb + n]
            G6 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                               [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
            # (round 6: packed only where the first cut runs -- PTOCR_WINO4R=0, or the split experiment, which shares its layout; each F(4x4)
            # form is 4x the conv weights and its own fp64 einsum)
            self.wino4_ok = True
            U6 = None
            if not WINO4R or WINO_SPLIT:
                U6 = torch.zeros(cw, cin, 36, dtype=torch.float64)
                U6[:cout] = torch.einsum("ar,ocrs,bs->ocab", G6, w, G6).reshape(cout, cin, 36)
                U6 = U6.reshape(cw // 64, 2, 32, cin // 4, 2, 2, 12, 3)               # [ct, nb, n, chunk, h, t, w, e]
                U6 = U6.permute(0, 3, 6, 7, 4, 2, 1, 5)                                # [ct, chunk, w, e, h, n, nb, t]
                self.wino4_u = U6.contiguous().float().to(device)
            # round-5 re-cut (conv_wino4r.hip): wave (wh, wi) owns the frequency row xi = 6 wi + j for the output channels 32 wh + n;
            # packed [ct][chunk][wh][wi][q][kh][n][jj][t] = U[xi = 6 wi + 2 q + jj][cin = 4 chunk + 2 kh + t][cout = 64 ct + 32 wh + n]
            self.wino4r_u = None
            if WINO4R:
                U6r = torch.zeros(cw, cin, 36, dtype=torch.float64)
                U6r[:cout] = torch.einsum("ar,ocrs,bs->ocab", G6, w, G6).reshape(cout, cin, 36)
                U6r = U6r.reshape(cw // 64, 2, 32, cin // 4, 2, 2, 6, 3, 2)          # [ct, wh, n, chunk, kh, t, wi, q, jj]
                U6r = U6r.permute(0, 3, 1, 6, 7, 4, 2, 8, 5)                         # [ct, chunk, wh, wi, q, kh, n, jj, t]
                self.wino4r_u = U6r.contiguous().float().to(device)
            # experiment (PTOCR_WINO_SPLIT=1): the same layout with two bf16 pieces in the place of each fp32 -- bf16(U) in the low half,
            # bf16(U - bf16(U)) in the high half (ptocr_conv3x3_wino4_split_f32)
            self.wino4_us = None
            if WINO_SPLIT:
                u32 = U6.contiguous().float()
                hi = u32.to(torch.bfloat16)
                mid = (u32 - hi.float()).to(torch.bfloat16)
                word = hi.view(torch.int16).to(torch.int32).bitwise_and(0xffff) | (mid.view(torch.int16).to(torch.int32) << 16)
                self.wino4_us = word.view(torch.float32).contiguous().to(device)


class PackedConvT2x2:
    """ConvTranspose2d(k=2, s=2) as a 1x1 GEMM with 4*Cp columns: column (a*2+b)*Cp + co, Cp = roundup(Co, 32)."""

    def __init__(self, convt, bn, device, relu):
        w, b = fold_bn(convt.weight, convt.bias, bn, cout_dim=1)       # [Cin, Co, 2, 2]
        cin, co = w.shape[0], w.shape[1]
        cin_pad, cp = _rup(cin, 32), _rup(co, 32)
        cout_k = _rup(4 * cp, 64)
        wp = torch.zeros(cout_k, cin_pad, dtype=torch.float64)
        bp = torch.zeros(cout_k, dtype=torch.float64)
        for a in range(2):
            for bb in range(2):
                g = (a * 2 + bb) * cp
                wp[g:g + co, :cin] = w[:, :, a, bb].t()
                bp[g:g + co] = b
        self.w = wp.float().contiguous().to(device)
        self.b = bp.float().contiguous().to(device)
        self.cin, self.cout, self.kh, self.kw = cin_pad, cout_k, 1, 1
        self.stride, self.pad_h, self.pad_w = 1, 0, 0
        self.relu = _act_code(relu)
        self.convt = True
        self.co = cp
        self.cout_real, self.c_tensor = co, cp


def _wino4_wins(H, W, Cin):
    """F(4x4,3x3) or F(2x2,3x3) for this layer: patches per image of the best geometry of each kernel x cycles per patch (main
    loop per 4-channel chunk + fixed part, measured with the s_memtime probes).  Depends on the map size and the channels only,
    never on the batch (results are compared bit-exactly across batch sizes); PTOCR_WINO4=0/1 forces one."""
    if WINO4_MODE in ("0", "1"):
        return WINO4_MODE == "1"
    cd = lambda a, b: (a + b - 1) // b
    n = 16                                                 # nominal batch: every multi-image geometry divides it
    p2 = min(100 * n * cd(H, 16) * cd(W, 16), 100 * n * cd(H, 32) * cd(W, 8), 106 * cd(n, 2) * cd(H, 8) * cd(W, 16),
             112 * cd(n, 4) * cd(H, 4) * cd(W, 16), 112 * cd(n, 4) * cd(H, 8) * cd(W, 8)) / 100.0
    p4 = _lib.lib().ptocr_conv3x3_wino4_patches(n, H, W)
    return p4 * (WINO_COST[2] * (Cin // 4) + WINO_COST[3]) < p2 * (WINO_COST[0] * (Cin // 4) + WINO_COST[1])


def conv2d(x, pc, res=None, res_mode=RES_NONE, out=None, out_up=1, out_coff=0, store=None):
    """x f32[N,H,W,Cin] -> f32[N,Ho*,Wo*,C] (new tensor unless `out` is given for in-place concat; `store` = number of
    columns written, default the padded tensor width, pass the real channel count for exact concat slices)."""
    _require_cuda(x, "conv2d")
    N, H, W, Cin = x.shape
    assert Cin == pc.cin, (Cin, pc.cin)
    Ho = (H + 2 * pc.pad_h - pc.kh) // pc.stride + 1
    Wo = (W + 2 * pc.pad_w - pc.kw) // pc.stride + 1
    scale = 2 if pc.convt else out_up
    if out is None:
        out = torch.empty((N, Ho * scale, Wo * scale, pc.c_tensor), dtype=torch.float32, device=x.device)
    # The fast kernels address their tensors with 32-bit byte offsets.  A batch whose input, output (the 256-channel concat buffer of a
    # 64-image batch is 3 GB) or residual reaches 2 GiB is run as two half batches into the same output -- images are independent and
    # every kernel is batch-invariant, so the result is the same; without this the layer fell back to the first-generation kernel
    # (7.7 ms per layer in the 64-image run_ocr workload).
    lim = 2 ** 31
    if N > 1 and (x.numel() * 4 >= lim or out.numel() * 4 >= lim or (res is not None and res.numel() * 4 >= lim)):
        h = N // 2
        for a, b in ((0, h), (h, N)):
            conv2d(x[a:b], pc, res=None if res is None else res[a:b], res_mode=res_mode, out=out[a:b], out_up=out_up, out_coff=out_coff,
                   store=store)
        return out
    if USE_STEM_KERNEL and getattr(pc, "stem_w", None) is not None and res is None and out_up == 1 and out_coff == 0 \
            and out.shape[3] == 64 and (store is None or store == 64) and N * H * W * 16 < 2 ** 31:
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(_lib.lib().ptocr_conv7x7s2_stem_f32(_lib.ptr(x), _lib.ptr(pc.stem_w), _lib.ptr(pc.stem_b), _lib.ptr(out),
                                                       N, H, W, int(pc.relu), _lib.cur_stream()), "ptocr_conv7x7s2_stem_f32")
        if PROFILE is not None:
            e1.record()
            PROFILE.append((e0, e1))
            if PROFILE_LABELS is not None:
                PROFILE_LABELS.append("stem7x7 %dx%dx%dx3->64" % (N, H, W))
        return out
    if USE_PW64_KERNEL and getattr(pc, "pw_w", None) is not None and out_up == 1 and res_mode in (RES_NONE, RES_ADD_UP2_POST_RELU) \
            and (store is None or store == pc.c_tensor) and PW_MIN_PIXELS <= H * W and N * H * W * Cin * 4 < 2 ** 31 \
            and (res_mode == RES_NONE or (H % 2 == 0 and W % 2 == 0 and res.shape[3] == pc.c_tensor)):
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(_lib.lib().ptocr_conv1x1_small_k_f32(_lib.ptr(x), _lib.ptr(pc.pw_w), _lib.ptr(pc.pw_b),
                                                        _lib.ptr(res) if res is not None else C.c_void_p(0), _lib.ptr(out),
                                                        N, H, W, Cin, pc.c_tensor, int(pc.relu), int(res_mode == RES_ADD_UP2_POST_RELU),
                                                        out.shape[3], out_coff, _lib.cur_stream()), "ptocr_conv1x1_small_k_f32")
        if PROFILE is not None:
            e1.record()
            PROFILE.append((e0, e1))
            if PROFILE_LABELS is not None:
                PROFILE_LABELS.append("pw1x1 %dx%dx%dx%d->%d" % (N, H, W, Cin, pc.c_tensor))
        return out
    if USE_WINOGRAD and getattr(pc, "wino_u", None) is not None and (res_mode == RES_NONE or (res_mode == RES_ADD_PRE_RELU and out_up == 1)) \
            and out_up <= 8 and (store if store is not None else pc.c_tensor) % 4 == 0 and N * H * W * Cin * 4 < 2 ** 31 \
            and (store if store is not None else pc.c_tensor) <= pc.wino_cout:
        cs = store if store is not None else pc.c_tensor      # columns written: zero weights / bias beyond the real channels
        four = getattr(pc, "wino4_ok", getattr(pc, "wino4_u", None) is not None) and _wino4_wins(H, W, Cin) and out.numel() * 4 < 2 ** 31 and (res is None or res.numel() * 4 < 2 ** 31)
        if PROFILE is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        split = four and getattr(pc, "wino4_us", None) is not None
        recut = four and not split and getattr(pc, "wino4r_u", None) is not None
        fn = (_lib.lib().ptocr_conv3x3_wino4_split_f32 if split else _lib.lib().ptocr_conv3x3_wino4r_f32 if recut else _lib.lib().ptocr_conv3x3_wino4_f32) if four else _lib.lib().ptocr_conv3x3_wino_f32
        _lib.check(fn(_lib.ptr(x), _lib.ptr((pc.wino4_us if split else pc.wino4r_u if recut else pc.wino4_u) if four else pc.wino_u), _lib.ptr(pc.wino_b),
                      _lib.ptr(res) if res is not None else C.c_void_p(0), _lib.ptr(out),
                      N, H, W, Cin, pc.wino_cout, cs, int(pc.relu), res_mode,
                      res.shape[3] if res is not None else 0, out.shape[3], out_coff,
                      out_up, _lib.cur_stream()), "ptocr_conv3x3_wino4_f32" if four else "ptocr_conv3x3_wino_f32")
        if PROFILE is not None:
            e1.record()
            PROFILE.append((e0, e1))
            if PROFILE_LABELS is not None:
                PROFILE_LABELS.append("wino%s3x3 %dx%dx%dx%d->%d%s" % ("4" if four else "", N, H, W, Cin, pc.cout_real,
                                                                      " up%d" % out_up if out_up > 1 else ""))
        return out
    if pc.convt:
        cout_k, cstore = 4 * pc.co, 0
        assert 4 * pc.co == pc.cout, "transposed-conv column groups must fill the GEMM width"
    else:
        cout_k = pc.cout
        cstore = store if store is not None else pc.c_tensor
        cstore = 0 if cstore == cout_k else cstore
    d = ConvDesc(N, H, W, Cin, cout_k, pc.kh, pc.kw, pc.stride, pc.pad_h, pc.pad_w, Ho, Wo,
                 int(pc.relu), res_mode, out_up, out.shape[3], out_coff, int(pc.convt), cstore,
                 res.shape[3] if res is not None else 0)
    L = _lib.lib()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(L.ptocr_conv2d_f32(C.byref(d), _lib.ptr(x), _lib.ptr(pc.w), _lib.ptr(pc.b),
                                  _lib.ptr(res) if res is not None else C.c_void_p(0), _lib.ptr(out),
                                  _lib.cur_stream()), "ptocr_conv2d_f32")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1))
        if PROFILE_LABELS is not None:
            PROFILE_LABELS.append("conv%dx%d s%d %dx%dx%dx%d->%d%s" % (pc.kh, pc.kw, pc.stride, N, H, W, Cin, cout_k,
                                                                      " up%d" % out_up if out_up > 1 else ""))
    return out


class Pyramid:
    """Four 64-channel planes of a virtual f32[N,H,W,256] tensor in ONE allocation, plane j at 1 / 2^shifts[j] of the resolution: what
    the reference's FPN builds with interpolate + cat (pytocr/modeling/necks/fpn.py:118-131), never upsampled -- the consumer
    (conv3x3_pyramid -> ptocr_conv3x3_wino4r_pyramid_f32) reads pixel (y >> shift, x >> shift) of a plane for pixel (y, x)."""

    def __init__(self, N, H, W, shifts, device):
        self.N, self.H, self.W, self.shifts = N, H, W, tuple(int(v) for v in shifts)
        sizes = [N * (H >> v) * (W >> v) * 64 for v in self.shifts]
        self.offs = [sum(sizes[:j]) for j in range(4)]
        self.sizes = sizes
        self.buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
        self.shape = (N, H, W, 256)

    @staticmethod
    def fits(N, H, W, shifts=(3, 2, 1, 0)):
        return all(H % (1 << v) == 0 and W % (1 << v) == 0 for v in shifts) and \
            sum(N * (H >> v) * (W >> v) * 64 for v in shifts) * 4 < 2 ** 31

    def plane(self, j):
        v = self.shifts[j]
        return self.buf[self.offs[j]:self.offs[j] + self.sizes[j]].view(self.N, self.H >> v, self.W >> v, 64)

    def materialize(self):
        """the concat tensor f32[N,H,W,256] (tests, `return_all_feats`, consumers without the pyramid kernel)"""
        parts = []
        for j, v in enumerate(self.shifts):
            t = self.plane(j)
            if v:
                t = t.repeat_interleave(1 << v, dim=1).repeat_interleave(1 << v, dim=2)
            parts.append(t)
        return torch.cat(parts, dim=3).contiguous()


def pyramid_conv_ok(pc, N, H, W):
    """can conv3x3_pyramid run this 3x3 conv on a Pyramid of an [N,H,W,256] input?  (a function of the map, never of the batch's content)"""
    return bool(USE_PYRAMID and USE_WINOGRAD and WINO4R and getattr(pc, "wino4r_u", None) is not None and getattr(pc, "wino4_us", None) is None
                and pc.cin == 256 and pc.kh == 3 and pc.stride == 1 and pc.c_tensor % 4 == 0 and pc.c_tensor <= pc.wino_cout
                and _wino4_wins(H, W, 256) and Pyramid.fits(N, H, W) and N * H * W * pc.c_tensor * 4 < 2 ** 31)


def conv3x3_pyramid(pyr, pc):
    """3x3 / stride 1 conv (+ folded BN, ReLU) of the virtual concat a Pyramid stands for -> f32[N,H,W,C]; bit-identical to
    conv2d(pyr.materialize(), pc)"""
    _require_cuda(pyr.buf, "conv3x3_pyramid")
    N, H, W = pyr.N, pyr.H, pyr.W
    assert pyramid_conv_ok(pc, N, H, W), "conv3x3_pyramid: this conv / map cannot take a pyramid (ask pyramid_conv_ok first)"
    out = torch.empty((N, H, W, pc.c_tensor), dtype=torch.float32, device=pyr.buf.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    off, sh = (C.c_longlong * 4)(*pyr.offs), (C.c_int * 4)(*pyr.shifts)
    _lib.check(_lib.lib().ptocr_conv3x3_wino4r_pyramid_f32(_lib.ptr(pyr.buf), off, sh, C.c_longlong(pyr.buf.numel()), _lib.ptr(pc.wino4r_u),
                                                           _lib.ptr(pc.wino_b), _lib.ptr(out), N, H, W, pc.wino_cout, pc.c_tensor, int(pc.relu),
                                                           out.shape[3], 0, _lib.cur_stream()), "ptocr_conv3x3_wino4r_pyramid_f32")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1))
        if PROFILE_LABELS is not None:
            PROFILE_LABELS.append("wino43x3 %dx%dx%dx256->%d pyramid" % (N, H, W, pc.cout_real))
    return out


class PackedDW:
    """Depthwise conv + BN (+act): weights f32[k*k][Cp] tap-major, channels zero-padded to a multiple of 32."""

    def __init__(self, conv, bn, device, act):
        w, b = fold_bn(conv.weight, conv.bias, bn)                      # [C, 1, k, k]
        c, _, k, _ = w.shape
        cp = _rup(c, 32)
        wp = torch.zeros(k * k, cp, dtype=torch.float64)
        wp[:, :c] = w[:, 0].reshape(c, k * k).t()
        bp = torch.zeros(cp, dtype=torch.float64)
        bp[:c] = b
        self.w, self.b = wp.float().contiguous().to(device), bp.float().contiguous().to(device)
        self.k, self.stride, self.stride_w, self.act, self.c = k, conv.stride[0], conv.stride[1], _act_code(act), cp


def dwconv(x, pd):
    _require_cuda(x, "dwconv")
    N, H, W, Cc = x.shape
    assert Cc == pd.c
    pad = (pd.k - 1) // 2
    Ho, Wo = (H + 2 * pad - pd.k) // pd.stride + 1, (W + 2 * pad - pd.k) // pd.stride_w + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_dwconv2_f32(_lib.ptr(x), _lib.ptr(pd.w), _lib.ptr(pd.b), _lib.ptr(y), N, H, W, Cc, pd.k, pd.stride, pd.stride_w,
                                            pd.act, _lib.cur_stream()), "ptocr_dwconv2_f32")
    return y


def cls_head(x, w, b):
    """x f32[N,H,W,Cp] -> softmax f32[N,K]: AvgPool2d(2,2) + AdaptiveAvgPool2d(1) + Linear + softmax (one kernel)"""
    _require_cuda(x, "cls_head")
    N, H, W, Cc = x.shape
    K = w.shape[0]
    y = torch.empty((N, K), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_cls_head_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, H, W, Cc, K, _lib.cur_stream()),
               "ptocr_cls_head_f32")
    return y


class PackedSE:
    def __init__(self, se, device):
        w1 = se.fc1.weight.detach().float().cpu()[:, :, 0, 0]            # [S, C]
        w2 = se.fc2.weight.detach().float().cpu()[:, :, 0, 0]            # [C, S]
        S, Cc = w1.shape
        cp = _rup(Cc, 32)
        w1p = torch.zeros(S, cp); w1p[:, :Cc] = w1
        w2p = torch.zeros(cp, S); w2p[:Cc] = w2
        b2p = torch.full((cp,), -3.0); b2p[:Cc] = se.fc2.bias.detach().float().cpu()      # hardsigmoid(-3) = 0 keeps pad channels 0
        self.w1, self.b1 = w1p.contiguous().to(device), se.fc1.bias.detach().float().cpu().contiguous().to(device)
        self.w2, self.b2 = w2p.contiguous().to(device), b2p.contiguous().to(device)
        self.S, self.c = S, cp


def se_scale_(x, ps):
    """in place: x[n,:,:,c] *= hardsigmoid(fc2(relu(fc1(mean(x)))))"""
    _require_cuda(x, "se_scale_")
    N, H, W, Cc = x.shape
    assert Cc == ps.c
    work = torch.empty(N * ((H * W + 2047) // 2048 + 1) * Cc, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_se_scale_f32(_lib.ptr(x), _lib.ptr(ps.w1), _lib.ptr(ps.b1), _lib.ptr(ps.w2), _lib.ptr(ps.b2),
                                             _lib.ptr(work), N, H, W, Cc, ps.S, _lib.cur_stream()), "ptocr_se_scale_f32")
    return x


def nchw_to_nhwc(x, cpad):
    _require_cuda(x, "nchw_to_nhwc")
    x = x.contiguous().float()
    N, Cc, H, W = x.shape
    y = torch.empty((N, H, W, cpad), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_nchw_to_nhwc_f32(_lib.ptr(x), _lib.ptr(y), N, Cc, H, W, cpad, _lib.cur_stream()),
               "ptocr_nchw_to_nhwc_f32")
    return y


def nhwc_to_nchw(x):
    _require_cuda(x, "nhwc_to_nchw")
    N, H, W, Cc = x.shape
    y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_nhwc_to_nchw_f32(_lib.ptr(x), _lib.ptr(y), N, Cc, H, W, _lib.cur_stream()),
               "ptocr_nhwc_to_nchw_f32")
    return y


def stem_relu_pool(x4, x_nchw, pc):
    """ResNet stem conv 7x7/s2 + BN + ReLU + MaxPool2d(3, 2, 1) in one kernel (the full-resolution stem output never reaches HBM);
    input either f32[N,H,W,4] (x4) or the model's f32[N,3,H,W] (x_nchw).  None when the fused kernel does not apply."""
    x = x4 if x_nchw is None else x_nchw
    _require_cuda(x, "stem_relu_pool")
    if not (USE_STEM_KERNEL and USE_STEM_POOL and getattr(pc, "stem_w", None) is not None and pc.relu == ACT_RELU and x.dtype == torch.float32):
        return None
    if x_nchw is None:
        N, H, W, Cc = x.shape
        if Cc != 4 or N * H * W * 16 >= 2 ** 31:
            return None
    else:
        N, Cc, H, W = x.shape
        if Cc != 3 or N * H * W * 12 >= 2 ** 31:
            return None
    x = x.contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = _lib.lib().ptocr_conv7x7s2_stem_relu_pool_f32 if x_nchw is None else _lib.lib().ptocr_conv7x7s2_stem_relu_pool_nchw_f32
    _lib.check(fn(_lib.ptr(x), _lib.ptr(pc.stem_w), _lib.ptr(pc.stem_b), _lib.ptr(out), N, H, W, _lib.cur_stream()),
               "ptocr_conv7x7s2_stem_relu_pool_f32")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1))
        if PROFILE_LABELS is not None:
            PROFILE_LABELS.append("stem7x7+pool %dx%dx%dx3->64" % (N, H, W))
    return out


CONV_POOL_FUSE = _os.environ.get("PTOCR_CONV_POOL_FUSE", "1") != "0"      # 0: conv + ReLU and MaxPool2d(2, 2) as two launches


def conv2d_relu_pool2(x, pc):
    """3x3 / s1 / p1 conv + ReLU + MaxPool2d(2, 2) (CRNN conv1 + pooling1, rec_vgg.py:28-35): one launch when the layer runs on the F(4x4)
    Winograd kernel (its epilogue takes the maxima of the 2x2 windows inside each 4x4 output tile), conv2d + maxpool2d otherwise"""
    _require_cuda(x, "conv2d_relu_pool2")
    N, H, W, Cin = x.shape
    fused = (CONV_POOL_FUSE and USE_WINOGRAD and getattr(pc, "wino4_ok", False) and getattr(pc, "wino4_us", None) is None and pc.relu
             and pc.kh == 3 and pc.kw == 3 and pc.stride == 1 and pc.pad_h == 1 and pc.pad_w == 1 and not pc.convt
             and H % 2 == 0 and W % 2 == 0 and pc.c_tensor % 4 == 0 and pc.c_tensor <= pc.wino_cout
             and _wino4_wins(H, W, Cin) and N * H * W * max(Cin, pc.c_tensor) * 4 < 2 ** 31)
    if not fused:
        return maxpool2d(conv2d(x, pc), 2, 2, 0)
    out = torch.empty((N, H // 2, W // 2, pc.c_tensor), dtype=torch.float32, device=x.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    recut = getattr(pc, "wino4r_u", None) is not None
    _lib.check((_lib.lib().ptocr_conv3x3_wino4r_pool2_f32 if recut else _lib.lib().ptocr_conv3x3_wino4_pool2_f32)(
        _lib.ptr(x), _lib.ptr(pc.wino4r_u if recut else pc.wino4_u), _lib.ptr(pc.wino_b), _lib.ptr(out), N, H, W, Cin,
        pc.wino_cout, pc.c_tensor, out.shape[3], _lib.cur_stream()), "ptocr_conv3x3_wino4_pool2_f32")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1))
        if PROFILE_LABELS is not None:
            PROFILE_LABELS.append("wino43x3 %dx%dx%dx%d->%d pool2" % (N, H, W, Cin, pc.cout_real))
    return out


def maxpool2d(x, k, s, p):
    _require_cuda(x, "maxpool2d")
    kh, kw = (k, k) if isinstance(k, int) else k
    sh, sw = (s, s) if isinstance(s, int) else s
    ph, pw = (p, p) if isinstance(p, int) else p
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_maxpool2d_f32(_lib.ptr(x), _lib.ptr(y), N, H, W, Cc, kh, kw, sh, sw, ph, pw, Ho, Wo,
                                              _lib.cur_stream()), "ptocr_maxpool2d_f32")
    return y


def convt2x2_sigmoid(x, w4, bias):
    _require_cuda(x, "convt2x2_sigmoid")
    N, H, W, Cc = x.shape
    y = torch.empty((N, 1, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_convt2x2_sigmoid_f32(_lib.ptr(x), _lib.ptr(w4), C.c_float(bias), _lib.ptr(y), N, H, W, Cc,
                                                     _lib.cur_stream()), "ptocr_convt2x2_sigmoid_f32")
    return y


def linear(x, w, b):
    """x f32[M,K] @ w[Nout,K]^T + b -> f32[M,Nout] (MFMA GEMM; K % 32 == 0, Nout % 64 == 0)"""
    _require_cuda(x, "linear")
    M, K = x.shape
    Nout = w.shape[0]
    y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_linear_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, K, Nout, Nout,
                                           _lib.cur_stream()), "ptocr_linear_f32")
    return y


def lstm_bidir(xproj, w_hh, B, T):
    """xproj f32[B*T, 8H] (= [B][T][2][4H]), w_hh f32[2,4H,H] -> f32[B*T, 2H]"""
    _require_cuda(xproj, "lstm_bidir")
    H = w_hh.shape[2]
    out = torch.empty((B * T, 2 * H), dtype=torch.float32, device=xproj.device)
    _lib.check(_lib.lib().ptocr_lstm_bidir_f32(_lib.ptr(xproj), _lib.ptr(w_hh), _lib.ptr(out), T, B, H, _lib.cur_stream()),
               "ptocr_lstm_bidir_f32")
    return out


def stem_from_nchw(x, pc):
    """ResNet stem applied to the model's own input tensor x f32[N,3,H,W] -> f32[N,Ho,Wo,64] (the stem kernel reads the three
    planes itself); falls back to the boundary layout pass + conv2d when the stem kernel does not apply"""
    _require_cuda(x, "stem_from_nchw")
    N, Cc, H, W = x.shape
    if not USE_STEM_KERNEL or getattr(pc, "stem_w", None) is None or Cc != 3 or x.dtype != torch.float32 or N * H * W * 12 >= 2 ** 31:
        return conv2d(nchw_to_nhwc(x, 4), pc)
    x = x.contiguous()
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().ptocr_conv7x7s2_stem_nchw_f32(_lib.ptr(x), _lib.ptr(pc.stem_w), _lib.ptr(pc.stem_b), _lib.ptr(out),
                                                        N, H, W, int(pc.relu), _lib.cur_stream()), "ptocr_conv7x7s2_stem_nchw_f32")
    if PROFILE is not None:
        e1.record()
        PROFILE.append((e0, e1))
        if PROFILE_LABELS is not None:
            PROFILE_LABELS.append("stem7x7 %dx%dx%dx3->64 (nchw)" % (N, H, W))
    return out


def conv3x3_relu_pool2(x4, pc):
    """maxpool2x2(relu(conv3x3(x4) + b)) for a layer with <= 4 input channels and 64 outputs (CRNN conv0 + pooling0), fused:
    x4 f32[N,H,W,4] -> f32[N,H/2,W/2,64]; falls back to the generic conv + pool kernels when PTOCR_SMALL_CONV_KERNEL=0"""
    _require_cuda(x4, "conv3x3_relu_pool2")
    if not USE_SMALL_CONV_KERNEL or getattr(pc, "small_w", None) is None or x4.shape[3] != 4:
        return maxpool2d(conv2d(x4, pc), 2, 2, 0)
    N, H, W, _ = x4.shape
    out = torch.empty((N, H // 2, W // 2, 64), dtype=torch.float32, device=x4.device)
    _lib.check(_lib.lib().ptocr_conv3x3_small_relu_pool_f32(_lib.ptr(x4), _lib.ptr(pc.small_w), _lib.ptr(pc.small_b), _lib.ptr(out),
                                                            N, H, W, pc.small_cin, _lib.cur_stream()),
               "ptocr_conv3x3_small_relu_pool_f32")
    return out


def lstm_check():
    """raises if a split-LSTM call's inter-workgroup exchange timed out (call after the stream was synchronised)"""
    _lib.check(_lib.lib().ptocr_lstm_check(), "ptocr_lstm_check")


def ctc_greedy(x, C, is_prob):
    """x f32[rows, ld] -> (idx int32[rows], prob f32[rows])"""
    _require_cuda(x, "ctc_greedy")
    rows, ld = x.shape
    idx = torch.empty(rows, dtype=torch.int32, device=x.device)
    prob = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_ctc_greedy_f32(_lib.ptr(x), rows, C, ld, int(is_prob), _lib.ptr(idx), _lib.ptr(prob),
                                               _lib.cur_stream()), "ptocr_ctc_greedy_f32")
    return idx, prob


def linear_ctc_greedy(x, w, b, C):
    """CTC head FC fused with the greedy reductions: x f32[M,K], w f32[Np,K] (Np % 128 == 0, rows >= C zero), b f32[Np]
    -> (idx int32[M], prob f32[M]) over the first C columns; the logits are never written"""
    _require_cuda(x, "linear_ctc_greedy")
    M, K = x.shape
    Np = w.shape[0]
    idx = torch.empty(M, dtype=torch.int32, device=x.device)
    prob = torch.empty(M, dtype=torch.float32, device=x.device)
    work = torch.empty((M, Np // 64, 4), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_linear_ctc_greedy_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), M, K, Np, C, _lib.ptr(work),
                                                      _lib.ptr(idx), _lib.ptr(prob), _lib.cur_stream()), "ptocr_linear_ctc_greedy_f32")
    return idx, prob


FUSE_CTC = _os.environ.get("PTOCR_FUSE_CTC", "1") != "0"     # 0: FC writes the logits, ctc_greedy re-reads them (round-1 path)


def softmax_rows(x, C):
    _require_cuda(x, "softmax_rows")
    rows, ld = x.shape
    y = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_softmax_rows_f32(_lib.ptr(x), rows, C, ld, _lib.ptr(y), C, _lib.cur_stream()),
               "ptocr_softmax_rows_f32")
    return y


def db_head_tail(x, w1, b1, w2, b2):
    """x f32[N,H,W,64] -> maps f32[N,1,4H,4W]: both transposed convs of the DB head + sigmoid in one kernel"""
    _require_cuda(x, "db_head_tail")
    N, H, W, Cc = x.shape
    y = torch.empty((N, 1, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ptocr_db_head_tail_f32(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), C.c_float(b2), _lib.ptr(y),
                                                 N, H, W, Cc, _lib.cur_stream()), "ptocr_db_head_tail_f32")
    return y


class PackedModule(torch.nn.Module):
    """Mixin: packed (BN-folded, kernel-layout) weights are rebuilt whenever a parameter/buffer changed
    (load_state_dict, .to(device), in-place edits) -- detected through tensor versions and device."""

    def _sig(self):
        sig = []
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t._version, t.data_ptr(), str(t.device)))
        return tuple(sig)

    def packed(self):
        sig = self._sig()
        if getattr(self, "_packed_sig", None) != sig:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("pytorchocr_amd: model is on %s; move it to a cuda (ROCm) device -- no CPU fallback" % dev)
            self._packed = self._pack(dev)
            self._packed_sig = sig
        return self._packed

    def _check_eval(self):
        if self.training:
            raise NotImplementedError("pytorchocr_amd implements the inference (eval) hot path only; call .eval()")
