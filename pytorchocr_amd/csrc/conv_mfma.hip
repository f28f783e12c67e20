// fp32 implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chain).
//
// Replaces the ATen conv2d / batch_norm / relu / add / interpolate / cat / conv_transpose2d(k2,s2) calls of
//   pytocr/modeling/backbones/det_resnet.py:66-82,282-309   pytocr/modeling/necks/fpn.py:102-134
//   pytocr/modeling/heads/det_db_head.py:9-17                pytocr/modeling/backbones/rec_vgg.py:78-120
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin (tap-major, channel-minor: an NHWC tap is
// one contiguous Cin vector, so the im2col gather is 16-byte coalesced pieces).
//
// Block = 256 threads = 4 waves (one per SIMD).  Block tile BM x BN x 32, wave tile 64 x 64 = 2x2 MFMA tiles
// of 32x32 (64 accumulator VGPRs).  A (pixels x k) and B (cout x k) tiles live in LDS as [row][32+4] floats:
// the +4 pad makes every ds_read_b128 fragment read conflict-free (row stride 144 B => 16-B slot = 9*row mod 16).
// A lane reads 4 consecutive k with one ds_read_b128; lane half h takes k = 8*kk + 4*h + t for MFMA t, the same
// permutation of k on both operands, so the contraction is unchanged.
// Next tile's global loads are issued into registers before the MFMAs of the current tile (latency hides under
// 64 MFMAs x 64 cycles); LDS is single-buffered so 3 blocks fit a CU and cover each other's barriers.
// Epilogue (all fused): +bias (BN folded), residual add, ReLU, FPN nearest-x2 upsample-add, nearest-upsampled
// store into a channel slice of a wider tensor (concat in place), ConvTranspose 2x2/s2 pixel scatter.
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDT = BK + 4;   // LDS row stride in floats

struct ConvArgs {
    const float *x, *w, *bias, *res;
    float *y;
    int N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, Ho, Wo;
    int M, Kpad, nk;
    int relu, res_mode, out_up, out_ldc, out_coff, convt, co_real;
    int mtiles;
};

template <int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs p) {
    constexpr int WAVES_N = BN / 64;            // 1 or 2
    constexpr int WAVES_M = 4 / WAVES_N;        // 4 or 2
    static_assert(BM == WAVES_M * 64, "wave tile is 64x64");
    constexpr int A_ROWS = BM / 32;             // 16-B pieces per thread for A
    constexpr int B_ROWS = BN / 32;

    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
    float *As = smem;
    float *Bs = smem + BM * LDT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * 64;
    const int wn0 = (wave % WAVES_N) * 64;

    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs, so give each XCD a contiguous run of
    // M tiles (neighbouring strips share their halo rows in that XCD's L2).  Speed only.
    int bid = blockIdx.x;
    {
        const int nwg = p.mtiles, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = bid * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread gather descriptors: rows r = (tid>>3) + 32*i, piece c4 = tid&7
    const int c4 = tid & 7;
    const int lrow = tid >> 3;
    int a_iy0[A_ROWS], a_ix0[A_ROWS];
    long a_base[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; i++) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / (p.Ho * p.Wo);
            const int rem = m - n * (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad_h;
            a_ix0[i] = ox * p.stride - p.pad_w;
            a_base[i] = (long)n * p.H * p.W;
        } else {
            a_iy0[i] = -100000; a_ix0[i] = -100000; a_base[i] = 0;   // never in bounds -> zeros
        }
    }
    const float *wrow[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; i++) wrow[i] = p.w + (long)(n0 + lrow + 32 * i) * p.Kpad + c4 * 4;

    f32x4 ra[A_ROWS], rb[B_ROWS];
    auto gload = [&](int ks) {
        int kh, kw, c0;
        bool tap_ok = true;
        if (SMALLC) {                              // Cin == 4: one 16-B piece per tap
            const int tap = ks * 8 + c4;
            kh = tap / p.KW; kw = tap - kh * p.KW; c0 = 0;
            tap_ok = tap < p.KH * p.KW;
        } else {                                   // Cin % 32 == 0: the whole k-step sits in one tap
            const int k0 = ks * BK;
            const int tap = k0 / p.Cin;
            c0 = k0 - tap * p.Cin + c4 * 4;
            kh = tap / p.KW; kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4 *>(p.x + ((a_base[i] + (long)iy * p.W + ix) * p.Cin + c0));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; i++) rb[i] = *reinterpret_cast<const f32x4 *>(wrow[i] + ks * BK);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_ROWS; i++) *reinterpret_cast<f32x4 *>(&As[(lrow + 32 * i) * LDT + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; i++) *reinterpret_cast<f32x4 *>(&Bs[(lrow + 32 * i) * LDT + c4 * 4]) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    const float *a_frag = As + (wm0 + frow) * LDT + 4 * fh;
    const float *b_frag = Bs + (wn0 + frow) * LDT + 4 * fh;

    gload(0);
    lstore();
    __syncthreads();
    for (int ks = 0; ks < p.nk; ks++) {
        const bool more = ks + 1 < p.nk;
        if (more) gload(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(a_frag + kk * 8);
            f32x4 a1 = *reinterpret_cast<const f32x4 *>(a_frag + 32 * LDT + kk * 8);
            f32x4 b0 = *reinterpret_cast<const f32x4 *>(b_frag + kk * 8);
            f32x4 b1 = *reinterpret_cast<const f32x4 *>(b_frag + 32 * LDT + kk * 8);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    // ---- epilogue.  D layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (m >= p.M) continue;
            int n = 0, oy = 0, ox = 0;
            const bool need_pix = p.res_mode == PTOCR_RES_ADD_UP2_POST_RELU || p.out_up > 1 || p.convt;
            if (need_pix) {
                n = m / HoWo;
                const int rem = m - n * HoWo;
                oy = rem / p.Wo; ox = rem - oy * p.Wo;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = n0 + wn0 + j * 32 + frow;
                float v = acc[i][j][r] + p.bias[col];
                if (p.res_mode == PTOCR_RES_ADD_PRE_RELU) v += p.res[(long)m * p.Cout + col];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.res_mode == PTOCR_RES_ADD_UP2_POST_RELU)
                    v += p.res[(((long)n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + col];
                if (p.convt) {
                    const int ab = col / p.co_real, co = col - ab * p.co_real;
                    const int Y = 2 * oy + (ab >> 1), X = 2 * ox + (ab & 1);
                    p.y[(((long)n * (2 * p.Ho) + Y) * (2 * p.Wo) + X) * p.out_ldc + p.out_coff + co] = v;
                } else if (p.out_up > 1) {
                    const int U = p.out_up;
                    for (int dy = 0; dy < U; dy++)
                        for (int dx = 0; dx < U; dx++)
                            p.y[(((long)n * (p.Ho * U) + oy * U + dy) * (p.Wo * U) + ox * U + dx) * p.out_ldc + p.out_coff + col] = v;
                } else {
                    p.y[(long)m * p.out_ldc + p.out_coff + col] = v;
                }
            }
        }
    }
}

template <int BM, int BN, bool SMALLC>
static int launch_conv(const ConvArgs &a, hipStream_t s) {
    dim3 grid(a.mtiles, a.Cout / BN);
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, SMALLC>), grid, dim3(256), 0, s, a);
    return launch_ok("conv_mfma_kernel");
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_conv2d_f32(const ptocr_conv_desc *d, const float *d_x, const float *d_w, const float *d_bias,
                                const float *d_res, float *d_y, void *stream) {
    PT_CHECK(d && d_x && d_w && d_bias && d_y, "ptocr_conv2d_f32: null argument");
    PT_CHECK(d->Cin == 4 || d->Cin % 32 == 0, "ptocr_conv2d_f32: Cin must be 4 or a multiple of 32 (got %d)", d->Cin);
    PT_CHECK(d->Cout % 64 == 0, "ptocr_conv2d_f32: Cout must be a multiple of 64 (got %d)", d->Cout);
    PT_CHECK(d->stride >= 1 && d->KH >= 1 && d->KW >= 1 && d->out_up >= 1, "ptocr_conv2d_f32: bad geometry");
    PT_CHECK(d->Ho == (d->H + 2 * d->pad_h - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad_w - d->KW) / d->stride + 1,
             "ptocr_conv2d_f32: Ho/Wo do not match the conv geometry");
    PT_CHECK(d->res_mode == PTOCR_RES_NONE || d_res, "ptocr_conv2d_f32: residual mode without d_res");
    PT_CHECK(d->res_mode != PTOCR_RES_ADD_UP2_POST_RELU || (d->Ho % 2 == 0 && d->Wo % 2 == 0), "upsample-add needs even Ho,Wo");
    PT_CHECK(!(d->convt2x2 && (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->Cout % 4)), "convt2x2 is a 1x1 GEMM with Cout=4*Co");
    PT_CHECK((long)d->N * d->Ho * d->Wo < (1L << 31), "ptocr_conv2d_f32: too many output pixels");
    const int co_real = d->convt2x2 ? d->Cout / 4 : d->Cout;
    PT_CHECK(d->out_ldc >= d->out_coff + co_real, "ptocr_conv2d_f32: out_ldc too small");
    ConvArgs a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.Ho = d->Ho; a.Wo = d->Wo;
    a.M = d->N * d->Ho * d->Wo;
    const int K = d->KH * d->KW * d->Cin;
    a.Kpad = cdiv(K, BK) * BK;
    a.nk = a.Kpad / BK;
    a.relu = d->relu; a.res_mode = d->res_mode; a.out_up = d->out_up; a.out_ldc = d->out_ldc; a.out_coff = d->out_coff;
    a.convt = d->convt2x2; a.co_real = co_real;
    hipStream_t s = (hipStream_t)stream;
    const bool smallc = d->Cin == 4;
    if (d->Cout % 128 == 0 && !smallc) {
        a.mtiles = cdiv(a.M, 128);
        return launch_conv<128, 128, false>(a, s);
    }
    a.mtiles = cdiv(a.M, 256);
    return smallc ? launch_conv<256, 64, true>(a, s) : launch_conv<256, 64, false>(a, s);
}

// y[M,Nout] = x[M,K] @ w[Nout,K]^T + bias: the same MFMA kernel as a 1x1 convolution over M "pixels"
extern "C" int ptocr_linear_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int M, int K, int Nout,
                                int ldy, void *stream) {
    PT_CHECK(M >= 1 && K % 32 == 0 && Nout % 64 == 0 && ldy >= Nout, "ptocr_linear_f32: need K %% 32 == 0, Nout %% 64 == 0, ldy >= Nout");
    ptocr_conv_desc d;
    memset(&d, 0, sizeof d);
    d.N = 1; d.H = M; d.W = 1; d.Cin = K; d.Cout = Nout; d.KH = 1; d.KW = 1; d.stride = 1; d.Ho = M; d.Wo = 1;
    d.out_up = 1; d.out_ldc = ldy;
    return ptocr_conv2d_f32(&d, d_x, d_w, d_bias, nullptr, d_y, stream);
}
