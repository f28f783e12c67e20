// Pointwise (1x1) convolution with Cin = 32 or 64 and Cout a multiple of 32, up to 256 (+ folded BN, ReLU / Hardswish, optional
// FPN top-down add) on fp32 MFMA.  The layer it exists for is the FPN lateral `in2` (fpn.py:46-51,112,117): 64 -> 256 channels on the
// 1/4-resolution map -- 61.7 GFLOP per batch of 32 but 1.93 GB of output, i.e. bound by the HBM write, not by the MFMA.  The
// generic implicit GEMM (conv_mfma.hip) spends 1.56 ms on it (one short K loop per workgroup: all prologue and epilogue);
// here a persistent workgroup (256 threads, 1 per CU) keeps the whole weight matrix W[64][Cout] in LDS, streams 128-pixel
// tiles through a double-buffered LDS tile (loads of tile i+1 in flight while tile i computes) and stores 16 bytes per lane
// straight from the accumulators.
// MFMA roles: A = weights (rows = 32 output channels), B = pixels (columns = 32 pixels): a lane ends up with 4 consecutive
// output channels of ONE pixel per accumulator quad.  Wave w owns pixels [32w, 32w+32) of the tile and all Cout/32 channel
// blocks (up to 8 MFMA tiles = 128 accumulator registers), 32 k-steps of v_mfma_f32_32x32x2_f32.
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PW_TM = 128;                      // pixels per tile
constexpr int PW_MAXC = 256;

struct Pw64Args {
    const float *x, *w, *bias, *res;
    float *y;
    long M;                                     // pixels N*H*W
    int H, W, Cout, relu, res_up2, out_ldc, out_coff, ntiles;
    long x_bytes;
};

// Round 4: TWO TEAMS of four waves per workgroup (two waves per SIMD), half a period apart.  With one wave per SIMD the MFMA loop of a
// tile (16.4 k cycles) and its epilogue (32 sixteen-byte stores per lane, 1.93 GB per batch) took turns, on every CU at the same time:
// the matrix pipe idle while the chip stored, HBM idle while it multiplied (857 us against an MFMA floor of 393).  Now team A multiplies
// while team B stores and stages, then they swap -- the workgroup barrier is the phase clock, team 1 runs one barrier late.  A wave has
// 256 registers at two per SIMD, so a tile is taken in PASSES of at most four 32-channel blocks (64 accumulator registers + 64 for the
// top-down rows of the pass, requested when its MFMAs are issued and used one phase later; the next pixel tile is requested in the last
// store phase of this one and waits in 32 registers).  Each team has ONE pixel tile in LDS, the weights are shared.
#ifndef PW64_DBG
#define PW64_DBG 0          // timing experiments only: 1 no stores, 2 no top-down row loads
#endif
// (Knock-outs, tools/dbg/pw64_knock.sh: 745 us; 614 without the stores, 622 without the top-down rows, 607 with neither -- the MFMA
// loop alone runs at 65 % of the matrix pipe.  k-contiguous LDS layouts for both operands -- one 16-byte read per four k steps instead
// of four 4-byte ones -- measured WORSE: 810 us, 681 with both knock-outs; not kept.)
template <int PW_K, int NMT>                    // input channels (32 or 64), Cout / 32
__global__ __launch_bounds__(512, 1) void conv_pw64_kernel(Pw64Args p) {
    constexpr int NP = NMT > 4 ? 2 : 1, MTP = (NMT + NP - 1) / NP;           // passes per tile, channel blocks per pass (the last pass may have one fewer)
    constexpr int PW_RS = PW_K + 4;             // LDS row stride of the pixel tile (16-byte aligned rows)
    constexpr int QPP = PW_K / 4;               // 16-byte pieces per pixel
    constexpr int NPC = PW_TM * QPP / 256;      // pieces per thread of a team
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                           // [64][Cout]
    float *Xb = smem + PW_K * NMT * 32;         // [2 teams][PW_TM][PW_RS]
    float *Bl = Xb + 2 * PW_TM * PW_RS;         // [Cout] bias
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    constexpr int COUT = NMT * 32;

// Weights in LDS as [k / 2][cout][k & 1] (round 6): with the [k][COUT] layout the kh = 0 / kh = 1 halves of a wave read the same 32 banks
// (COUT % 64 == 0), a two-way conflict on every weight read.  Measured on one box, alternating builds: conv_pw128 (in3, with the top-down
// rows) 0.4375 -> 0.425 ms; conv_pw64 (in2) 0.834 -> 0.842 ms -- its two teams already keep the LDS pipe's turns apart -- so: off there.
#ifndef PW64_W_PAIRS
#define PW64_W_PAIRS 0
#endif
#ifndef PW128_W_PAIRS
#define PW128_W_PAIRS 1
#endif
#if PW64_W_PAIRS
    for (int i = threadIdx.x; i < PW_K * COUT / 4; i += 512) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(p.w)[i];
        const int k = (4 * i) / COUT, co = (4 * i) - k * COUT;
        float *dst = Wl + (((k >> 1) * COUT + co) << 1) + (k & 1);
        dst[0] = v[0]; dst[2] = v[1]; dst[4] = v[2]; dst[6] = v[3];
    }
#else
    for (int i = threadIdx.x; i < PW_K * COUT / 4; i += 512) reinterpret_cast<f32x4 *>(Wl)[i] = reinterpret_cast<const f32x4 *>(p.w)[i];
#endif
    if (threadIdx.x < COUT) Bl[threadIdx.x] = p.bias[threadIdx.x];

    // tile loader: 128 pixels x QPP float4 pieces, NPC per thread of the team; piece f = tid + 256 r -> pixel f / QPP, quad f % QPP
    f32x4 xreg[NPC];
    auto gload = [&](int tile) {
#pragma unroll
        for (int r = 0; r < NPC; r++) {
            const int f = tid + 256 * r;
            const long m = (long)tile * PW_TM + f / QPP;                  // beyond M: out of range -> zeros
            xreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(m * (PW_K * 4) + (f % QPP) * 16), 0, 0));
        }
    };
    float *Xt = Xb + team * PW_TM * PW_RS;
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < NPC; r++) {
            const int f = tid + 256 * r;
            *reinterpret_cast<f32x4 *>(Xt + (f / QPP) * PW_RS + (f % QPP) * 4) = xreg[r];
        }
    };

    const int c = lane & 31, kh = lane >> 5;
#if PW64_W_PAIRS
    const float *wp = Wl + 2 * c + kh;                           // W[2s + kh][32 mt + c] at ((s COUT + 32 mt + c) 2 + kh)
#else
    const float *wp = Wl + kh * COUT + c;                        // W[2s + kh][32 mt + c]
#endif
    const float *xp = Xt + (32 * wave + c) * PW_RS + kh;         // X[32w + c][2s + kh]

    f32x16 acc[MTP];
    f32x4 rres[MTP][4];
    auto multiply = [&](int tile, int pass) {
#pragma unroll
        for (int a = 0; a < MTP; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
        const float *wq = wp + (PW64_W_PAIRS ? 64 : 32) * MTP * pass;
#pragma unroll 4
        for (int s = 0; s < PW_K / 2; s++) {
            const float b = xp[2 * s];
#pragma unroll
            for (int mt = 0; mt < MTP; mt++)
                if (MTP * pass + mt < NMT) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[PW64_W_PAIRS ? s * 2 * COUT + 64 * mt : (2 * s) * COUT + 32 * mt], b, acc[mt], 0, 0, 0);
        }
        if (p.res_up2 && !(PW64_DBG & 2)) {                      // + nearest-upsampled coarser map, after the ReLU: used one phase later
            const long m = (long)tile * PW_TM + 32 * wave + c;
            const long mm = m < p.M ? m : 0;
            const int hw = p.H * p.W;
            const int n = (int)(mm / hw), rem = (int)(mm - (long)n * hw);
            const int oy = rem / p.W, ox = rem - oy * p.W;
            const float *rp = p.res + (((long)n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * COUT + 4 * kh + 32 * MTP * pass;
#pragma unroll
            for (int mt = 0; mt < MTP; mt++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    if (MTP * pass + mt < NMT) rres[mt][g] = *reinterpret_cast<const f32x4 *>(rp + 32 * mt + 8 * g);
        }
    };
    auto store_pass = [&](int tile, int pass) {                  // lane holds, for pixel m, output channels 32 mt + 8 g + 4 kh + {0..3} of the pass
        const long m = (long)tile * PW_TM + 32 * wave + c;
        if (m >= p.M) return;
        float *yp = p.y + m * p.out_ldc + p.out_coff + 32 * MTP * pass;
        const float *bl = Bl + 32 * MTP * pass;
#pragma unroll
        for (int mt = 0; mt < MTP; mt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (MTP * pass + mt >= NMT) continue;
                const int co = 32 * mt + 8 * g + 4 * kh;
                f32x4 v = f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]} + *reinterpret_cast<const f32x4 *>(bl + co);
                if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.relu == 2) {                                  // Hardswish: x * relu6(x + 3) / 6
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = v[k] * fminf(fmaxf(v[k] + 3.f, 0.f), 6.f) * (1.f / 6.f);
                }
                if (p.res_up2) v += rres[mt][g];
                if ((PW64_DBG & 1) && v[0] != 123.456f) continue;
                *reinterpret_cast<f32x4 *>(yp + co) = v;
            }
    };

    // stream t = 2 * workgroup + team takes tiles t, t + 2 G, t + 4 G, ..; every wave of the workgroup passes the same number of barriers
    // (2 per pass + 1: team 1 waits one phase at the start, team 0 at the end)
    const int step = 2 * (int)gridDim.x;
    const int iters = (p.ntiles + step - 1) / step;              // >= the tiles of either team
    int tile = 2 * (int)blockIdx.x + team;
    gload(tile);                                                 // beyond the last tile: zeros, never stored
    lstore();
    __syncthreads();
    if (team == 1) __syncthreads();
    for (int it = 0; it < iters; it++, tile += step) {
        const int next = tile + step;
        const bool have = tile < p.ntiles, has_next = next < p.ntiles;
#pragma unroll
        for (int pass = 0; pass < NP; pass++) {
            if (have) multiply(tile, pass);
            __syncthreads();                                     // (the other team's stores are issued)
            if (pass == NP - 1 && has_next) gload(next);         // the tile in LDS is consumed; the next one travels behind this phase's stores
            if (have) store_pass(tile, pass);
            if (pass == NP - 1 && has_next) lstore();
            __syncthreads();
        }
    }
    if (team == 0) __syncthreads();
}

template <int PW_K, int NMT>
static int launch_pw64(const Pw64Args &a, hipStream_t stream) {
    const size_t lds = sizeof(float) * (PW_K * NMT * 32 + 2 * PW_TM * (PW_K + 4) + NMT * 32);
    static DynLds dyn;
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&conv_pw64_kernel<PW_K, NMT>), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int grid = a.ntiles < 2 * n_cu ? (a.ntiles + 1) / 2 : n_cu;        // one persistent workgroup per CU, two tile streams (teams) in each
    hipLaunchKernelGGL((conv_pw64_kernel<PW_K, NMT>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    return launch_ok("conv_pw64_kernel");
}


// ---- the same scheme for 128 input channels (the FPN lateral in3: 128 -> 256 at 1/8 resolution, fpn.py:40-45,113,116): a workgroup
// owns HALF of the output channels (64 KB of weights in LDS) and the 128-pixel input tile is single-buffered (67 KB) -- the next
// tile waits in registers while this one computes, one more barrier per tile than the double-buffered form.  The generic kernel
// runs this layer at 47 % of the MFMA floor (eight short k-steps per workgroup: all prologue and epilogue).
template <int NMT>                              // output channels of a workgroup / 32
__global__ __launch_bounds__(256, 1) void conv_pw128_kernel(Pw64Args p, int cout_total) {
    constexpr int K = 128, RS = K + 4, QPP = K / 4, NPC = PW_TM * QPP / 256;        // 16 pieces per thread
    constexpr int COUT = NMT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                           // [K][COUT]
    float *Xb = smem + K * COUT;                // [PW_TM][RS]
    float *Bl = Xb + PW_TM * RS;                // [COUT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const int nsplit = cout_total / COUT;
    const int cbase = ((int)blockIdx.x % nsplit) * COUT;         // this workgroup's output channels: [cbase, cbase + COUT)
    const int first = (int)blockIdx.x / nsplit, stride = (int)gridDim.x / nsplit;

    for (int i = tid; i < K * COUT / 4; i += 256) {
        const int k = i / (COUT / 4), c4 = i - k * (COUT / 4);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p.w + (long)k * cout_total + cbase + c4 * 4);
#if PW128_W_PAIRS
        float *dst = Wl + (((k >> 1) * COUT + c4 * 4) << 1) + (k & 1);
        dst[0] = v[0]; dst[2] = v[1]; dst[4] = v[2]; dst[6] = v[3];
#else
        reinterpret_cast<f32x4 *>(Wl)[i] = v;
#endif
    }
    if (tid < COUT) Bl[tid] = p.bias[cbase + tid];

    f32x4 xreg[NPC];
    auto gload = [&](int tile) {
#pragma unroll
        for (int r = 0; r < NPC; r++) {
            const int f = tid + 256 * r;
            const long m = (long)tile * PW_TM + f / QPP;                  // beyond M: out of range -> zeros
            xreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(m * (K * 4) + (f % QPP) * 16), 0, 0));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < NPC; r++) {
            const int f = tid + 256 * r;
            *reinterpret_cast<f32x4 *>(Xb + (f / QPP) * RS + (f % QPP) * 4) = xreg[r];
        }
    };

    const int c = lane & 31, kh = lane >> 5;
#if PW128_W_PAIRS
    const float *wp = Wl + 2 * c + kh;
#else
    const float *wp = Wl + kh * COUT + c;                        // W[2s + kh][32 mt + c]
#endif
    const float *xp = Xb + (32 * wave + c) * RS + kh;            // X[32w + c][2s + kh]

    int tile = first;                                            // host launches gridDim.x / nsplit <= ntiles
    gload(tile);
    lstore();
    __syncthreads();
    for (;;) {
        const int next = tile + stride;
        const bool has_next = next < p.ntiles;
        if (has_next) gload(next);                               // in flight during this tile's MFMAs

        const long m = (long)tile * PW_TM + 32 * wave + c;
        const bool live = m < p.M;
        f32x4 rres[NMT][4];
        if (p.res_up2) {                                         // + nearest-upsampled coarser map, after the ReLU
            const long mm = live ? m : 0;
            const int hw = p.H * p.W;
            const int n = (int)(mm / hw), rem = (int)(mm - (long)n * hw);
            const int oy = rem / p.W, ox = rem - oy * p.W;
            const float *rp = p.res + (((long)n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * cout_total + cbase + 4 * kh;
#pragma unroll
            for (int mt = 0; mt < NMT; mt++)
#pragma unroll
                for (int g = 0; g < 4; g++) rres[mt][g] = *reinterpret_cast<const f32x4 *>(rp + 32 * mt + 8 * g);
        }
        f32x16 acc[NMT];
#pragma unroll
        for (int a = 0; a < NMT; a++) acc[a] = (f32x16)(0.f);
#pragma unroll 4
        for (int s = 0; s < K / 2; s++) {
            const float b = xp[2 * s];
#pragma unroll
            for (int mt = 0; mt < NMT; mt++)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[PW128_W_PAIRS ? s * 2 * COUT + 64 * mt : (2 * s) * COUT + 32 * mt], b, acc[mt], 0, 0, 0);
        }
        if (live) {
            float *yp = p.y + m * p.out_ldc + p.out_coff + cbase;
#pragma unroll
            for (int mt = 0; mt < NMT; mt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = 32 * mt + 8 * g + 4 * kh;
                    f32x4 v = f32x4{acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]} +
                              *reinterpret_cast<const f32x4 *>(Bl + co);
                    if (p.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (p.relu == 2) {
#pragma unroll
                        for (int k = 0; k < 4; k++) v[k] = v[k] * fminf(fmaxf(v[k] + 3.f, 0.f), 6.f) * (1.f / 6.f);
                    }
                    if (p.res_up2) v += rres[mt][g];
                    *reinterpret_cast<f32x4 *>(yp + co) = v;
                }
        }
        if (!has_next) break;
        __syncthreads();                                         // every wave is done with the tile in LDS
        lstore();
        __syncthreads();
        tile = next;
    }
}

static int launch_pw128(const Pw64Args &a, int cout_total, hipStream_t stream) {
    constexpr int NMT = 4;
    const size_t lds = sizeof(float) * (128 * NMT * 32 + PW_TM * (128 + 4) + NMT * 32);
    static DynLds dyn;
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&conv_pw128_kernel<NMT>), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int nsplit = cout_total / (NMT * 32);
    int per = n_cu / nsplit;                                     // persistent workgroups per output-channel block
    if (per > a.ntiles) per = a.ntiles;
    if (per < 1) per = 1;
    hipLaunchKernelGGL((conv_pw128_kernel<NMT>), dim3((unsigned)(per * nsplit)), dim3(256), lds, stream, a, cout_total);
    return launch_ok("conv_pw128_kernel");
}

}  // namespace ptocr

using namespace ptocr;

template <int PW_K>
static int dispatch_pw(const Pw64Args &a, hipStream_t s) {
    switch (a.Cout / 32) {
        case 1: return launch_pw64<PW_K, 1>(a, s);
        case 2: return launch_pw64<PW_K, 2>(a, s);
        case 3: return launch_pw64<PW_K, 3>(a, s);
        case 4: return launch_pw64<PW_K, 4>(a, s);
        case 5: return launch_pw64<PW_K, 5>(a, s);
        case 6: return launch_pw64<PW_K, 6>(a, s);
        case 7: return launch_pw64<PW_K, 7>(a, s);
        default: return launch_pw64<PW_K, 8>(a, s);
    }
}

// d_x: f32[N,H,W,Cin], Cin = 32, 64 or 128 (128: Cout a multiple of 128); d_w: f32[Cin][Cout] (k-major, BN folded, zero rows / columns for padded channels);
// act: 0 none, 1 ReLU, 2 Hardswish; d_res (res_up2 = 1): f32[N,H/2,W/2,Cout], added AFTER the activation (fpn.py:133-134), H and W
// even; d_y: f32[N,H,W,out_ldc], channels [out_coff, out_coff + Cout).
extern "C" int ptocr_conv1x1_small_k_f32(const float *d_x, const float *d_w, const float *d_bias, const float *d_res, float *d_y,
                                         int N, int H, int W, int Cin, int Cout, int act, int res_up2, int out_ldc, int out_coff,
                                         void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y, "ptocr_conv1x1_small_k_f32: null argument");
    PT_CHECK(N > 0 && H > 0 && W > 0, "ptocr_conv1x1_small_k_f32: empty tensor");
    PT_CHECK(Cin == 32 || Cin == 64 || Cin == 128, "ptocr_conv1x1_small_k_f32: Cin must be 32, 64 or 128 (got %d)", Cin);
    PT_CHECK(Cout % 32 == 0 && Cout >= 32 && Cout <= PW_MAXC, "ptocr_conv1x1_small_k_f32: Cout must be a multiple of 32 up to %d", PW_MAXC);
    PT_CHECK(Cin != 128 || Cout % 128 == 0, "ptocr_conv1x1_small_k_f32: with 128 input channels Cout must be a multiple of 128");
    PT_CHECK(act >= 0 && act <= 2, "ptocr_conv1x1_small_k_f32: activation must be none, ReLU or Hardswish");
    PT_CHECK(!res_up2 || (d_res && H % 2 == 0 && W % 2 == 0), "ptocr_conv1x1_small_k_f32: the upsample-add needs d_res and even H, W");
    PT_CHECK(out_ldc % 4 == 0 && out_coff % 4 == 0 && out_ldc >= out_coff + Cout, "ptocr_conv1x1_small_k_f32: channel strides must be multiples of 4");
    Pw64Args a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.M = (long)N * H * W; a.H = H; a.W = W; a.Cout = Cout; a.relu = act; a.res_up2 = res_up2; a.out_ldc = out_ldc; a.out_coff = out_coff;
    a.x_bytes = a.M * Cin * 4;
    PT_CHECK(a.x_bytes < (1L << 31), "ptocr_conv1x1_small_k_f32: tensor larger than 2 GiB");
    a.ntiles = (int)((a.M + PW_TM - 1) / PW_TM);
    if (Cin == 128) return launch_pw128(a, Cout, (hipStream_t)stream);
    return Cin == 32 ? dispatch_pw<32>(a, (hipStream_t)stream) : dispatch_pw<64>(a, (hipStream_t)stream);
}

// the Cin = 64, ReLU-or-none form under its first name (the FPN lateral in2 of DBNet-r18)
extern "C" int ptocr_conv1x1_k64_f32(const float *d_x, const float *d_w, const float *d_bias, const float *d_res, float *d_y,
                                     int N, int H, int W, int Cout, int relu, int res_up2, int out_ldc, int out_coff, void *stream) {
    PT_CHECK(relu == 0 || relu == 1, "ptocr_conv1x1_k64_f32: activation must be none or ReLU");
    return ptocr_conv1x1_small_k_f32(d_x, d_w, d_bias, d_res, d_y, N, H, W, 64, Cout, relu, res_up2, out_ldc, out_coff, stream);
}
