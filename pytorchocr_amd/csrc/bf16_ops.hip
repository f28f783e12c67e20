// bf16 inference path of the MobileNetV3 detector (BASELINE.json configs[3]: DBNet mbv3-small x1.0, bf16): NHWC bf16 activations in
// HBM, v_mfma_f32_32x32x16_bf16 with fp32 accumulation for every 1x1 / 3x3 GEMM-shaped layer, fp32 arithmetic in the depthwise /
// squeeze-excitation / head-tail kernels, fp32 probability maps out.
//
// Replaces (same layers as the fp32 kernels, reference file:line): ConvBNActivation / InvertedResidual / SqueezeExcitation of
// pytocr/modeling/backbones/det_mobilenet_v3.py:38-151,154-279, FPN of necks/fpn.py:102-134, DBHead of heads/det_db_head.py:9-17,47-50.
//
// Why it looks unlike the fp32 conv kernels: this network is HBM-bound (8.4 GFLOP per 736x1280 image against ~60 MB of activations),
// bf16 MFMA is 16x the fp32 rate, so nothing is staged through LDS -- both MFMA operands are read from global memory in their fragment
// layout (a lane's 8 consecutive k values are 16 contiguous bytes of an NHWC pixel or of a K-contiguous weight row), weights stay in
// L2/L1, every activation is read once and written once, and what a layer's neighbours can do on the way (BN, activation, residual,
// SE scale on the input, SE pooling on the output, nearest upsample + concat, FPN top-down add) is fused into it.
//
// MFMA roles: A = weights (row = output channel), B = pixels (column = pixel).  The accumulator of lane (pixel = lane & 31,
// half = lane >> 5) then holds, per register quad g, four CONSECUTIVE output channels 8g + 4*half + {0..3} of its pixel: one packed
// 8-byte bf16 store per quad, no shuffle, no LDS.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace ptocr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float actf(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);       // Hardswish
    return v;
}

// The activation code is a compile-time parameter of the hot kernels: with a run-time code the compiler turned every actf() into a
// tree of scalar branches (the code is wave-uniform) -- 144 branches in one 3x3 epilogue, ~100 per depthwise run.
template <int ACT>
__device__ __forceinline__ float actc(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);       // Hardswish
    return v;
}
#define PT_ACT_SWITCH(act, F) do { if ((act) == 0) { F(0); } else if ((act) == 1) { F(1); } else { F(2); } } while (0)

__device__ __forceinline__ bf16x8 zero8() { bf16x8 z; for (int j = 0; j < 8; j++) z[j] = (__bf16)0.f; return z; }

// ---------------------------------------------------------------------------------------------- 1x1 convolution
struct PwArgs {
    const __bf16 *x, *w, *res;
    const float *bias, *scale;      // scale: SE gate f32[N][Cin] applied to the INPUT (x * scale, rounded to bf16 once), or null
    __bf16 *y;
    long M;                         // pixels
    int Cin;                        // padded input channels (multiple of 16) == channel stride of x
    int ntile;                      // 32-wide output channel tiles (weights are zero-padded to ntile * 32 rows)
    int cstore;                     // channels written (multiple of 4)
    int act, res_mode;              // res_mode 0 none / 1 add before the activation (same shape) / 2 nearest-x2 upsampled add after it
    int H, W;                       // output geometry (res_mode 2, and the image index of a pixel: n = m / (H * W))
    int out_ldc, out_coff, res_ldc;
};

#ifndef PW_DBG
#define PW_DBG 0           // timing experiments only: 1 weight loads from one address, 2 pixel loads from one address, 4 no SE scaling
#endif
template <int NT, bool RES, int ACT, int GRP>
__global__ __launch_bounds__(256) void pw_bf16_kernel(PwArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long m = ((long)blockIdx.x * 4 + wave) * 32 + r;
    const long mc = m < p.M ? m : p.M - 1;
    const int t0 = blockIdx.y * NT;
    const __bf16 *xrow = p.x + mc * p.Cin + 8 * h;
    const long HW = (long)p.H * p.W;
    const float *srow = p.scale ? p.scale + (mc / HW) * p.Cin + 8 * h : nullptr;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const __bf16 *wrow = p.w + ((long)t0 * 32 + r) * p.Cin + 8 * h;
    const int nks = p.Cin >> 4;
    // (round 6) The bias of the workgroup's NT * 32 output channels goes to LDS HERE, before the K loop.  It used to be read from global
    // memory inside the epilogue, a 16-byte load per (tile, channel group), each followed by `s_waitcnt vmcnt(0)` -- and the vector-memory
    // counter is in order and counts STORES: every one of those twelve loads waited for its own round trip AND for the acknowledgement of
    // the tile's stores issued before it (the pattern DESIGN 3.0 found in the Winograd epilogue; here it was most of a small-map launch).
    // (requested here, parked in LDS behind the K loop: the load's round trip overlaps the loop's)
    __shared__ float sbias[NT * 32];
    float bias_mine = 0.f;
    if (threadIdx.x < NT * 32) { const int c = t0 * 32 + (int)threadIdx.x; if (c < p.cstore) bias_mine = p.bias[c]; }
    // K loop in groups of four slices, all loads of a group issued before its first MFMA, no branch inside: the small maps have one or
    // two waves per SIMD and Cin up to 576 -- with one slice per iteration every MFMA waited for its own operands' round trip to L2
    // (36 dependent round trips for the 576-channel layers).  GRP = 1 (Cin < 64) keeps the one-slice loop: the 64 extra registers of the
    // group cost the large-map layers a third of their waves (97 -> 129 us).  Tiles beyond ntile read tile 0's rows (valid memory) and are dropped below.
    const __bf16 *wr[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) wr[t] = wrow + (t0 + t < p.ntile ? (long)t * 32 * p.Cin : 0L);
    auto scaled = [&](bf16x8 b, int ks) {
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(srow + ks * 16), s1 = *reinterpret_cast<const f32x4 *>(srow + ks * 16 + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { b[j] = (__bf16)((float)b[j] * s0[j]); b[4 + j] = (__bf16)((float)b[4 + j] * s1[j]); }
        return b;
    };
    int ks = 0;
    if constexpr (GRP > 1)
    for (; ks + GRP <= nks; ks += GRP) {
        bf16x8 b[GRP], a[NT][GRP];
#pragma unroll
        for (int u = 0; u < GRP; u++) b[u] = *reinterpret_cast<const bf16x8 *>(xrow + ((PW_DBG & 2) ? u : ks + u) * 16);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int u = 0; u < GRP; u++) a[t][u] = *reinterpret_cast<const bf16x8 *>(wr[t] + ((PW_DBG & 1) ? u : ks + u) * 16);
        if (srow && !(PW_DBG & 4)) {
#pragma unroll
            for (int u = 0; u < GRP; u++) b[u] = scaled(b[u], ks + u);
        }
#pragma unroll
        for (int u = 0; u < GRP; u++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][u], b[u], acc[t], 0, 0, 0);
    }
    for (; ks < nks; ks++) {
        bf16x8 b = *reinterpret_cast<const bf16x8 *>(xrow + ks * 16);
        bf16x8 a[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) a[t] = *reinterpret_cast<const bf16x8 *>(wr[t] + ks * 16);
        if (srow) b = scaled(b, ks);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b, acc[t], 0, 0, 0);
    }
    // Epilogue through LDS: a lane's accumulator quads are 8-byte pieces of its pixel's row, 160 bytes apart from the next lane's --
    // stored directly they reach HBM as partial sectors (measured 2.4x write amplification).  Each wave parks its 32 pixels x
    // NT*32 channels in its own LDS tile and writes it back row-major, 16 bytes per lane, consecutive lanes on consecutive
    // chunks of a pixel row.  Everything is added in fp32 BEFORE the one rounding to bf16.
    // RES = false: bias + activation on the way in, the tile holds bf16.
    // RES = true (block residual / FPN top-down add): the tile holds fp32 (conv + bias) and the residual is read in the write-back
    // phase, where it is one coalesced 16-byte load per lane -- read on the way in it was twelve 8-byte loads per lane, 192 bytes
    // apart from the neighbouring lane's.
    // One 32-channel tile at a time (4.6 KB of LDS per wave in fp32 instead of 12.8 KB for three): the kernel is latency-bound on the
    // large maps and LDS was what limited the waves per CU (12 -> 32).  A pixel's 32 channels are four 16-byte chunks: lane l takes
    // chunk l % 4 of pixel l / 4 (+ 16 in the second round) -- no division anywhere; for the top-down add the wave's first pixel is
    // decomposed once and a pixel's (n, y, x) follows by carries (run-time integer divisions, ~40 VALU instructions each, four per item,
    // made the first version instruction-bound).
    constexpr int TW = RES ? 32 + 4 : 32 + 8;                   // tile row stride in elements (16-byte padded: conflict-free 16-B accesses)
    typedef typename std::conditional<RES, float, __bf16>::type tile_t;
    __shared__ __attribute__((aligned(16))) tile_t tile[4][32][TW];
    if (threadIdx.x < NT * 32) sbias[threadIdx.x] = bias_mine;
    __syncthreads();
    const long mw0 = ((long)blockIdx.x * 4 + wave) * 32;
    int n0 = 0, oy0 = 0, ox0 = 0;
    if (RES && p.res_mode == 2) {
        n0 = (int)(mw0 / HW);
        const int rem = (int)(mw0 - (long)n0 * HW);
        oy0 = rem / p.W; ox0 = rem - oy0 * p.W;
    }
    const int ch = lane & 3;
    // (round 6) RES: the residual pieces of ALL tiles are requested here, before the first store of the epilogue -- read inside the
    // write-back phase, each load's wait was also a wait for the stores issued before it (in-order counter), once per tile and round
    bf16x4 rres[RES ? NT : 1][2][2];
    if constexpr (RES) {
#pragma unroll
        for (int round = 0; round < 2; round++) {
            const int pr = (lane >> 2) + 16 * round;
            const long mm = mw0 + pr;
            const __bf16 *rpix;
            if (p.res_mode == 1) rpix = p.res + mm * p.res_ldc;
            else {
                int n = n0, oy = oy0, ox = ox0 + pr;
                while (ox >= p.W) { ox -= p.W; oy++; }
                while (oy >= p.H) { oy -= p.H; n++; }
                rpix = p.res + (((long)n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.res_ldc;
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int tbase = (t0 + t) * 32, tend = min(p.cstore, tbase + 32), c = tbase + ch * 8;
                // (straight-line loads from addresses that are always valid -- the tensor's first bytes where the piece does not exist, its
                // value is never used --, so that all of them are in flight together: one round trip)
                const bool on = t0 + t < p.ntile && c < tend && mm < p.M;
                const __bf16 *a0 = on ? rpix + c : p.res, *a1 = (on && c + 8 <= tend) ? rpix + c + 4 : a0;
                rres[t][round][0] = *reinterpret_cast<const bf16x4 *>(a0);
                rres[t][round][1] = *reinterpret_cast<const bf16x4 *>(a1);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int tbase = (t0 + t) * 32, tend = min(p.cstore, tbase + 32);
        if (t0 + t >= p.ntile || tend <= tbase) break;          // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int c = tbase + 8 * g + 4 * h;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < tend) {
                const f32x4 bias = *reinterpret_cast<const f32x4 *>(&sbias[32 * t + 8 * g + 4 * h]);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = acc[t][4 * g + j] + bias[j];
                if (!RES) {
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = actc<ACT>(v[j]);
                }
            }
            if constexpr (RES) {
                *reinterpret_cast<f32x4 *>(&tile[wave][r][8 * g + 4 * h]) = v;
            } else {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4 *>(&tile[wave][r][8 * g + 4 * h]) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int c = tbase + ch * 8;
        const bool full = c + 8 <= tend;
#pragma unroll
        for (int round = 0; round < 2; round++) {
            const int pr = (lane >> 2) + 16 * round;
            const long mm = mw0 + pr;
            if (c >= tend || mm >= p.M) continue;
            __bf16 *dst = p.y + mm * p.out_ldc + p.out_coff + c;
            if constexpr (RES) {
                const bf16x4 r0 = rres[t][round][0], r1 = rres[t][round][1];
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&tile[wave][pr][ch * 8]), v1 = *reinterpret_cast<const f32x4 *>(&tile[wave][pr][ch * 8 + 4]);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float a0 = v0[j], a1 = v1[j];
                    if (p.res_mode == 1) { a0 = actc<ACT>(a0 + (float)r0[j]); a1 = actc<ACT>(a1 + (float)r1[j]); }
                    else { a0 = actc<ACT>(a0) + (float)r0[j]; a1 = actc<ACT>(a1) + (float)r1[j]; }
                    o[j] = (__bf16)a0; o[4 + j] = (__bf16)a1;
                }
                if (full) *reinterpret_cast<bf16x8 *>(dst) = o;
                else { bf16x4 oh = {o[0], o[1], o[2], o[3]}; *reinterpret_cast<bf16x4 *>(dst) = oh; }
            } else {
                if (full) *reinterpret_cast<bf16x8 *>(dst) = *reinterpret_cast<const bf16x8 *>(&tile[wave][pr][ch * 8]);
                else *reinterpret_cast<bf16x4 *>(dst) = *reinterpret_cast<const bf16x4 *>(&tile[wave][pr][ch * 8]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                        // the tile is read: the next 32 channels may overwrite it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------- 3x3 / stride 1 / pad 1 convolution
// (the FPN smoothing convs and the head's first conv: Cin = 96 -> Cout <= 32).  Implicit GEMM over K = 9 * Cin, one 32-channel
// output tile per wave; the nine taps of a pixel are nine 16-byte-piece reads of neighbouring NHWC pixels (L1 / L2 serve the re-use).
struct C3Args {
    const __bf16 *x, *w;
    const float *bias;
    __bf16 *y;
    long M;
    int H, W, Cin, cstore, act, out_up, out_ldc, out_coff;
};

__global__ __launch_bounds__(256) void conv3x3_bf16_kernel(C3Args p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long m = ((long)blockIdx.x * 4 + wave) * 32 + r;
    const long mc = m < p.M ? m : p.M - 1;
    const long HW = (long)p.H * p.W;
    const int n = (int)(mc / HW);
    const int rem = (int)(mc - (long)n * HW);
    const int oy = rem / p.W, ox = rem - oy * p.W;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const int ncs = p.Cin >> 4;
    const __bf16 *wrow = p.w + (long)r * 9 * p.Cin + 8 * h;
    for (int tap = 0; tap < 9; tap++) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const __bf16 *src = p.x + (((long)n * p.H + (ok ? iy : oy)) * p.W + (ok ? ix : ox)) * p.Cin + 8 * h;
        for (int cs = 0; cs < ncs; cs++) {
            bf16x8 b = *reinterpret_cast<const bf16x8 *>(src + cs * 16);
            if (!ok) b = zero8();
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wrow + tap * p.Cin + cs * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    if (m >= p.M) return;
    const int U = p.out_up;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int c = 8 * g + 4 * h;
        if (c >= p.cstore) continue;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (__bf16)actf(acc[4 * g + j] + bias[j], p.act);
        for (int uy = 0; uy < U; uy++)
            for (int ux = 0; ux < U; ux++)
                *reinterpret_cast<bf16x4 *>(p.y + (((long)n * p.H * U + oy * U + uy) * ((long)p.W * U) + ox * U + ux) * p.out_ldc + p.out_coff + c) = o;
    }
}

// The same layer with the input staged through LDS: a workgroup owns a tile of 8 rows x 32 columns of output pixels, loads the
// 10 x 34 halo patch once (every input byte leaves HBM / L2 once, 1.33x with the halo, instead of nine times through a thrashing
// L1), and its four waves (two output rows each) read their B fragments from LDS.  Pixel stride in LDS = 2 Cin + 16 bytes: the 16
// lanes of a ds_read_b128 group then fall on 16 different 16-byte bank slots (13 r mod 16 is a permutation).
// NCS = Cin / 16 at compile time (0: run-time count): the tap loop is unrolled and the weight fragments of tap t+1 are requested before
// the MFMAs of tap t -- with a run-time inner loop every MFMA pair waited for its own 16-byte weight load from global memory (54
// dependent L1 round trips per tile, several times the MFMA time).
#ifndef C3_DBG
#define C3_DBG 0             // timing experiments only: 1 no patch loads, 2 no MFMA loop (one tap), 4 no stores, 8 no weight prefetch
#endif
constexpr int C3_TH = 8, C3_TW = 32, C3_MAXCIN = 96;
#if C3_DBG & 16
__device__ unsigned long long c3_dbg_t[16 * 10];
#define C3_PROBE(i) do { if (blockIdx.x == 8 && threadIdx.x == 0 && c3_it >= 2 && c3_it < 18) c3_dbg_t[(c3_it - 2) * 10 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ptocr_c3_dbg_read(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3_dbg_t), sizeof(c3_dbg_t)); }
#else
#define C3_PROBE(i)
#endif
template <int NCS>
__global__ __launch_bounds__(256) void conv3x3_bf16_lds_kernel(C3Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[(C3_TH + 2) * (C3_TW + 2) * (2 * C3_MAXCIN + 16)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = cdiv(p.W, C3_TW);
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * C3_TH, tx0 = (blockIdx.x % tiles_x) * C3_TW;
    const int nch = NCS > 0 ? 2 * NCS : p.Cin >> 3;             // compile-time where it can be: the piece -> (pixel, chunk) divisions
    const int stride = 16 * nch + 16;
    const __bf16 *ximg = p.x + (long)n * p.H * p.W * p.Cin;
    // all loads of a thread first (16 in flight for Cin = 96), then the LDS stores: load -> store per iteration is 16 serial round trips
    constexpr int C3_NLD = ((C3_TH + 2) * (C3_TW + 2) * (C3_MAXCIN / 8) + 255) / 256;
    const int npiece = (C3_TH + 2) * (C3_TW + 2) * nch;
    bf16x8 pv[C3_NLD];
#pragma unroll
    for (int k = 0; k < C3_NLD; k++) {
        const int i = threadIdx.x + 256 * k;
        const int pix = i / nch, ch = i - pix * nch;
        const int iy = ty0 - 1 + pix / (C3_TW + 2), ix = tx0 - 1 + pix % (C3_TW + 2);
        pv[k] = zero8();
        if (!(C3_DBG & 1) && i < npiece && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            pv[k] = *reinterpret_cast<const bf16x8 *>(ximg + ((long)iy * p.W + ix) * p.Cin + ch * 8);
    }
#pragma unroll
    for (int k = 0; k < C3_NLD; k++) {
        const int i = threadIdx.x + 256 * k;
        const int pix = i / nch, ch = i - pix * nch;
        if (i < npiece) *reinterpret_cast<bf16x8 *>(patch + pix * stride + ch * 16) = pv[k];
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
    const __bf16 *wrow = p.w + (long)r * 9 * p.Cin + 8 * h;
    if (C3_DBG & 2) { acc[0][0] = patch[threadIdx.x]; }
    else if (NCS > 0) {
        bf16x8 a_cur[NCS > 0 ? NCS : 1], a_nxt[NCS > 0 ? NCS : 1];
#pragma unroll
        for (int cs = 0; cs < NCS; cs++) a_cur[cs] = *reinterpret_cast<const bf16x8 *>(wrow + cs * 16);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3, dx = tap % 3;               // patch coordinates of the tap: (row + dy, col + dx)
            const unsigned char *b0 = patch + ((2 * wave + dy) * (C3_TW + 2) + r + dx) * stride + 16 * h;
            const unsigned char *b1 = b0 + (C3_TW + 2) * stride;
            if (tap < 8) {
#pragma unroll
                for (int cs = 0; cs < NCS; cs++) a_nxt[cs] = *reinterpret_cast<const bf16x8 *>(wrow + (tap + 1) * p.Cin + cs * 16);
            }
#pragma unroll
            for (int cs = 0; cs < NCS; cs++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[cs], *reinterpret_cast<const bf16x8 *>(b0 + cs * 32), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[cs], *reinterpret_cast<const bf16x8 *>(b1 + cs * 32), acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int cs = 0; cs < NCS; cs++) a_cur[cs] = a_nxt[cs];
        }
    } else {
        const int ncs = p.Cin >> 4;
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3, dx = tap % 3;
            const unsigned char *b0 = patch + ((2 * wave + dy) * (C3_TW + 2) + r + dx) * stride + 16 * h;
            const unsigned char *b1 = b0 + (C3_TW + 2) * stride;
            for (int cs = 0; cs < ncs; cs++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wrow + tap * p.Cin + cs * 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *reinterpret_cast<const bf16x8 *>(b0 + cs * 32), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *reinterpret_cast<const bf16x8 *>(b1 + cs * 32), acc[1], 0, 0, 0);
            }
        }
    }
    const int U = p.out_up, ox = tx0 + r;
    if (ox >= p.W) return;
    if ((C3_DBG & 4) && acc[0][0] != 123.25f) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int oy = ty0 + 2 * wave + j;
        if (oy >= p.H) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int c = 8 * g + 4 * h;
            if (c >= p.cstore) continue;
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + c);
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = (__bf16)actf(acc[j][4 * g + q] + bias[q], p.act);
            for (int uy = 0; uy < U; uy++)
                for (int ux = 0; ux < U; ux++)
                    *reinterpret_cast<bf16x4 *>(p.y + (((long)n * p.H * U + oy * U + uy) * ((long)p.W * U) + ox * U + ux) * p.out_ldc + p.out_coff + c) = o;
        }
    }
}

// Persistent form for Cin = 16 NCS: a workgroup walks tiles (XCD-aware order), the patch of tile t+1 travels global -> registers while
// tile t computes, and the outputs leave through LDS as 16-byte pieces (a pixel's cstore channels are contiguous: 48 bytes for 24
// channels = three pieces instead of six 8-byte stores into a 192-byte-stride concat buffer).  History of the 32 x 184 x 320 layer:
// non-persistent 345 us (patch loads 170 + MFMA loop 185 + stores 105, added up); persistent with the weight fragments fetched per
// tap from global memory 320 us -- memory returns in order, so every such fetch waited behind the 16 loads of the NEXT patch
// issued just before the loop and the prefetch never overlapped anything (155 us without the in-loop fetches); all 54 fragments
// resident in registers (216 VGPRs, one wave per SIMD) 200 us; taller tiles (12 / 16 rows) spill: 305 / 540 us.
// Eight waves, two per SIMD, so that one wave's LDS reads and address arithmetic hide behind the other's MFMAs.  The 216 weight registers
// do not fit twice in a SIMD's file, so the K dimension is split: waves 0-3 hold the weight fragments of channel slices [0, NCS/2),
// waves 4-7 of [NCS/2, NCS) (108 registers each); both halves compute the same two output rows per wave, swap one row's partial
// sums through LDS and finish one row each.  s_memtime probes (-DC3_DBG=16, tools/dbg/c3_probe.sh), cycles per tile: first
// eight-wave version 14.2 k (prefetch address arithmetic 1.35 k, MFMA phase 4.8 k, partial-sum exchange with 64-way bank conflicts
// + an epilogue of 144 scalar branches around run-time activation codes and a bias load queued behind the prefetch 5.8 k, stores
// 1.15 k, patch to LDS 0.8 k); now 9.6 k (0.8 / 4.8 / 2.0 / 0.65 / 0.5), 155 us.
// PLANES (round 4): the input is FOUR tensors [N,H,W,Cin/4] one after the other (the FPN's concat kept as four planes: each smoothing
// conv then writes whole cache lines -- a 48-byte slice of a 192-byte pixel of the concat buffer is a partial line for each of the four
// writers) and the patch is gathered plane by plane: the same pieces land at the same LDS offsets, everything after the staging is shared.
template <int NCS, int ACT, bool PLANES = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_bf16_pers8_kernel(C3Args p, int tiles_x, int tiles_per_img, int total) {
    constexpr int TH = 8, HC = NCS / 2;
    constexpr int NCH = 2 * NCS, STRIDE = 16 * NCH + 16;
    constexpr int NPL = PLANES ? 4 : 1, PCH = NCH / NPL;             // planes, 16-byte pieces of a pixel in one plane
    static_assert(NCH % NPL == 0, "planes split the channels evenly");
    constexpr int NPIECE = (TH + 2) * (C3_TW + 2) * PCH;             // pieces per plane
    constexpr int NLD = (NPIECE + 511) / 512;                        // loads per thread and plane
    __shared__ __attribute__((aligned(16))) unsigned char patch[(TH + 2) * (C3_TW + 2) * STRIDE];
    static_assert(sizeof(patch) >= 4 * 64 * 32 * 4 + TH * C3_TW * 32 * 2, "patch buffer doubles as partial-sum + output staging");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rw = wave & 3, half = wave >> 2;
    bf16x8 pv[NPL][NLD];
    int n, ty0, tx0;
    auto decode = [&](int t) {
        n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        ty0 = (rem / tiles_x) * TH; tx0 = (rem % tiles_x) * C3_TW;
    };
    const int pcin = p.Cin / NPL;                                    // channels of one plane
    // Tile-invariant part of the patch addressing, once per thread: piece k of this thread is pixel (py, px) of the patch, channels
    // 8 ch..  Per tile the loads go through a buffer descriptor of the IMAGE: rows above and below it fall outside the descriptor
    // and return zeros by themselves, columns left and right of it are sent out of range by hand -- four VALU instructions per piece
    // (64-bit addresses, two range checks and a branch around every load were 27, 1.7 k cycles per tile).
    unsigned pbyte[NLD];
    int ppx[NLD], lds_off[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int i = threadIdx.x + 512 * k;
        const int pix = i / PCH, ch = i - pix * PCH;
        const int py = pix / (C3_TW + 2), px = pix - py * (C3_TW + 2);
        pbyte[k] = (unsigned)(((py * p.W + px) * pcin + ch * 8) * 2);
        ppx[k] = i < NPIECE ? px : (1 << 20);                    // beyond any width: never loaded, never stored
        lds_off[k] = pix * STRIDE + ch * 16;
    }
    const int img_bytes = p.H * p.W * pcin * 2;                  // < 2^31 (host check)
    const long plane_elems = p.M * pcin;                         // N * H * W * (channels of a plane)
    auto gload = [&]() {
        const unsigned org = (unsigned)(((ty0 - 1) * p.W + (tx0 - 1)) * pcin * 2);       // negative for the first row / column: wraps out of range
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.x + pl * plane_elems + (long)n * p.H * p.W * pcin), 0, img_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const unsigned vo = (unsigned)(tx0 - 1 + ppx[k]) < (unsigned)p.W ? org + pbyte[k] : 0xffffffffu;
                pv[pl][k] = (C3_DBG & 1) ? zero8() : __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, 0, 0));
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
            for (int k = 0; k < NLD; k++)
                if (ppx[k] < (1 << 20)) *reinterpret_cast<bf16x8 *>(patch + lds_off[k] + pl * PCH * 16) = pv[pl][k];
    };
    const __bf16 *wrow = p.w + (long)r * 9 * p.Cin + 8 * h + half * HC * 16;
    bf16x8 aw[9][HC];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int cs = 0; cs < HC; cs++) aw[tap][cs] = *reinterpret_cast<const bf16x8 *>(wrow + tap * p.Cin + cs * 16);
    const int U = p.out_up, CS = p.cstore, PCS = CS >> 3;            // 16-byte pieces per output pixel
    f32x4 biasr[4];                                                  // a global load inside the tile loop would queue behind the prefetch (in-order return)
#pragma unroll
    for (int g = 0; g < 4; g++) biasr[g] = 8 * g + 4 * h < CS ? *reinterpret_cast<const f32x4 *>(p.bias + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};       // cstore % 4 == 0
    constexpr int NWB = 2;                                           // write-back items per thread: TH * C3_TW * PCS <= 2 * 512 (PCS <= 4)
    int wb_src[NWB], wb_yx[NWB], wb_part[NWB];
#pragma unroll
    for (int w = 0; w < NWB; w++) {
        const int i = threadIdx.x + 512 * w;
        const int px = i / PCS, part_i = i - px * PCS;
        wb_src[w] = i < TH * C3_TW * PCS ? px * 32 + part_i * 8 : -1;
        wb_yx[w] = ((px / C3_TW) << 16) | (px % C3_TW);
        wb_part[w] = part_i * 8;
    }

    int tile, tstep, tend;                                           // XCD-aware tile order, as in the four-wave kernel
    if ((gridDim.x & 7) == 0) {
        const int j = blockIdx.x & 7;
        tile = (int)((long)j * total / 8) + (int)(blockIdx.x >> 3);
        tstep = (int)(gridDim.x >> 3);
        tend = (int)((long)(j + 1) * total / 8);
    } else { tile = blockIdx.x; tstep = gridDim.x; tend = total; }
    if (tile >= tend) return;
    decode(tile);
    gload();
    lstore();
    __syncthreads();
    f32x4 *part = reinterpret_cast<f32x4 *>(patch);                                    // [8 waves][4 quads][64 lanes] partial sums handed to the other half: a lane's 16-byte pieces are 1 KB apart, a wave's contiguous
    __bf16 *ob = reinterpret_cast<__bf16 *>(patch + 4 * 64 * 2 * sizeof(f32x16));     // [8 rows][32 cols][CS] output staging
    int c3_it = 0;
    for (;;) {
        C3_PROBE(0);
        const int c_n = n, c_ty0 = ty0, c_tx0 = tx0;
        const int next = tile + tstep;
        const bool has_next = next < tend;
        if (has_next) { decode(next); gload(); }                 // in flight during this tile's MFMAs (no other global load until lstore)

        C3_PROBE(1);
        f32x16 acc[2];
        acc[0] = (f32x16)(0.f); acc[1] = (f32x16)(0.f);
#pragma unroll
        for (int tap = 0; tap < ((C3_DBG & 2) ? 1 : 9); tap++) {
            const int dy = tap / 3, dx = tap % 3;
            const unsigned char *b0 = patch + ((2 * rw + dy) * (C3_TW + 2) + r + dx) * STRIDE + 16 * h + half * HC * 32;
            const unsigned char *b1 = b0 + (C3_TW + 2) * STRIDE;
#pragma unroll
            for (int cs = 0; cs < HC; cs++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[tap][cs], *reinterpret_cast<const bf16x8 *>(b0 + cs * 32), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[tap][cs], *reinterpret_cast<const bf16x8 *>(b1 + cs * 32), acc[1], 0, 0, 0);
            }
        }
        C3_PROBE(2);
        __syncthreads();                                         // the patch is consumed: its LDS now carries partial sums and outputs
        C3_PROBE(3);
        // both halves finish one output row each: a wave hands the other half its partial sums of the row it does NOT finish (four 16-byte
        // pieces per lane, lane-contiguous: conflict-free) and adds what it receives to the row it keeps (lower half row 0, upper row 1)
        // (the two halves are separate straight-line copies: indexing acc[] with the run-time half sends the accumulators to scratch)
        auto give = [&](const f32x16 &a) {
#pragma unroll
            for (int g = 0; g < 4; g++) part[((wave * 4 + g) << 6) + lane] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        };
        auto finish = [&](const f32x16 &a, int j) {
            f32x4 o2[4];
#pragma unroll
            for (int g = 0; g < 4; g++) o2[g] = part[(((wave ^ 4) * 4 + g) << 6) + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 o;                                        // staging rows are 32 channels wide: no channel test, no branch between the LDS reads
#pragma unroll
                for (int q = 0; q < 4; q++) o[q] = (__bf16)actc<ACT>((a[4 * g + q] + o2[g][q]) + biasr[g][q]);
                *reinterpret_cast<bf16x4 *>(ob + ((2 * rw + j) * C3_TW + r) * 32 + 8 * g + 4 * h) = o;
            }
        };
        if (half) give(acc[0]); else give(acc[1]);
        __syncthreads();
        C3_PROBE(4);
        if (half) finish(acc[1], 1); else finish(acc[0], 0);
        __syncthreads();
        C3_PROBE(5);
#pragma unroll
        for (int w = 0; w < NWB; w++) {
            const int oy = c_ty0 + (wb_yx[w] >> 16), ox = c_tx0 + (wb_yx[w] & 0xffff);
            if (wb_src[w] < 0 || oy >= p.H || ox >= p.W) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ob + wb_src[w]);
            if ((C3_DBG & 4) && v[0] != (__bf16)123.f) continue;
            __bf16 *dst = p.y + (((long)c_n * p.H * U + oy * U) * ((long)p.W * U) + ox * U) * p.out_ldc + p.out_coff + wb_part[w];
            for (int uy = 0; uy < U; uy++)
                for (int ux = 0; ux < U; ux++) *reinterpret_cast<bf16x8 *>(dst + ((long)uy * p.W * U + ux) * p.out_ldc) = v;
        }
        C3_PROBE(6);
        if (!has_next) break;
        __syncthreads();                                         // output staging read: the LDS takes the next patch
        C3_PROBE(7);
        lstore();
        C3_PROBE(8);
        __syncthreads();
        tile = next;
        c3_it++;
    }
}

// ---- FPN lateral INSIDE the smoothing conv (round 4).  The largest FPN level of the bf16 detector ran as two launches: the lateral
// `in2` (1x1, 16 -> 96, + ReLU, + nearest-x2 top-down add: 362 MB written at 184 x 320 x 32 images, 223 us) and the smoothing conv
// `out2` (3x3, 96 -> 24: reads those 362 MB back, 190 us).  The lateral's output has no other consumer, so this kernel never writes it:
// a workgroup stages the 16-channel patch of c2 (10 x 34 pixels, 11 KB) and the matching 6 x 18 pixels of the top-down tensor in LDS,
// computes the 96 lateral channels of its patch with one bf16 MFMA per 32 pixels x 32 channels (K = 16) -- bias, ReLU, top-down add in
// fp32, ONE rounding to bf16: the same arithmetic in the same order as pw_bf16_kernel, so the patch holds bit for bit what the 3x3 kernel
// used to read -- writes them where conv3x3_bf16_pers8_kernel's patch lives, and runs that kernel's MFMA phase, exchange and write-back
// unchanged.  Prefetch per tile: 5 sixteen-byte pieces per thread instead of 8.
struct C3LatArgs {
    C3Args c;                       // the smoothing conv (x unused; Cin = 96)
    const __bf16 *x2, *wl, *td;     // lateral input [N,H,W,16], lateral weights [96][16], top-down tensor [N,H/2,W/2,td_ldc]
    const float *bl;                // lateral bias f32[96]
    int td_ldc;
};

#ifndef C3L_DBG
#define C3L_DBG 0          // timing experiments only: 1 no lateral phase, 2 no 3x3 MFMAs, 4 no exchange of the channel halves, 8 no global stores, 16 no prefetch loads
#endif
template <int ACT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_lat_bf16_kernel(C3LatArgs q, int tiles_x, int tiles_per_img, int total) {
    constexpr int NCS = 6, TH = 8, HC = NCS / 2;
    constexpr int NCH = 2 * NCS, STRIDE = 16 * NCH + 16;         // 96 channels: 12 pieces of 16 bytes + 16 bytes of padding per patch pixel
    constexpr int PW = C3_TW + 2, PH = TH + 2, NPX = PH * PW;    // 34 x 10 = 340 patch pixels
    constexpr int TDW = C3_TW / 2 + 2, TDH = TH / 2 + 2, NTD = TDH * TDW;   // 18 x 6 = 108 top-down pixels
    constexpr int TDS = 16 * NCH + 16;                           // top-down pixel stride in LDS (same padding rule)
    constexpr int N2 = (NPX * 2 + 511) / 512, NT3 = (NTD * NCH + 511) / 512;     // prefetch pieces per thread: 2 + 3
    const C3Args &p = q.c;
    __shared__ __attribute__((aligned(16))) unsigned char patch[NPX * STRIDE];
    __shared__ __attribute__((aligned(16))) unsigned char c2l[NPX * 32];
    __shared__ __attribute__((aligned(16))) unsigned char tdl[NTD * TDS];
    __shared__ __attribute__((aligned(16))) float s_bl[96];
    static_assert(sizeof(patch) >= 4 * 64 * 32 * 4 + TH * C3_TW * 32 * 2, "patch buffer doubles as partial-sum + output staging");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rw = wave & 3, half = wave >> 2;
    if (threadIdx.x < 96) s_bl[threadIdx.x] = q.bl[threadIdx.x];
    int n, ty0, tx0;
    auto decode = [&](int t) {
        n = t / tiles_per_img;
        const int rem = t - n * tiles_per_img;
        ty0 = (rem / tiles_x) * TH; tx0 = (rem % tiles_x) * C3_TW;
    };
    // tile-invariant piece coordinates, once per thread
    int c2_py[N2], c2_px[N2], c2_lds[N2];
#pragma unroll
    for (int k = 0; k < N2; k++) {
        const int i = threadIdx.x + 512 * k, pix = i >> 1;
        c2_py[k] = pix / PW; c2_px[k] = i < NPX * 2 ? pix - (pix / PW) * PW : (1 << 20);
        c2_lds[k] = pix * 32 + (i & 1) * 16;
    }
    int td_py[NT3], td_px[NT3], td_ch[NT3], td_lds[NT3];
#pragma unroll
    for (int k = 0; k < NT3; k++) {
        const int i = threadIdx.x + 512 * k, pix = i / NCH;
        td_ch[k] = i - pix * NCH;
        td_py[k] = pix / TDW; td_px[k] = i < NTD * NCH ? pix - (pix / TDW) * TDW : (1 << 20);
        td_lds[k] = pix * TDS + td_ch[k] * 16;
    }
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    bf16x8 pv2[N2], pvt[NT3];
    auto gload = [&]() {
        const __bf16 *x2 = q.x2 + (long)n * p.H * p.W * 16;
        const __bf16 *td = q.td + (long)n * H2 * W2 * q.td_ldc;
#pragma unroll
        for (int k = 0; k < N2; k++) {
            const int iy = ty0 - 1 + c2_py[k], ix = tx0 - 1 + c2_px[k];
            pv2[k] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? *reinterpret_cast<const bf16x8 *>(x2 + ((long)iy * p.W + ix) * 16 + (threadIdx.x & 1) * 8) : zero8();
        }
#pragma unroll
        for (int k = 0; k < NT3; k++) {
            const int sy = (ty0 >> 1) - 1 + td_py[k], sx = (tx0 >> 1) - 1 + td_px[k];
            pvt[k] = ((unsigned)sy < (unsigned)H2 && (unsigned)sx < (unsigned)W2) ? *reinterpret_cast<const bf16x8 *>(td + ((long)sy * W2 + sx) * q.td_ldc + td_ch[k] * 8) : zero8();
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int k = 0; k < N2; k++) if (c2_px[k] < (1 << 20)) *reinterpret_cast<bf16x8 *>(c2l + c2_lds[k]) = pv2[k];
#pragma unroll
        for (int k = 0; k < NT3; k++) if (td_px[k] < (1 << 20)) *reinterpret_cast<bf16x8 *>(tdl + td_lds[k]) = pvt[k];
    };
    // the smoothing conv's weight fragments (as conv3x3_bf16_pers8_kernel) and the lateral's: rows 32 cb + r, k half h
    const __bf16 *wrow = p.w + (long)r * 9 * p.Cin + 8 * h + half * HC * 16;
    bf16x8 aw[9][HC];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int cs = 0; cs < HC; cs++) aw[tap][cs] = *reinterpret_cast<const bf16x8 *>(wrow + tap * p.Cin + cs * 16);
    bf16x8 al[3];
#pragma unroll
    for (int cb = 0; cb < 3; cb++) al[cb] = *reinterpret_cast<const bf16x8 *>(q.wl + (long)(32 * cb + r) * 16 + 8 * h);
    const int U = p.out_up, CS = p.cstore, PCS = CS >> 3;
    f32x4 biasr[4];
#pragma unroll
    for (int g = 0; g < 4; g++) biasr[g] = 8 * g + 4 * h < CS ? *reinterpret_cast<const f32x4 *>(p.bias + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NWB = 2;
    int wb_src[NWB], wb_yx[NWB], wb_part[NWB];
#pragma unroll
    for (int w = 0; w < NWB; w++) {
        const int i = threadIdx.x + 512 * w;
        const int px = i / PCS, part_i = i - px * PCS;
        wb_src[w] = i < TH * C3_TW * PCS ? px * 32 + part_i * 8 : -1;
        wb_yx[w] = ((px / C3_TW) << 16) | (px % C3_TW);
        wb_part[w] = part_i * 8;
    }
    int tile, tstep, tend;
    if ((gridDim.x & 7) == 0) {
        const int j = blockIdx.x & 7;
        tile = (int)((long)j * total / 8) + (int)(blockIdx.x >> 3);
        tstep = (int)(gridDim.x >> 3);
        tend = (int)((long)(j + 1) * total / 8);
    } else { tile = blockIdx.x; tstep = gridDim.x; tend = total; }
    if (tile >= tend) return;
    decode(tile);
    gload();
    lstore();
    __syncthreads();
    f32x4 *part = reinterpret_cast<f32x4 *>(patch);
    __bf16 *ob = reinterpret_cast<__bf16 *>(patch + 4 * 64 * 2 * sizeof(f32x16));
    for (;;) {
        const int c_n = n, c_ty0 = ty0, c_tx0 = tx0;
        // ---- the lateral: 11 groups of 32 patch pixels x 3 blocks of 32 channels, one (group, block) unit per wave and round
        if (!(C3L_DBG & 1))
        for (int u = wave; u < 11 * 3; u += 8) {
            const int g = u / 3, cb = u - g * 3;
            const int pix = 32 * g + r;                          // this lane's patch pixel (MFMA column)
            const int pixc = pix < NPX ? pix : NPX - 1;
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(c2l + pixc * 32 + 16 * h);
            f32x16 la = (f32x16)(0.f);
            la = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cb == 0 ? al[0] : cb == 1 ? al[1] : al[2], b, la, 0, 0, 0);
            const int py = pixc / PW, px = pixc - py * PW;
            const int iy = c_ty0 - 1 + py, ix = c_tx0 - 1 + px;
            const bool inside = pix < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // top-down pixel of (iy, ix): (iy >> 1, ix >> 1), relative to the staged region's origin ((ty0 >> 1) - 1, (tx0 >> 1) - 1)
            const int tpy = ((c_ty0 + py + 1) >> 1) - (c_ty0 >> 1), tpx = ((c_tx0 + px + 1) >> 1) - (c_tx0 >> 1);     // = (iy >> 1) - origin, for iy >= -1
            const unsigned char *tdp = tdl + (tpy * TDW + tpx) * TDS + (32 * cb + 4 * h) * 2;
            // Round 6 (knock-outs: this phase was 86 of the launch's 197 us): straight-line code -- the eight LDS operands of the unit are
            // requested together (clamped pixel: every lane reads), bias + ReLU + top-down add on float pairs, ONE conversion instruction per
            // pair, the padding pixels cleared by a select on the packed result, only the store predicated.  The compiler had turned the
            // per-element `inside ? ... : 0` into seventeen exec-mask branches per unit, each behind its own LDS wait.
#pragma unroll
            for (int qh = 0; qh < 2; qh++) {                     // (two channel quads at a time: all four in flight spilled three registers)
                f32x4 bias[2];
                bf16x4 t4[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    bias[k] = *reinterpret_cast<const f32x4 *>(s_bl + 32 * cb + 8 * (2 * qh + k) + 4 * h);
                    t4[k] = *reinterpret_cast<const bf16x4 *>(tdp + 16 * (2 * qh + k));
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int qd = 2 * qh + k;
                    f32x4 v = f32x4{la[4 * qd], la[4 * qd + 1], la[4 * qd + 2], la[4 * qd + 3]} + bias[k];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.f);
                    v += __builtin_convertvector(t4[k], f32x4);
                    const bf16x4 o = __builtin_convertvector(v, bf16x4);
                    u32x2 ow = __builtin_bit_cast(u32x2, o);
                    ow[0] = inside ? ow[0] : 0u; ow[1] = inside ? ow[1] : 0u;
                    if (pix < NPX) *reinterpret_cast<u32x2 *>(patch + pix * STRIDE + (32 * cb + 8 * qd + 4 * h) * 2) = ow;
                }
            }
        }
        __syncthreads();
        const int next = tile + tstep;
        const bool has_next = next < tend;
        if (has_next) { decode(next); if (!(C3L_DBG & 16)) gload(); }                 // in flight during this tile's MFMAs

        f32x16 acc[2];
        acc[0] = (f32x16)(0.f); acc[1] = (f32x16)(0.f);
        if (!(C3L_DBG & 2))
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3, dx = tap % 3;
            const unsigned char *b0 = patch + ((2 * rw + dy) * PW + r + dx) * STRIDE + 16 * h + half * HC * 32;
            const unsigned char *b1 = b0 + PW * STRIDE;
#pragma unroll
            for (int cs = 0; cs < HC; cs++) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[tap][cs], *reinterpret_cast<const bf16x8 *>(b0 + cs * 32), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[tap][cs], *reinterpret_cast<const bf16x8 *>(b1 + cs * 32), acc[1], 0, 0, 0);
            }
        }
        __syncthreads();                                         // the patch is consumed: its LDS now carries partial sums and outputs
        auto give = [&](const f32x16 &a) {
#pragma unroll
            for (int g = 0; g < 4; g++) part[((wave * 4 + g) << 6) + lane] = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        };
        auto finish = [&](const f32x16 &a, int j) {
            f32x4 o2[4];
#pragma unroll
            for (int g = 0; g < 4; g++) o2[g] = part[(((wave ^ 4) * 4 + g) << 6) + lane];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 o;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) o[qq] = (__bf16)actc<ACT>((a[4 * g + qq] + o2[g][qq]) + biasr[g][qq]);
                *reinterpret_cast<bf16x4 *>(ob + ((2 * rw + j) * C3_TW + r) * 32 + 8 * g + 4 * h) = o;
            }
        };
        if (!(C3L_DBG & 4)) { if (half) give(acc[0]); else give(acc[1]); }
        __syncthreads();
        if (!(C3L_DBG & 4)) { if (half) finish(acc[1], 1); else finish(acc[0], 0); }
        else if (acc[0][0] == 123.f) ob[threadIdx.x] = (__bf16)acc[1][3];
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWB; w++) {
            const int oy = c_ty0 + (wb_yx[w] >> 16), ox = c_tx0 + (wb_yx[w] & 0xffff);
            if (wb_src[w] < 0 || oy >= p.H || ox >= p.W) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(ob + wb_src[w]);
            if ((C3L_DBG & 8) && v[0] != (__bf16)123.f) continue;
            __bf16 *dst = p.y + (((long)c_n * p.H * U + oy * U) * ((long)p.W * U) + ox * U) * p.out_ldc + p.out_coff + wb_part[w];
            for (int uy = 0; uy < U; uy++)
                for (int ux = 0; ux < U; ux++) *reinterpret_cast<bf16x8 *>(dst + ((long)uy * p.W * U + ux) * p.out_ldc) = v;
        }
        if (!has_next) break;
        __syncthreads();                                         // output staging read: the LDS takes the next tile's inputs
        lstore();
        __syncthreads();
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------- depthwise conv (+ SE pooling)
// block = (C/8 channel groups) x (L = 256 / (C/8) pixel lanes) over a chunk of output pixels of one image; a thread
// computes 8 channels of its pixels (16-byte loads and stores, fp32 arithmetic).  With `partial` set the kernel also leaves the
// per-channel sum of its chunk (of the ACTIVATED fp32 values, combined over the pixel lanes in a fixed order) for the
// Squeeze-Excitation average pool, so that pool never reads the tensor again.
// pixels per block: about 24 blocks per image whatever the map size (each block first stages the layer's weights in LDS: fewer,
// longer blocks amortise that; 64 per image cost the 5x5 layers on the small maps a third of their time) (a 23 x 40 map in 1024-pixel chunks would put ONE block per
// image on a 256-CU chip, each thread walking hundreds of pixels).  A function of the map size only, never of the batch: the
// chunking fixes the summation order of the SE pool, and an image's result must not depend on the batch it travels in.
static inline int dw_chunk(int N, int HWo) {
    (void)N;
    long c = (HWo + 23) / 24;
    c = (c + 15) / 16 * 16;
    return (int)(c < 16 ? 16 : (c > 1024 ? 1024 : c));
}

struct DwArgs {
    const __bf16 *x;
    const float *w, *bias;          // w f32[k*k][C] tap-major
    __bf16 *y;
    float *partial;                 // f32[N][nblk][C] or null
    int H, W, C, k, stride, Ho, Wo, act, nblk, chunk;
    int ob;                         // channel octets per block
    int xcd;                        // 1: XCD-aware workgroup order
};

// K, S compile-time (taps unrolled); a thread walks RUNS of 4 consecutive outputs of one row: the (3 S + K) input columns of a run are
// loaded once per kernel row and feed all four outputs (5x5 / s1: 40 loads per 4 outputs instead of 100), the weights and the bias
// sit in LDS (the first version re-read them from global memory per tap: 75 loads per output pixel made the layer issue-bound at
// 7x its HBM time).  A block owns whole output rows (p.chunk of them).
#ifndef STEM_DBG
#define STEM_DBG 0         // timing experiments only: 1 every block reads image row 0, 2 no loads, 4 no stores
#endif
#ifndef STEM_ST_T
#define STEM_ST_T 1        // 1: the stem's stores transposed through LDS (lane-contiguous kilobytes)
#endif
#ifndef STEM_W_LDS
#define STEM_W_LDS 0
#endif
#ifndef DW_R51
#define DW_R51 4         // outputs per run: 5x5 stride 1, 5x5 stride 2, 3x3 stride 1, 3x3 stride 2
#endif
#ifndef DW_R52
#define DW_R52 4
#endif
#ifndef DW_R31
#define DW_R31 8
#endif
#ifndef DW_R32
#define DW_R32 8
#endif
#ifndef DW_DBG
#define DW_DBG 0            // timing experiments only: 1 no weight staging, 2 no global loads, 4 no stores
#endif
template <int K, int S, int ACT>
__global__ __launch_bounds__(256) void dwconv_bf16_kernel(DwArgs p) {
    constexpr int R = (S == 1 && K == 5) ? DW_R51 : (S == 2 && K == 5) ? DW_R52 : (S == 1) ? DW_R31 : DW_R32, NC = (R - 1) * S + K, PAD = (K - 1) / 2;       // 5x5 stride 1: runs of 8 outputs (12 loads per row for 8 outputs instead of 16; 3x3 loses with 8)
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    // A block owns p.ob channel octets (blockIdx.z picks the group): its slice of the weights is a few KB of LDS instead of (K*K+1)*C
    // floats, so that occupancy is set by registers -- the loads of a run are K dependent round trips and need many waves to hide.
    const int OB = p.ob, CB = OB * 8;
    const int C8 = p.C >> 3;
    float *wl = dw_lds;                                         // [K*K][CB] tap-major, then [CB] bias
    // XCD-aware order: consecutive workgroup ids go round the 8 XCDs, and the row chunks of one image share halo rows -- with the plain
    // (x, y, z) order every XCD's L2 fetched every image.  Virtual id v = (id % 8) * (B / 8) + id / 8 puts whole images on one XCD.
    int n = blockIdx.y, blk = blockIdx.x, zg = blockIdx.z;
    if (p.xcd) {
        const int B = gridDim.x * gridDim.y * gridDim.z, per = B >> 3;
        const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int v = id < per * 8 ? (id & 7) * per + (id >> 3) : id;      // the B % 8 tail maps to itself
        blk = v % gridDim.x;
        const int t = v / gridDim.x;
        zg = t % gridDim.z;
        n = t / gridDim.z;
    }
    const int cg0 = zg * OB;
    float (*red)[8] = reinterpret_cast<float (*)[8]>(dw_lds + (K * K + 1) * CB);
#if !(DW_DBG & 1)
    for (int i = threadIdx.x; i < (K * K + 1) * OB * 2; i += 256) {
        const int tap = i / (OB * 2), piece = i - tap * (OB * 2);
        if (cg0 * 2 + piece < C8 * 2)
            reinterpret_cast<f32x4 *>(wl)[i] = tap < K * K ? reinterpret_cast<const f32x4 *>(p.w)[tap * (C8 * 2) + cg0 * 2 + piece]
                                                           : reinterpret_cast<const f32x4 *>(p.bias)[cg0 * 2 + piece];
    }
#endif
    __syncthreads();
    const int L = 256 / OB;
    const int ql = threadIdx.x % OB, pl = threadIdx.x / OB;
    const int q = cg0 + ql;                                     // global channel octet
    const int runs_x = (p.Wo + R - 1) / R;
    const int r0 = blk * p.chunk, r1 = min(r0 + p.chunk, p.Ho);
    const int nruns = (r1 - r0) * runs_x;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sum[j] = 0.f;
    if (pl < L && q < C8) {
        const float *bl = wl + K * K * CB + ql * 8;
        for (int run = pl; run < nruns; run += L) {
            const int ry = run / runs_x;
            const int oy = r0 + ry, ox0 = (run - ry * runs_x) * R;
            // The layer is VALU-bound, not memory-bound (5x5: 25 taps x 8 channels per output and thread): the multiply-adds run as
            // packed fp32 FMAs on channel pairs (v_pk_fma_f32, half the instructions of scalar FMAs, a quarter of the mul + add pairs
            // the file-wide -ffp-contract=off would give); a bf16 pair becomes two floats with one shift and one mask.
            f32x2 acc[R][4];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[r][j] = f32x2{bl[2 * j], bl[2 * j + 1]};
#pragma unroll
            for (int a = 0; a < K; a++) {
                const int iy = oy * S - PAD + a;
                if ((unsigned)iy >= (unsigned)p.H) continue;
                f32x2 wr[K][4];
#pragma unroll
                for (int b = 0; b < K; b++) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wl + (a * K + b) * CB + ql * 8), w1 = *reinterpret_cast<const f32x4 *>(wl + (a * K + b) * CB + ql * 8 + 4);
                    wr[b][0] = f32x2{w0[0], w0[1]}; wr[b][1] = f32x2{w0[2], w0[3]}; wr[b][2] = f32x2{w1[0], w1[1]}; wr[b][3] = f32x2{w1[2], w1[3]};
                }
                // All the columns of the row are loaded before the first is used (clamped addresses, no branch: with a bounds check and
                // `continue` per column every load sat in its own basic block and was waited for alone -- NC dependent round trips per
                // row); a column outside the image is then zeroed, which adds +0 products where the first version skipped the taps.
                const __bf16 *row = p.x + (((long)n * p.H + iy) * p.W) * p.C + q * 8;
                const int off0 = (ox0 * S - PAD) * p.C, offmax = (p.W - 1) * p.C;
                u32x4 dd[NC];
#pragma unroll
                for (int ci = 0; ci < NC; ci++) {
#if DW_DBG & 2
                    dd[ci] = u32x4{(unsigned)ci, (unsigned)iy, 0u, 0u};
#else
                    dd[ci] = *reinterpret_cast<const u32x4 *>(row + min(max(off0 + ci * p.C, 0), offmax));
#endif
                }
#pragma unroll
                for (int ci = 0; ci < NC; ci++) {
                    const int ix = ox0 * S - PAD + ci;
                    const u32x4 d = (unsigned)ix < (unsigned)p.W ? dd[ci] : u32x4{0u, 0u, 0u, 0u};
                    f32x2 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = f32x2{__builtin_bit_cast(float, d[j] << 16), __builtin_bit_cast(float, d[j] & 0xffff0000u)};
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int b = ci - r * S;                 // compile-time: the tap this column is for output r of the run
                        if (b >= 0 && b < K) {
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[r][j] = __builtin_elementwise_fma(v[j], wr[b][j], acc[r][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (ox0 + r >= p.Wo) continue;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; j++) { const float t = actc<ACT>(acc[r][j >> 1][j & 1]); sum[j] += t; o[j] = (__bf16)t; }
                if (!(DW_DBG & 4) || o[0] == (__bf16)123.f) *reinterpret_cast<bf16x8 *>(p.y + (((long)n * p.Ho + oy) * p.Wo + ox0 + r) * p.C + q * 8) = o;
            }
        }
    }
    if (!p.partial) return;
#pragma unroll
    for (int j = 0; j < 8; j++) red[threadIdx.x][j] = sum[j];
    __syncthreads();
    if (pl == 0 && q < C8) {
        for (int l = 1; l < L; l++)
#pragma unroll
            for (int j = 0; j < 8; j++) sum[j] += red[l * OB + ql][j];
        float *dst = p.partial + ((long)n * p.nblk + blk) * p.C + q * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) dst[j] = sum[j];
    }
}

// ---- expansion INSIDE the depthwise conv (round 4).  The second block of MobileNetV3-small expands 16 -> 72 channels at 184 x 320 (a
// 1x1 conv: 301 MB written per 32 images) only for the depthwise 3x3 / stride 2 that follows to read them back (the largest round trip of
// the backbone: 96 + 134 us).  Here a workgroup takes a tile of 4 x 16 outputs: it stages the 9 x 33-pixel input patch (16 channels,
// 9.5 KB), expands it with one bf16 MFMA per 32 pixels x 32 channels (bias, activation, ONE rounding to bf16 -- what pw_bf16_kernel
// would have stored, bit for bit; pixels outside the image stay zero: the depthwise conv pads the EXPANDED tensor) into 51 KB of LDS and
// runs dwconv_bf16_kernel's arithmetic (same taps in the same order, packed fp32 FMAs) from there.  The expanded tensor is never written.
struct ExDwArgs {
    const __bf16 *x, *we;           // input [N,H,W,16], expansion weights bf16[96][16] (rows beyond the layer's channels are zero)
    const float *be, *wd, *bd;      // expansion bias f32[96]; depthwise weights f32[9][C] tap-major, bias f32[C]
    __bf16 *y;                      // [N,Ho,Wo,C]
    int H, W, Ho, Wo, C;            // C = padded expanded channels (multiple of 16, <= 96)
};

constexpr int XD_THREADS = 320;     // five waves: ten 32-pixel groups of the patch (two per wave) and, at 80 channels, 32 runs x 10 channel octets (one per thread)

template <int C, int ACT_E, int ACT_D>
__global__ __launch_bounds__(XD_THREADS) void expand_dw3x3s2_bf16_kernel(ExDwArgs p, int tiles_x, int tiles_y) {
    constexpr int TOH = 4, TOW = 16, PH = 2 * TOH + 1, PW = 2 * TOW + 1, NPX = PH * PW;      // 9 x 33 = 297 input pixels
    constexpr int NT = XD_THREADS, NW = NT / 64;
    __shared__ __attribute__((aligned(16))) unsigned char xin[NPX * 32];
    __shared__ __attribute__((aligned(16))) float wl[10 * C];  // depthwise [9 taps][C] + bias [C]
    extern __shared__ __attribute__((aligned(16))) unsigned char ex[];      // [NPX][EXS]: the expanded patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.x, n = blockIdx.y;
    const int oy0 = (t / tiles_x) * TOH, ox0 = (t % tiles_x) * TOW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    constexpr int C8 = C >> 3;
    constexpr int EXS = 2 * C + 16;                                 // bytes per expanded pixel: an odd number of 16-byte chunks (C % 16 == 0)
    const bool tile_inside = iy0 >= 0 && ix0 >= 0 && iy0 + PH <= p.H && ix0 + PW <= p.W;      // block-uniform
    // phase 1: the input patch (zeros outside the image), all loads in flight at once, and the depthwise table
    const __bf16 *ximg = p.x + (long)n * p.H * p.W * 16;
    {
        constexpr int NL = (NPX * 2 + NT - 1) / NT, NWL = (10 * C + NT - 1) / NT;
        bf16x8 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = tid + NT * k, pix = i >> 1, py = pix / PW, px = pix - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            v[k] = zero8();
            if (i < NPX * 2 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v[k] = *reinterpret_cast<const bf16x8 *>(ximg + ((long)iy * p.W + ix) * 16 + (i & 1) * 8);
        }
        float w4[NWL];
#pragma unroll
        for (int k = 0; k < NWL; k++) {
            const int i = tid + NT * k;
            w4[k] = i < 9 * C ? p.wd[i] : i < 10 * C ? p.bd[i - 9 * C] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = tid + NT * k;
            if (i < NPX * 2) *reinterpret_cast<bf16x8 *>(xin + (i >> 1) * 32 + (i & 1) * 16) = v[k];
        }
#pragma unroll
        for (int k = 0; k < NWL; k++) {
            const int i = tid + NT * k;
            if (i < 10 * C) wl[i] = w4[k];
        }
    }
    // the expansion's weights and bias stay in registers: the kernel runs two workgroups per CU (LDS), registers are free
    constexpr int ncb = (C + 31) >> 5;
    bf16x8 ae[ncb];
    f32x4 be[ncb][4];
#pragma unroll
    for (int cb = 0; cb < ncb; cb++) {
        ae[cb] = *reinterpret_cast<const bf16x8 *>(p.we + (long)(32 * cb + r) * 16 + 8 * h);
#pragma unroll
        for (int qd = 0; qd < 4; qd++) be[cb][qd] = *reinterpret_cast<const f32x4 *>(p.be + 32 * cb + 8 * qd + 4 * h);
    }
    __syncthreads();
    // phase 2: the expansion, 32 pixels per wave and round: the pixel operand read once, the (up to) three MFMAs back to back
    constexpr int NG = (NPX + 31) / 32;                          // 10 pixel groups
    auto expand = [&](auto ins_tag) {
        constexpr bool INS = decltype(ins_tag)::value;          // every pixel of the patch inside the image: no selects
        for (int g = wave; g < NG; g += NW) {
            const int pix = 32 * g + r, pixc = pix < NPX ? pix : NPX - 1;
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(xin + pixc * 32 + 16 * h);
            bool inside = true;
            if (!INS) {
                const int py = pixc / PW, px = pixc - py * PW;
                inside = (unsigned)(iy0 + py) < (unsigned)p.H && (unsigned)(ix0 + px) < (unsigned)p.W;
            }
            f32x16 acc[ncb];
#pragma unroll
            for (int cb = 0; cb < ncb; cb++) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae[cb], b, (f32x16)(0.f), 0, 0, 0);
            unsigned char *dst = ex + pix * EXS + 8 * h;
            if (pix < NPX) {
#pragma unroll
                for (int cb = 0; cb < ncb; cb++) {
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) {
                        if (32 * cb + 8 * qd < C) {             // C is a multiple of 16: both halves of the wave in or out
                            const f32x2 s0 = f32x2{acc[cb][4 * qd], acc[cb][4 * qd + 1]} + f32x2{be[cb][qd][0], be[cb][qd][1]};
                            const f32x2 s1 = f32x2{acc[cb][4 * qd + 2], acc[cb][4 * qd + 3]} + f32x2{be[cb][qd][2], be[cb][qd][3]};
                            const f32x4 v = {actc<ACT_E>(s0[0]), actc<ACT_E>(s0[1]), actc<ACT_E>(s1[0]), actc<ACT_E>(s1[1])};
                            bf16x4 o = __builtin_convertvector(v, bf16x4);
                            if (!INS && !inside) o = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                            *reinterpret_cast<bf16x4 *>(dst + (32 * cb + 8 * qd) * 2) = o;
                        }
                    }
                }
            }
        }
    };
    if (tile_inside) expand(std::true_type{});
    else expand(std::false_type{});
    __syncthreads();
    // phase 3: the depthwise 3x3 / stride 2 on the expanded patch: runs of two outputs per thread and channel octet, every output's taps
    // in dwconv_bf16_kernel<3, 2>'s order (rows, then columns)
    constexpr int R = 2, K = 3, S = 2, NC = (R - 1) * S + K, RPR = TOW / R, NRUN = TOH * RPR;
    for (int task = tid; task < NRUN * C8; task += NT) {
        const int pl = task / C8, ql = task - pl * C8;              // run, channel octet
        const int ry = pl / RPR, rx0 = (pl - ry * RPR) * R;          // run's output row and first output column inside the tile
        const int oy = oy0 + ry;
        if (oy >= p.Ho) continue;
        const float *bl = wl + 9 * C + ql * 8;
        f32x2 acc[R][4];
#pragma unroll
        for (int rr = 0; rr < R; rr++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[rr][j] = f32x2{bl[2 * j], bl[2 * j + 1]};
#pragma unroll
        for (int a = 0; a < K; a++) {
            const int iy = oy * S - 1 + a;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            f32x2 wr[K][4];
#pragma unroll
            for (int b = 0; b < K; b++) {
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wl + (a * K + b) * C + ql * 8), w1 = *reinterpret_cast<const f32x4 *>(wl + (a * K + b) * C + ql * 8 + 4);
                wr[b][0] = f32x2{w0[0], w0[1]}; wr[b][1] = f32x2{w0[2], w0[3]}; wr[b][2] = f32x2{w1[0], w1[1]}; wr[b][3] = f32x2{w1[2], w1[3]};
            }
            const unsigned char *row = ex + ((ry * S + a) * PW) * EXS + ql * 16;
#pragma unroll
            for (int ci = 0; ci < NC; ci++) {                   // a column outside the image is zero in the patch: dwconv_bf16_kernel adds the same +0 products
                const u32x4 d = *reinterpret_cast<const u32x4 *>(row + (rx0 * S + ci) * EXS);
                f32x2 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = f32x2{__builtin_bit_cast(float, d[j] << 16), __builtin_bit_cast(float, d[j] & 0xffff0000u)};
#pragma unroll
                for (int rr = 0; rr < R; rr++) {
                    const int b = ci - rr * S;
                    if (b >= 0 && b < K) {
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[rr][j] = __builtin_elementwise_fma(v[j], wr[b][j], acc[rr][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int ox = ox0 + rx0 + rr;
            if (ox >= p.Wo) continue;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (__bf16)actc<ACT_D>(acc[rr][j >> 1][j & 1]);
            *reinterpret_cast<bf16x8 *>(p.y + (((long)n * p.Ho + oy) * p.Wo + ox) * C + (long)ql * 8) = o;
        }
    }
}

template <int K, int S>
static int launch_dw(const DwArgs &p, int N, hipStream_t stream) {
    const size_t lds = sizeof(float) * ((size_t)(K * K + 1) * p.ob * 8 + 256 * 8);
    PT_CHECK(lds <= 64 * 1024, "ptocr_dwconv_bf16: %zu bytes of LDS for %d channel octets per block", lds, p.ob);
#define PT_DW(A) hipLaunchKernelGGL((dwconv_bf16_kernel<K, S, A>), dim3(p.nblk, N, cdiv(p.C / 8, p.ob)), dim3(256), lds, stream, p)
    PT_ACT_SWITCH(p.act, PT_DW);
#undef PT_DW
    return launch_ok("dwconv_bf16_kernel");
}

// channel octets per block: all of them up to 16, else the largest divisor in [6, 16] (no idle lanes), else 16 with a ragged last group.
// A function of C only (the SE pool's summation order follows the lane layout).
static inline int dw_octets(int C8) {
    static const int mx = getenv("PTOCR_DW_OB") ? atoi(getenv("PTOCR_DW_OB")) : 36;    // measured: 16 / 36 / 72 -> 3667 / 3611 / 3666 us per forward
    if (C8 <= mx) return C8;
    for (int d = mx; d >= 6; d--)
        if (C8 % d == 0) return d;
    return mx;
}

// output rows per block: the pixel budget of dw_chunk in whole rows
static inline int dw_rows(int N, int Ho, int Wo) {
    const int r = dw_chunk(N, Ho * Wo) / Wo;
    return r < 1 ? 1 : r;
}

// ---------------------------------------------------------------------------------------------- stem: 3x3 / stride 2 / pad 1, RGB
// reads the model's own input f32[N,3,H,W] (no layout pass), writes bf16[N,Ho,Wo,16] (+ folded BN + Hardswish): 27 taps x 16
// channels per pixel on the VALU, weights (f32[27][16], row (c*3 + ky)*3 + kx) and bias broadcast from LDS.
template <int ACT>
__global__ __launch_bounds__(256) void stem3x3s2_bf16_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                             __bf16 *__restrict__ y, int H, int W, int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) float sw[27 * 16 + 16];
    for (int i = threadIdx.x; i < 27 * 16 + 16; i += 256) sw[i] = i < 27 * 16 ? w[i] : bias[i - 27 * 16];
    __syncthreads();
    // grid = (column blocks, rows, images): no division per pixel (the flat 64-bit index of the first version cost two 64-bit divisions)
    const long n = blockIdx.z;
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= Wo) return;
    const long i = (n * Ho + oy) * Wo + ox;
    f32x2 acc[8];                                                // packed fp32 FMAs on channel pairs: the 432 multiply-adds per pixel bound this kernel
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = *reinterpret_cast<const f32x2 *>(sw + 27 * 16 + 2 * j);
    for (int c = 0; c < 3; c++)
        for (int ky = 0; ky < 3; ky++) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; kx++) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float v = x[((n * 3 + c) * H + iy) * (long)W + ix];
                const f32x2 vv = {v, v};
                const f32x2 *wr = reinterpret_cast<const f32x2 *>(sw + ((c * 3 + ky) * 3 + kx) * 16);
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = __builtin_elementwise_fma(vv, wr[j], acc[j]);
            }
        }
    bf16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; j++) { o0[j] = (__bf16)actc<ACT>(acc[j >> 1][j & 1]); o1[j] = (__bf16)actc<ACT>(acc[4 + (j >> 1)][j & 1]); }
    *reinterpret_cast<bf16x8 *>(y + i * 16) = o0;
    *reinterpret_cast<bf16x8 *>(y + i * 16 + 8) = o1;
}

// Two output pixels per thread (W % 4 == 0): the five input columns 4t-1 .. 4t+3 of a (channel, kernel row) are ONE aligned 16-byte load
// (lane-contiguous: a wave reads 1 KB of a row) plus the left neighbour, instead of six 4-byte loads two floats apart.
template <int ACT>
__global__ __launch_bounds__(512) void stem3x3s2_bf16_pair_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                                  __bf16 *__restrict__ y, int H, int W, int Ho, int Wo) {
#if STEM_W_LDS
    __shared__ __attribute__((aligned(16))) float sw[27 * 16 + 16];
    for (int i = threadIdx.x; i < 27 * 16 + 16; i += blockDim.x) sw[i] = i < 27 * 16 ? w[i] : bias[i - 27 * 16];
    __syncthreads();
    const float *wb = sw + 27 * 16;
#else
    // The weights are the same for every lane and every index below is a compile-time constant: read straight from the (read-only,
    // wave-uniform) global arrays they become scalar loads and SGPR operands of the packed FMAs -- from LDS each of the 216 weight
    // pairs was a broadcast ds_read per thread, as many LDS instructions as half the FMAs.
    const float *sw = w, *wb = bias;
#endif
    const long n = blockIdx.z;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;       // (an XCD-aware row order was tried: no change, 178 us)
    if (!STEM_ST_T && 2 * t >= Wo) return;                     // (with the transposed stores a thread past the row still takes part in its wave's exchange: its loads are clamped below)
    const int tl = min(t, (W >> 2) - 1);                          // load column of a thread past the row: the last one (its outputs are never stored)
    f32x2 accA[8], accB[8];
#pragma unroll
    for (int j = 0; j < 8; j++) accA[j] = accB[j] = *reinterpret_cast<const f32x2 *>(wb + 2 * j);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            // (round 6) no branch around a row outside the image: its loads go to the nearest row inside and their values are replaced by zeros
            // (the products then add +0: the sums are what they were).  With `continue` every (channel, kernel row) was a basic block of its
            // own -- load, wait, multiply, nine times over (160 us = 60 of arithmetic + 50 of loads + 58 of stores, none hidden); as
            // straight-line code the compiler keeps two rows' loads in flight under the multiplies: 133.6 -> 120.6 us.  (All eighteen loads
            // of a thread requested before the first multiply: 132.7 us -- five waves per SIMD instead of eight.)
            const int iy = 2 * oy - 1 + ky;
            const bool row_in = (unsigned)iy < (unsigned)H;         // block-uniform
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            const float *row = x + ((n * 3 + c) * H + ((STEM_DBG & 1) ? 0 : iyc)) * (long)W;
            f32x4 q = (STEM_DBG & 2) ? f32x4{(float)t, (float)iy, 1.f, 2.f} : *reinterpret_cast<const f32x4 *>(row + 4 * tl);
            float left = (STEM_DBG & 2) ? 3.f : row[4 * (tl > 0 ? tl : 1) - 1];
            if (tl == 0) left = 0.f;
            if (!row_in) { q = f32x4{0.f, 0.f, 0.f, 0.f}; left = 0.f; }
            const f32x2 *wr = reinterpret_cast<const f32x2 *>(sw + ((c * 3 + ky) * 3) * 16);
            const float va[3] = {left, q[0], q[1]}, vb[3] = {q[1], q[2], q[3]};
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const f32x2 a2 = {va[kx], va[kx]}, b2 = {vb[kx], vb[kx]};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    accA[j] = __builtin_elementwise_fma(a2, wr[kx * 8 + j], accA[j]);
                    accB[j] = __builtin_elementwise_fma(b2, wr[kx * 8 + j], accB[j]);
                }
            }
        }
    bf16x8 o[4];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        o[0][j] = (__bf16)actc<ACT>(accA[j >> 1][j & 1]); o[1][j] = (__bf16)actc<ACT>(accA[4 + (j >> 1)][j & 1]);
        o[2][j] = (__bf16)actc<ACT>(accB[j >> 1][j & 1]); o[3][j] = (__bf16)actc<ACT>(accB[4 + (j >> 1)][j & 1]);
    }
    if ((STEM_DBG & 4) && o[0][0] != (__bf16)123.f) return;
#if STEM_ST_T
    // A thread's two pixels are 64 contiguous bytes and a wave's 128 pixels 4 KB, but stored thread by thread every instruction touched all
    // 32 lines of the 4 KB with 16 bytes per lane pair (the no-store knock-out: 58 of the kernel's 160 us).  Through LDS the pieces are
    // transposed so that instruction k writes the k-th KILOBYTE of the wave, lane by lane contiguous.
    __shared__ __attribute__((aligned(16))) bf16x8 stg[8][4 * 64 + 4];       // per wave: piece 4 lane + k (padding: the transposed reads of a quarter wave on distinct banks)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 4; k++) stg[wv][4 * lane + k] = o[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int t_w0 = blockIdx.x * blockDim.x + wv * 64;                   // first thread of the wave
        bf16x8 *wdst = reinterpret_cast<bf16x8 *>(y + ((n * Ho + oy) * Wo + 2 * t_w0) * 16);
        const int npiece = min(4 * 64, (Wo - 2 * t_w0) * 2);                  // pieces of this wave inside the row (Wo even)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int pc = 64 * k + lane;
            if (pc < npiece) wdst[pc] = stg[wv][pc];
        }
    }
#else
    bf16x8 *dst = reinterpret_cast<bf16x8 *>(y + ((n * Ho + oy) * Wo + 2 * t) * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = o[k];
#endif
}

// ---------------------------------------------------------------------------------------------- DB head tail
// ConvTranspose2d(C, C, 2, 2) + BN + ReLU -> ConvTranspose2d(C, 1, 2, 2) + bias -> sigmoid (det_db_head.py:13-17) for the small head
// of this detector (C = 24, stored as 32): one thread per input pixel produces its 4 x 4 block of fp32 probabilities.
// w1 f32[4][C][C] ((a*2+b), ci, co; BN folded), b1 f32[C], w2 f32[4][C] ((a2*2+b2), co), b2.
constexpr int HT_C = 24;
__global__ __launch_bounds__(256) void head_tail_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                                                             const float *__restrict__ w2, float b2, float *__restrict__ maps, int H, int W,
                                                             int ldc, long total) {
    __shared__ float sw1[4 * HT_C * HT_C], sb1[HT_C], sw2[4 * HT_C];
    for (int i = threadIdx.x; i < 4 * HT_C * HT_C; i += 256) sw1[i] = w1[i];
    for (int i = threadIdx.x; i < HT_C; i += 256) sb1[i] = b1[i];
    for (int i = threadIdx.x; i < 4 * HT_C; i += 256) sw2[i] = w2[i];
    __syncthreads();
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % W);
    const long t = i / W;
    const int oy = (int)(t % H);
    const long n = t / H;
    float xin[HT_C];
#pragma unroll
    for (int c8 = 0; c8 < HT_C / 8; c8++) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + i * ldc + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) xin[c8 * 8 + j] = (float)v[j];
    }
    float *out = maps + ((n * 4 * H + 4 * oy) * (4L * W) + 4 * ox);
    for (int ab = 0; ab < 4; ab++) {                        // position (a, b) of the first transposed conv
        float o[4] = {b2, b2, b2, b2};
        for (int co = 0; co < HT_C; co++) {
            float mid = sb1[co];
            const float *wr = sw1 + (ab * HT_C) * HT_C + co;
#pragma unroll
            for (int ci = 0; ci < HT_C; ci++) mid += xin[ci] * wr[ci * HT_C];
            mid = fmaxf(mid, 0.f);
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] += mid * sw2[q * HT_C + co];
        }
        const int a = ab >> 1, b = ab & 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int a2 = q >> 1, bb2 = q & 1;
            out[(long)(2 * a + a2) * (4L * W) + 2 * b + bb2] = 1.f / (1.f + expf(-o[q]));
        }
    }
}

#ifndef HT_FAST_SIGMOID
#define HT_FAST_SIGMOID 1   // v_exp_f32 + v_rcp_f32 (about 2 ulp) instead of expf + a division: the sigmoids were a third of the kernel's instructions
#endif
__device__ __forceinline__ float ht_sigmoid(float v) {
#if HT_FAST_SIGMOID
    return __builtin_amdgcn_rcpf(1.f + __expf(-v));
#else
    return 1.f / (1.f + expf(-v));
#endif
}

// The same tail on the matrix pipe: GEMM 1 (24 -> 4 x 24, per input pixel) as v_mfma_f32_32x32x16_bf16 with A = weights (rows = mid
// channels, bf16, zero-padded 24 -> 32 in both directions; eight fragments per lane, loaded once), B = 32 pixels straight from global
// memory; the lane then holds 16 mid channels of ONE pixel per (a, b) position, and the second contraction (24 -> 4 outputs) is
// register-local FMAs against per-lane copies of w2 / b1 plus one lane^32 exchange.  The scalar version spent 2304 LDS weight reads
// per pixel (307 us for 32 x 184 x 320).
__global__ __launch_bounds__(256) void head_tail_bf16_mfma_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ w2, float b2, float *__restrict__ maps, int H, int W,
                                                                  int ldc, long total) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    // A fragments: A[r = co][k = ci]: lane (r, h) holds ci = 16 ks + 8 h + j
    bf16x8 af[4][2];
#pragma unroll
    for (int ab = 0; ab < 4; ab++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // (round 6: every load issued, from a clamped index, and the padding selected afterwards -- as `cond ? w1[...] : 0` the compiler
                // made each of the 144 weight / bias loads of this prologue a branch with its own wait: seventy round trips one after the other
                // in front of a loop of nineteen iterations)
                const int ci = 16 * ks + 8 * h + j;
                const float wv = w1[(ab * HT_C + (ci < HT_C ? ci : 0)) * HT_C + (c < HT_C ? c : 0)];
                af[ab][ks][j] = (__bf16)((c < HT_C && ci < HT_C) ? wv : 0.f);
            }
    // this lane's mid channels: co = (i & 3) + 8 (i >> 2) + 4 h; pairs (i, i + 1) and (q, q + 1) share packed fp32 instructions
    f32x2 lb1[8], lw2[16][2];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int co = (i & 3) + 8 * (i >> 2) + 4 * h, coc = co < HT_C ? co : 0;
        const float bv = b1[coc];
        lb1[i >> 1][i & 1] = co < HT_C ? bv : 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) { const float wv = w2[q * HT_C + coc]; lw2[i][q >> 1][q & 1] = co < HT_C ? wv : 0.f; }
    }
    const long nwave = (long)gridDim.x * 4, wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    auto load = [&](long p0, bf16x8 &q0, bf16x8 &q1) {
        const long pix = p0 + c;
        const __bf16 *xp = x + (pix < total ? pix : 0) * ldc;
        q0 = *reinterpret_cast<const bf16x8 *>(xp + 8 * h);
        q1 = zero8();
        if (h == 0) q1 = *reinterpret_cast<const bf16x8 *>(xp + 16);        // channels 16..23; 24..31 (h = 1) are padding: zeros
    };
    long p0 = wid * 32;
    if (p0 >= total) return;
    bf16x8 bq0, bq1;
    load(p0, bq0, bq1);
    for (; p0 < total; p0 += nwave * 32) {
        const long pix = p0 + c;
        const bool live = pix < total;
        const bf16x8 cq0 = bq0, cq1 = bq1;
        if (p0 + nwave * 32 < total) load(p0 + nwave * 32, bq0, bq1);       // the next 32 pixels travel while these are computed
        f32x2 o[4][2];
#pragma unroll
        for (int ab = 0; ab < 4; ab++) {
            f32x16 acc = (f32x16)(0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ab][0], cq0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ab][1], cq1, acc, 0, 0, 0);
            o[ab][0] = o[ab][1] = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 s2 = f32x2{acc[i], acc[i + 1]} + lb1[i >> 1];
                const float m0 = fmaxf(s2[0], 0.f), m1 = fmaxf(s2[1], 0.f);
                o[ab][0] = __builtin_elementwise_fma(f32x2{m0, m0}, lw2[i][0], o[ab][0]);
                o[ab][1] = __builtin_elementwise_fma(f32x2{m0, m0}, lw2[i][1], o[ab][1]);
                o[ab][0] = __builtin_elementwise_fma(f32x2{m1, m1}, lw2[i + 1][0], o[ab][0]);
                o[ab][1] = __builtin_elementwise_fma(f32x2{m1, m1}, lw2[i + 1][1], o[ab][1]);
            }
        }
        // the two halves of the wave hold the two halves of the mid channels of the same pixel: lower lanes finish output rows 0-1
        // (positions ab = 0, 1), upper lanes rows 2-3 (ab = 2, 3): eight values cross, each lane has eight sigmoids and two row stores
        f32x2 mine[2][2];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int qq = 0; qq < 2; qq++) {
                const f32x2 keep = h ? o[2 + k][qq] : o[k][qq], send = h ? o[k][qq] : o[2 + k][qq];
                mine[k][qq] = keep + f32x2{__shfl_xor(send[0], 32), __shfl_xor(send[1], 32)};
            }
        if (live) {
            const unsigned upix = (unsigned)pix, t = upix / (unsigned)W;      // total < 2^31 (host check): 32-bit divisions, the 64-bit ones were ~300 instructions
            const int ox = (int)(upix - t * (unsigned)W);
            const unsigned n = t / (unsigned)H;
            const int oy = (int)(t - n * (unsigned)H);
            float *out = maps + (((long)n * 4 * H + 4 * oy + 2 * h) * (4L * W) + 4 * ox);
#pragma unroll
            for (int a2 = 0; a2 < 2; a2++) {
                f32x4 row;                                      // position b = 0 (mine[0]) then b = 1 (mine[1]), columns bb2 = 0, 1 of each
                row[0] = ht_sigmoid(mine[0][a2][0] + b2);
                row[1] = ht_sigmoid(mine[0][a2][1] + b2);
                row[2] = ht_sigmoid(mine[1][a2][0] + b2);
                row[3] = ht_sigmoid(mine[1][a2][1] + b2);
                *reinterpret_cast<f32x4 *>(out + (long)a2 * (4L * W)) = row;
            }
        }
    }
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_pwconv_bf16(const void *d_x, const void *d_w, const float *d_bias, const void *d_res, const float *d_scale, void *d_y,
                                 int N, int H, int W, int Cin, int Cout_pad, int cstore, int act, int res_mode, int res_ldc, int out_ldc,
                                 int out_coff, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && N >= 1 && H >= 1 && W >= 1, "ptocr_pwconv_bf16: null / empty argument");
    PT_CHECK(Cin % 16 == 0 && Cout_pad % 32 == 0 && cstore % 4 == 0 && cstore >= 4 && cstore <= Cout_pad && act >= 0 && act <= 2,
             "ptocr_pwconv_bf16: need Cin %% 16 == 0, Cout_pad %% 32 == 0 (zero-padded weight rows), cstore %% 4 == 0");
    PT_CHECK(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || d_res) && (res_mode != 2 || (H % 2 == 0 && W % 2 == 0)),
             "ptocr_pwconv_bf16: bad residual mode");
    PT_CHECK(out_ldc >= out_coff + cstore && out_ldc % 8 == 0 && out_coff % 8 == 0 && (res_mode == 0 || res_ldc % 4 == 0),
             "ptocr_pwconv_bf16: bad strides (out_ldc, out_coff multiples of 8: 16-byte stores)");
    PwArgs p;
    p.x = (const __bf16 *)d_x; p.w = (const __bf16 *)d_w; p.res = (const __bf16 *)d_res; p.bias = d_bias; p.scale = d_scale; p.y = (__bf16 *)d_y;
    p.M = (long)N * H * W; p.Cin = Cin; p.ntile = Cout_pad / 32; p.cstore = cstore; p.act = act; p.res_mode = res_mode; p.H = H; p.W = W;
    p.out_ldc = out_ldc; p.out_coff = out_coff; p.res_ldc = res_ldc;
    const long blocks = (p.M + 127) / 128;
    PT_CHECK(blocks < (1L << 31), "ptocr_pwconv_bf16: too many pixels");
    hipStream_t s = (hipStream_t)stream;
#define PT_PW_G(A, G) do { \
        if (res_mode == 0) { \
            if (p.ntile >= 3) hipLaunchKernelGGL((pw_bf16_kernel<3, false, A, G>), dim3((unsigned)blocks, cdiv(p.ntile, 3)), dim3(256), 0, s, p); \
            else if (p.ntile == 2) hipLaunchKernelGGL((pw_bf16_kernel<2, false, A, G>), dim3((unsigned)blocks, 1), dim3(256), 0, s, p); \
            else hipLaunchKernelGGL((pw_bf16_kernel<1, false, A, G>), dim3((unsigned)blocks, 1), dim3(256), 0, s, p); \
        } else { \
            if (p.ntile >= 3) hipLaunchKernelGGL((pw_bf16_kernel<3, true, A, G>), dim3((unsigned)blocks, cdiv(p.ntile, 3)), dim3(256), 0, s, p); \
            else if (p.ntile == 2) hipLaunchKernelGGL((pw_bf16_kernel<2, true, A, G>), dim3((unsigned)blocks, 1), dim3(256), 0, s, p); \
            else hipLaunchKernelGGL((pw_bf16_kernel<1, true, A, G>), dim3((unsigned)blocks, 1), dim3(256), 0, s, p); \
        } } while (0)
#define PT_PW1(A) PT_PW_G(A, 1)
#define PT_PW4(A) PT_PW_G(A, 4)
    // (round 6, measured: groups of TWO slices for 64 <= Cin < 160 -- three waves per SIMD instead of two -- change nothing: 20.8 against
    // 21.5 us for 96 -> 576 channels at 23x40; nor does a split-K form with four times the workgroups: these launches are not bound by the K
    // loop's round trips or by wave slots)
    if (Cin >= 64) PT_ACT_SWITCH(act, PT_PW4); else PT_ACT_SWITCH(act, PT_PW1);
#undef PT_PW1
#undef PT_PW4
#undef PT_PW_G
    return launch_ok("pw_bf16_kernel");
}

// The head's 3x3 conv reading the FPN output as FOUR planes bf16[4][N,H,W,24] (conv3x3_bf16_pers8_kernel<6, ACT, true>): Cin = 96 in the
// concat's channel order (plane, channel), y bf16[N,H,W,32] (cstore channels written, the rest of the 32 zero as in ptocr_conv3x3_bf16).
extern "C" int ptocr_conv3x3_planes_bf16(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int N, int H, int W, int cstore, int act,
                                         int out_ldc, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && N >= 1 && H >= 1 && W >= 1, "ptocr_conv3x3_planes_bf16: null / empty argument");
    PT_CHECK(cstore % 8 == 0 && cstore >= 8 && cstore <= 32 && act >= 0 && act <= 2 && out_ldc >= cstore && out_ldc % 8 == 0,
             "ptocr_conv3x3_planes_bf16: need cstore %% 8 == 0, cstore <= 32 <= out_ldc");
    C3Args p;
    p.x = (const __bf16 *)d_x; p.w = (const __bf16 *)d_w; p.bias = d_bias; p.y = (__bf16 *)d_y; p.M = (long)N * H * W; p.H = H; p.W = W; p.Cin = 96;
    p.cstore = cstore; p.act = act; p.out_up = 1; p.out_ldc = out_ldc; p.out_coff = 0;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int tiles_x = cdiv(W, C3_TW), tpi = tiles_x * cdiv(H, 8);
    const long total = (long)N * tpi;
    PT_CHECK(total < (1L << 31) && (long)H * W * 96 * 2 < (1L << 31), "ptocr_conv3x3_planes_bf16: too many tiles / image larger than 2 GiB");
    const int grid = total < (long)n_cu ? (int)total : n_cu;
#define PT_C3P(A) hipLaunchKernelGGL((conv3x3_bf16_pers8_kernel<6, A, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_x, tpi, (int)total)
    PT_ACT_SWITCH(act, PT_C3P);
#undef PT_C3P
    return launch_ok("conv3x3_bf16_pers8_kernel<planes>");
}

extern "C" int ptocr_conv3x3_bf16(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int N, int H, int W, int Cin, int cstore,
                                  int act, int out_up, int out_ldc, int out_coff, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && N >= 1 && H >= 1 && W >= 1, "ptocr_conv3x3_bf16: null / empty argument");
    PT_CHECK(Cin % 16 == 0 && cstore % 4 == 0 && cstore >= 4 && cstore <= 32 && act >= 0 && act <= 2 && out_up >= 1 && out_up <= 8,
             "ptocr_conv3x3_bf16: need Cin %% 16 == 0, cstore <= 32 (one output tile), out_up <= 8");
    PT_CHECK(out_ldc >= out_coff + cstore && out_ldc % 4 == 0 && out_coff % 4 == 0, "ptocr_conv3x3_bf16: bad strides");
    C3Args p;
    p.x = (const __bf16 *)d_x; p.w = (const __bf16 *)d_w; p.bias = d_bias; p.y = (__bf16 *)d_y; p.M = (long)N * H * W; p.H = H; p.W = W; p.Cin = Cin;
    p.cstore = cstore; p.act = act; p.out_up = out_up; p.out_ldc = out_ldc; p.out_coff = out_coff;
    const long blocks = (p.M + 127) / 128;
    PT_CHECK(blocks < (1L << 31), "ptocr_conv3x3_bf16: too many pixels");
    static const bool use_lds = !(getenv("PTOCR_BF16_C3_LDS") && atoi(getenv("PTOCR_BF16_C3_LDS")) == 0);
    if (use_lds && Cin == 96 && cstore % 8 == 0 && out_coff % 8 == 0 && out_ldc % 8 == 0) {     // the FPN / head convs of the mbv3 detector
        int n_cu = 0;
        if (int e_ = current_device_cus(&n_cu)) return e_;
        const int th = 8;
        const int tiles_x = cdiv(W, C3_TW), tpi = tiles_x * cdiv(H, th);
        const long total = (long)N * tpi;
        PT_CHECK(total < (1L << 31), "ptocr_conv3x3_bf16: too many tiles");
        const int grid = total < (long)n_cu ? (int)total : n_cu;             // one persistent workgroup per CU (weights in its registers)
        PT_CHECK((long)H * W * Cin * 2 < (1L << 31), "ptocr_conv3x3_bf16: image larger than 2 GiB");
        {
#define PT_C3(A) hipLaunchKernelGGL((conv3x3_bf16_pers8_kernel<6, A>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_x, tpi, (int)total)
            PT_ACT_SWITCH(act, PT_C3);
#undef PT_C3
        }
        return launch_ok("conv3x3_bf16_pers8_kernel");
    }
    if (use_lds && Cin <= C3_MAXCIN && N <= 65535) {
        if (Cin == 96) hipLaunchKernelGGL(conv3x3_bf16_lds_kernel<6>, dim3(cdiv(W, C3_TW) * cdiv(H, C3_TH), N), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(conv3x3_bf16_lds_kernel<0>, dim3(cdiv(W, C3_TW) * cdiv(H, C3_TH), N), dim3(256), 0, (hipStream_t)stream, p);
        return launch_ok("conv3x3_bf16_lds_kernel");
    }
    hipLaunchKernelGGL(conv3x3_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return launch_ok("conv3x3_bf16_kernel");
}

// The FPN lateral fused into the smoothing conv (conv3x3_lat_bf16_kernel): y = conv3x3(relu(bn(conv1x1(x2))) + nearest_x2(td)) with the
// intermediate never written.  x2 bf16[N,H,W,16] (16 padded input channels), wl bf16[96][16] + bl f32[96] (the lateral, BN folded),
// td bf16[N,H/2,W/2,td_ldc] (>= 96 channels), w3 bf16[32][9*96] + b3 f32[32] (the smoothing conv), y as in ptocr_conv3x3_bf16.
extern "C" int ptocr_conv3x3_lat_bf16(const void *d_x2, const void *d_wl, const float *d_bl, const void *d_td, int td_ldc, const void *d_w3,
                                      const float *d_b3, void *d_y, int N, int H, int W, int cstore, int act, int out_up, int out_ldc,
                                      int out_coff, void *stream) {
    PT_CHECK(d_x2 && d_wl && d_bl && d_td && d_w3 && d_b3 && d_y && N >= 1 && H >= 2 && W >= 2, "ptocr_conv3x3_lat_bf16: null / empty argument");
    PT_CHECK(H % 2 == 0 && W % 2 == 0 && td_ldc >= 96 && td_ldc % 8 == 0, "ptocr_conv3x3_lat_bf16: even map sizes and a top-down tensor of >= 96 channels");
    PT_CHECK(cstore % 8 == 0 && cstore >= 8 && cstore <= 32 && act >= 0 && act <= 2 && out_up >= 1 && out_up <= 8, "ptocr_conv3x3_lat_bf16: cstore <= 32, out_up <= 8");
    PT_CHECK(out_ldc >= out_coff + cstore && out_ldc % 8 == 0 && out_coff % 8 == 0, "ptocr_conv3x3_lat_bf16: bad strides");
    PT_CHECK((long)H * W * 96 * 2 < (1L << 31), "ptocr_conv3x3_lat_bf16: image larger than 2 GiB");
    C3LatArgs q;
    q.c.x = nullptr; q.c.w = (const __bf16 *)d_w3; q.c.bias = d_b3; q.c.y = (__bf16 *)d_y; q.c.M = (long)N * H * W; q.c.H = H; q.c.W = W; q.c.Cin = 96;
    q.c.cstore = cstore; q.c.act = act; q.c.out_up = out_up; q.c.out_ldc = out_ldc; q.c.out_coff = out_coff;
    q.x2 = (const __bf16 *)d_x2; q.wl = (const __bf16 *)d_wl; q.td = (const __bf16 *)d_td; q.bl = d_bl; q.td_ldc = td_ldc;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int tiles_x = cdiv(W, C3_TW), tpi = tiles_x * cdiv(H, 8);
    const long total = (long)N * tpi;
    PT_CHECK(total < (1L << 31), "ptocr_conv3x3_lat_bf16: too many tiles");
    const int grid = total < (long)n_cu ? (int)total : n_cu;
#define PT_C3L(A) hipLaunchKernelGGL((conv3x3_lat_bf16_kernel<A>), dim3(grid), dim3(512), 0, (hipStream_t)stream, q, tiles_x, tpi, (int)total)
    PT_ACT_SWITCH(act, PT_C3L);
#undef PT_C3L
    return launch_ok("conv3x3_lat_bf16_kernel");
}

extern "C" int ptocr_dwconv_bf16(const void *d_x, const float *d_w, const float *d_bias, void *d_y, float *d_partial, int N, int H, int W,
                                 int C, int k, int stride, int act, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && C % 8 == 0 && C <= 2048 && (k == 3 || k == 5) && (stride == 1 || stride == 2) && act >= 0 && act <= 2 && N <= 65535,
             "ptocr_dwconv_bf16: need C %% 8 == 0, k in {3,5}, stride in {1,2}");
    PT_CHECK(N >= 1 && H >= 1 && W >= 1 && ((long)W + 16) * C < (1L << 31), "ptocr_dwconv_bf16: empty tensor, or a row wider than 2 Gi elements");
    const int pad = (k - 1) / 2;
    DwArgs p;
    p.x = (const __bf16 *)d_x; p.w = d_w; p.bias = d_bias; p.y = (__bf16 *)d_y; p.partial = d_partial; p.H = H; p.W = W; p.C = C; p.k = k;
    p.stride = stride; p.Ho = (H + 2 * pad - k) / stride + 1; p.Wo = (W + 2 * pad - k) / stride + 1; p.act = act;
    p.chunk = dw_rows(N, p.Ho, p.Wo);
    p.nblk = cdiv(p.Ho, p.chunk);
    p.ob = dw_octets(C / 8);
    static const int xcd = getenv("PTOCR_DW_XCD") ? atoi(getenv("PTOCR_DW_XCD")) : 1;
    p.xcd = xcd;
    if (k == 3) return stride == 1 ? launch_dw<3, 1>(p, N, (hipStream_t)stream) : launch_dw<3, 2>(p, N, (hipStream_t)stream);
    return stride == 1 ? launch_dw<5, 1>(p, N, (hipStream_t)stream) : launch_dw<5, 2>(p, N, (hipStream_t)stream);
}

// 1x1 expansion (16 padded input channels) + depthwise 3x3 / stride 2 / pad 1 in one launch (expand_dw3x3s2_bf16_kernel): the expanded
// tensor is never written.  x bf16[N,H,W,16]; we bf16[r32(C)][16] + be f32[r32(C)] (BN folded; rows past the layer's channels zero); wd f32[9][C], bd f32[C]; y
// bf16[N,Ho,Wo,C]; C a multiple of 16, <= 96.  No SE pool (the block it is built for has none).
extern "C" int ptocr_expand_dw3x3s2_bf16(const void *d_x, const void *d_we, const float *d_be, const float *d_wd, const float *d_bd, void *d_y,
                                         int N, int H, int W, int C, int act_e, int act_d, void *stream) {
    PT_CHECK(d_x && d_we && d_be && d_wd && d_bd && d_y && N >= 1 && H >= 1 && W >= 1, "ptocr_expand_dw3x3s2_bf16: null / empty argument");
    PT_CHECK(C % 16 == 0 && C >= 16 && C <= 96 && act_e >= 0 && act_e <= 2 && act_d >= 0 && act_d <= 2, "ptocr_expand_dw3x3s2_bf16: C must be a multiple of 16, <= 96");
    PT_CHECK(N <= 65535, "ptocr_expand_dw3x3s2_bf16: batch too large for one launch");
    ExDwArgs p;
    p.x = (const __bf16 *)d_x; p.we = (const __bf16 *)d_we; p.be = d_be; p.wd = d_wd; p.bd = d_bd; p.y = (__bf16 *)d_y;
    p.H = H; p.W = W; p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; p.C = C;
    const int tiles_x = cdiv(p.Wo, 16), tiles_y = cdiv(p.Ho, 4);
    const dim3 grid(tiles_x * tiles_y, N);
#define PT_XD(CC, E) do { \
        constexpr int lds = 9 * 33 * (2 * CC + 16); \
        static DynLds dyn; \
        if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&expand_dw3x3s2_bf16_kernel<CC, E, E>), lds)) return e_; \
        hipLaunchKernelGGL((expand_dw3x3s2_bf16_kernel<CC, E, E>), grid, dim3(XD_THREADS), (size_t)lds, (hipStream_t)stream, p, tiles_x, tiles_y); \
    } while (0)
#define PT_XD_C(E) do { \
        switch (C) { \
        case 16: PT_XD(16, E); break; case 32: PT_XD(32, E); break; case 48: PT_XD(48, E); break; \
        case 64: PT_XD(64, E); break; case 80: PT_XD(80, E); break; default: PT_XD(96, E); break; \
        } \
    } while (0)
    if (act_e != act_d || (act_e != 1 && act_e != 2))
        return fail("ptocr_expand_dw3x3s2_bf16: the expansion and the depthwise conv must share ReLU or Hardswish (got %d, %d)", act_e, act_d);
    if (act_e == 1) PT_XD_C(1);
    else PT_XD_C(2);
#undef PT_XD_C
#undef PT_XD
    return launch_ok("expand_dw3x3s2_bf16_kernel");
}

extern "C" int ptocr_dwconv_bf16_nblk(int N, int H, int W, int k, int stride) {
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    return cdiv(Ho, dw_rows(N, Ho, Wo));
}

extern "C" int ptocr_stem3x3s2_bf16(const float *d_x, const float *d_w, const float *d_bias, void *d_y, int N, int H, int W, int act, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && N >= 1 && H >= 2 && W >= 2 && act >= 0 && act <= 2, "ptocr_stem3x3s2_bf16: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    PT_CHECK(N <= 65535 && Ho <= 65535, "ptocr_stem3x3s2_bf16: batch or height > 65535");
#define PT_STEM(A) hipLaunchKernelGGL(stem3x3s2_bf16_kernel<A>, dim3((unsigned)((Wo + 255) / 256), (unsigned)Ho, (unsigned)N), dim3(256), 0, (hipStream_t)stream, d_x, d_w, d_bias, \
                                     (__bf16 *)d_y, H, W, Ho, Wo)
    const int pairs = Wo / 2, pblocks = cdiv(pairs, 512), pthreads = cdiv(cdiv(pairs, pblocks), 64) * 64;     // whole waves, no idle tail block (320 pairs: one block of 320)
#define PT_STEM2(A) hipLaunchKernelGGL(stem3x3s2_bf16_pair_kernel<A>, dim3((unsigned)pblocks, (unsigned)Ho, (unsigned)N), dim3(pthreads), 0, (hipStream_t)stream, \
                                      d_x, d_w, d_bias, (__bf16 *)d_y, H, W, Ho, Wo)
    if (W % 4 == 0 && (reinterpret_cast<size_t>(d_x) & 15) == 0) PT_ACT_SWITCH(act, PT_STEM2);
    else PT_ACT_SWITCH(act, PT_STEM);
#undef PT_STEM2
#undef PT_STEM
    return launch_ok("stem3x3s2_bf16_kernel");
}

extern "C" int ptocr_db_head_tail_bf16(const void *d_x, const float *d_w1, const float *d_b1, const float *d_w2, float b2, float *d_maps,
                                       int N, int H, int W, int C, int ldc, void *stream) {
    PT_CHECK(d_x && d_w1 && d_b1 && d_w2 && d_maps && C == HT_C && ldc >= C && ldc % 8 == 0, "ptocr_db_head_tail_bf16: built for C == %d (got %d)", HT_C, C);
    const long total = (long)N * H * W;
    static const bool scalar = getenv("PTOCR_BF16_TAIL_SCALAR") && atoi(getenv("PTOCR_BF16_TAIL_SCALAR")) != 0;
    if (scalar || ldc < 24) {
        hipLaunchKernelGGL(head_tail_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)d_x, d_w1, d_b1,
                           d_w2, b2, d_maps, H, W, ldc, total);
        return launch_ok("head_tail_bf16_kernel");
    }
    PT_CHECK(total < (1L << 31), "ptocr_db_head_tail_bf16: too many pixels");
    const long groups = (total + 127) / 128;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    // few, long-lived waves (the per-lane weight registers are set up once per wave): three workgroups per CU is what the kernel's 150
    // registers allow -- 1024 workgroups ran as one full round and a second one a third full
    const unsigned grid = (unsigned)(groups < 3L * n_cu ? groups : 3L * n_cu);
    hipLaunchKernelGGL(head_tail_bf16_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)d_x, d_w1, d_b1, d_w2, b2, d_maps,
                       H, W, ldc, total);
    return launch_ok("head_tail_bf16_mfma_kernel");
}
