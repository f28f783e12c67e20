// ResNet stem WITH its max pool: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels (+ folded BN, ReLU) followed by MaxPool2d(3, 2, 1),
// in one kernel (det_resnet.py:193-197,284-287).  The stem's full-resolution output -- 1.93 GB per batch of 32 -- is what the separate
// pool kernel spent its 0.65 ms reading; here it never reaches HBM: only the pooled map (0.48 GB) is written.
//
// The convolution is conv_stem.hip's (K = 7 x 22 = 154 without the padding channel, A = weights, B = 32 pixels of one output row,
// scalar LDS operand reads with compile-time offsets); what changes is the tile walk and the epilogue:
//   * one 512-thread workgroup per CU (8 waves) owns 16 x 32 stem outputs at a time; wave w holds stem rows 2w (even) and 2w+1 (odd)
//     of the tile, i.e. exactly the rows 2py and 2py+1 of pooled row py = 8 t + w; the third row of that window, 2py-1, is the odd row
//     of the wave above -- one 8 KB row per wave through LDS -- or, for wave 0, the last row of the tile above: a persistent
//     workgroup walks a strip of tiles top to bottom and carries that row over (no vertical recompute);
//   * horizontally a strip is 32 stem columns starting at the ODD column 30 u - 1, so that lanes 2j, 2j+1, 2j+2 are the window of
//     pooled column 15 u + j (j = 0..14): a wave-shift DPP max over the neighbouring lanes; the 32nd column is the first of the
//     next strip (strips advance by 30 columns: 6.7 % of the stem recomputed, the price of not exchanging columns between workgroups);
//   * ReLU output is >= 0, so 0 stands in for the pool's -inf padding and for stem positions outside the image.
// The result is bit-identical to maxpool(stem(x)) (max is exact; the convolution accumulates in the same order as conv_stem.hip).
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_TH = 16, SP_TW = 32;          // stem outputs per tile (rows x columns)
constexpr int SP_SX = 30;                      // stem columns a strip advances by (15 pooled columns)
constexpr int SP_PR = 2 * SP_TH + 5;           // patch rows: 37
constexpr int SP_PC = 2 * SP_TW + 5;           // patch columns: 69
constexpr int SP_RS = SP_PC * 3 + 1;           // floats per patch row in LDS (207 + one zero the 22nd k of the last pixel reads)
constexpr int SP_KR = 22, SP_K = 7 * SP_KR;    // k per kernel row; 154
constexpr int SP_PATCH = SP_PR * SP_RS;
constexpr int SP_THREADS = 512;
constexpr int SP_PIECES = (SP_PR * SP_PC + SP_THREADS - 1) / SP_THREADS;      // 16-byte pieces (pixels) per thread: 5
constexpr int SP_EX = 64 * 16;                 // floats of one wave's odd row, one 32-channel half: 64 lanes x 16 values

struct StemPoolArgs {
    const float *x, *w, *bias;
    float *y;
    int N, H, W, Ho, Wo, Hp, Wp, strips, tiles_y, total;       // total = N * strips * tiles_y (tile index: strip-major, then top to bottom)
    long x_bytes;
};

template <bool NCHW>
__global__ __launch_bounds__(SP_THREADS, 1) void stem_pool_kernel(StemPoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                           // [SP_K][64]
    float *Pb = smem + SP_K * 64;               // [2][SP_PR][SP_RS]
    float *Ex = Pb + 2 * SP_PATCH;              // [8 waves][4 quads][64 lanes][4]: odd rows of the tile, one 32-channel half at a time
    float *Cy = Ex + 8 * SP_EX;                 // [2 halves][4 quads][64 lanes][4]: last row of the tile above (wave 7's odd row)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;

    // Round 6: the weights sit in LDS as [k / 2][cout][k & 1] -- the two k of an MFMA step side by side.  A lane (c, kh) reads W[2 s + kh][c]:
    // in the [k][64] layout of the packed tensor the kh = 0 and kh = 1 halves of the wave fell on the SAME 32 banks (a row is 64 floats),
    // a two-way conflict on every weight read; here the 64 lanes read 64 consecutive floats.
#ifndef STEM_W_PAIRS
#define STEM_W_PAIRS 1
#endif
#if STEM_W_PAIRS
    for (int i = tid; i < SP_K * 64 / 4; i += SP_THREADS) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(p.w)[i];
        const int k = (4 * i) >> 6, co = (4 * i) & 63;
        float *dst = Wl + (((k >> 1) * 64 + co) << 1) + (k & 1);
        dst[0] = v[0]; dst[2] = v[1]; dst[4] = v[2]; dst[6] = v[3];
    }
#else
    for (int i = tid; i < SP_K * 64 / 4; i += SP_THREADS) reinterpret_cast<f32x4 *>(Wl)[i] = reinterpret_cast<const f32x4 *>(p.w)[i];
#endif
    if (tid < 2 * SP_PR) Pb[(tid / SP_PR) * SP_PATCH + (tid % SP_PR) * SP_RS + SP_RS - 1] = 0.f;

    // tile index g -> image n, strip u, tile row t (consecutive g of a workgroup = one strip top to bottom, then its next strip)
    int n, oy0, ox0, trow;
    auto decode = [&](int g) {
        const int s = g / p.tiles_y;
        trow = g - s * p.tiles_y;
        n = s / p.strips;
        oy0 = trow * SP_TH; ox0 = (s - n * p.strips) * SP_SX - 1;
    };
    f32x4 preg[SP_PIECES];
    // (round 6, measured and dropped: the NCHW planes read as 16-byte pieces of four pixels instead of a dword at a time -- 4 loads per thread
    // and tile instead of 15 -- ran 1.64 ms against 1.55: the pieces start at odd pixels, i.e. are never 16-byte aligned)
    auto gload = [&]() {                        // patch of the tile last decoded; outside the image -> zeros (range-checked load)
#pragma unroll
        for (int r = 0; r < SP_PIECES; r++) {
            const int f = tid + SP_THREADS * r;
            const int pr = f / SP_PC, pc = f - pr * SP_PC;
            const int iy = 2 * oy0 - 3 + pr, ix = 2 * ox0 - 3 + pc;
            const bool ok = f < SP_PR * SP_PC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (NCHW) {
                const long plane = (long)p.H * p.W * 4;
                const unsigned off = ok ? (unsigned)((((long)n * 3 * p.H + iy) * p.W + ix) * 4) : oob;
                preg[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
                preg[r][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? (unsigned)(off + plane) : oob, 0, 0));
                preg[r][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? (unsigned)(off + 2 * plane) : oob, 0, 0));
            } else {
                const unsigned off = ok ? (unsigned)((((long)n * p.H + iy) * p.W + ix) * 16) : oob;
                preg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < SP_PIECES; r++) {
            const int f = tid + SP_THREADS * r;
            if (f < SP_PR * SP_PC) {
                const int pr = f / SP_PC, pc = f - pr * SP_PC;
                float *d = Pb + buf * SP_PATCH + pr * SP_RS + pc * 3;
                d[0] = preg[r][0]; d[1] = preg[r][1]; d[2] = preg[r][2];
            }
        }
    };

    const int c = lane & 31, kh = lane >> 5;
#if STEM_W_PAIRS
    const int wb = 2 * c + kh;                                   // weights: W[2s + kh][32 mt + c] at ((s * 64 + 32 mt + c) * 2 + kh)
#else
    const int wb = kh * 64 + c;
#endif
    const int xb = (4 * wave) * SP_RS + 6 * c + kh;              // pixels: patch[2 (2w + nt) + ky][6c + 2s' + kh]

    f32x4 bias4[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int g = 0; g < 4; g++) bias4[mt][g] = *reinterpret_cast<const f32x4 *>(p.bias + 32 * mt + 8 * g + 4 * kh);

    // tiles of this workgroup: the strips blockIdx.x, + gridDim.x, ... each walked top to bottom
    int strip = blockIdx.x;
    int g_tile = strip * p.tiles_y;
    decode(g_tile);
    gload();
    lstore(0);
    __syncthreads();
    for (int it = 0;; it++) {
        const int buf = it & 1;
        const int c_n = n, c_oy0 = oy0, c_ox0 = ox0, c_trow = trow;
        int next = g_tile + 1;                                   // next tile of the strip, or the first of this workgroup's next strip
        if (c_trow + 1 == p.tiles_y) { strip += (int)gridDim.x; next = strip * p.tiles_y; }
        const bool has_next = next < p.total;
        if (has_next) { decode(next); gload(); }                 // in flight during this tile's MFMAs

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = (f32x16)(0.f);
        const float *xp = Pb + buf * SP_PATCH + xb;
        const float *wp = Wl + wb;
#pragma unroll 1
        for (int ky = 0; ky < 7; ky++) {
#pragma unroll
            for (int s = 0; s < SP_KR / 2; s++) {
#if STEM_W_PAIRS
                const float a0 = wp[s * 128], a1 = wp[s * 128 + 64];
#else
                const float a0 = wp[(2 * s) * 64], a1 = wp[(2 * s) * 64 + 32];
#endif
                const float b0 = xp[2 * s], b1 = xp[2 * SP_RS + 2 * s];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            xp += SP_RS;
            wp += SP_KR * 64;                                    // (= (SP_KR / 2) * 128 in the paired layout)
        }

        // ---- epilogue.  Lane (c, kh) holds, for stem pixel (row 2w + nt, column c_ox0 + c), channels 32 mt + 8 g + 4 kh + {0..3}.
        const int ox = c_ox0 + c;
        const bool col_in = (unsigned)ox < (unsigned)p.Wo;
        const bool row_e = c_oy0 + 2 * wave < p.Ho, row_o = c_oy0 + 2 * wave + 1 < p.Ho;
        const int py = (c_oy0 >> 1) + wave;                      // pooled row of this wave
        const int px = (c_ox0 + 1) / 2 + (c >> 1);               // pooled column of the window centred on this (odd) lane
        const bool store = (c & 1) && c < 31 && py < p.Hp && px < p.Wp;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            f32x4 ev[4], od[4];                                  // bias + ReLU; positions outside the stem's output count as 0 (>= 0: ReLU)
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    ev[g][k] = (col_in && row_e) ? fmaxf(acc[mt][0][4 * g + k] + bias4[mt][g][k], 0.f) : 0.f;
                    od[g][k] = (col_in && row_o) ? fmaxf(acc[mt][1][4 * g + k] + bias4[mt][g][k], 0.f) : 0.f;
                }
            }
            // the row above: wave w-1's odd row of this tile, or (wave 0) the carried last row of the tile above / zeros at the top
            f32x4 up[4];
            if (wave == 0) {
#pragma unroll
                for (int g = 0; g < 4; g++) up[g] = c_trow ? *reinterpret_cast<const f32x4 *>(Cy + ((mt * 4 + g) * 64 + lane) * 4) : (f32x4)(0.f);
            }
            __syncthreads();                                     // every wave is done reading the exchange rows of the previous half / tile
#pragma unroll
            for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4 *>(Ex + ((wave * 4 + g) * 64 + lane) * 4) = od[g];     // [wave][g][lane]: 16-byte lanes contiguous
            if (wave == 7) {
#pragma unroll
                for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4 *>(Cy + ((mt * 4 + g) * 64 + lane) * 4) = od[g];
            }
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int g = 0; g < 4; g++) up[g] = *reinterpret_cast<const f32x4 *>(Ex + (((wave - 1) * 4 + g) * 64 + lane) * 4);
            }
            float *yp = p.y + (((long)c_n * p.Hp + py) * p.Wp + px) * 64 + 32 * mt + 4 * kh;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float vm = fmaxf(fmaxf(ev[g][k], od[g][k]), up[g][k]);
                    const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vm), 0x138, 0xf, 0xf, false));   // lane - 1
                    const float r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vm), 0x130, 0xf, 0xf, false));   // lane + 1
                    v[k] = fmaxf(vm, fmaxf(l, r));
                }
                if (store) *reinterpret_cast<f32x4 *>(yp + 8 * g) = v;
            }
        }
        if (!has_next) break;
        lstore(buf ^ 1);                                         // every wave finished reading buf^1 before the barriers above
        __syncthreads();
        g_tile = next;
    }
}

}  // namespace ptocr

using namespace ptocr;

static int stem_pool_launch(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, bool nchw,
                            void *stream, const char *who) {
    PT_CHECK(d_x && d_w && d_bias && d_y, "%s: null argument", who);
    PT_CHECK(N > 0 && H > 0 && W > 0, "%s: empty tensor", who);
    StemPoolArgs a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.Hp = (a.Ho - 1) / 2 + 1; a.Wp = (a.Wo - 1) / 2 + 1;
    a.strips = cdiv(a.Wp, SP_SX / 2); a.tiles_y = cdiv(a.Ho, SP_TH);
    const long total = (long)N * a.strips * a.tiles_y;
    a.x_bytes = (long)N * H * W * (nchw ? 12 : 16);
    PT_CHECK(total < (1L << 31) && a.x_bytes < (1L << 31), "%s: tensor larger than 2 GiB", who);
    a.total = (int)total;
    const size_t lds = sizeof(float) * (SP_K * 64 + 2 * SP_PATCH + 8 * SP_EX + 2 * SP_EX);
    static DynLds dyn0, dyn1;
    if (int e_ = raise_dyn_lds(dyn0, reinterpret_cast<const void *>(&stem_pool_kernel<false>), (int)lds)) return e_;
    if (int e_ = raise_dyn_lds(dyn1, reinterpret_cast<const void *>(&stem_pool_kernel<true>), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int nstrips = N * a.strips;
    const int grid = nstrips < n_cu ? nstrips : n_cu;            // one persistent workgroup per CU, whole strips each
    if (nchw) hipLaunchKernelGGL(stem_pool_kernel<true>, dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_pool_kernel<false>, dim3((unsigned)grid), dim3(SP_THREADS), lds, (hipStream_t)stream, a);
    return launch_ok("stem_pool_kernel");
}

// conv 7x7 / s2 / p3 (3 -> 64, folded BN) + ReLU + MaxPool2d(3, 2, 1): d_x f32[N,H,W,4] (RGB + one ignored channel), d_w as
// ptocr_conv7x7s2_stem_f32, d_y f32[N,Hp,Wp,64] with Ho = (H-1)/2+1, Hp = (Ho-1)/2+1 (likewise W).  Bit-identical to the two calls.
extern "C" int ptocr_conv7x7s2_stem_relu_pool_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                                  void *stream) {
    return stem_pool_launch(d_x, d_w, d_bias, d_y, N, H, W, false, stream, "ptocr_conv7x7s2_stem_relu_pool_f32");
}

// the same from the model's input tensor d_x f32[N,3,H,W]
extern "C" int ptocr_conv7x7s2_stem_relu_pool_nchw_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H,
                                                       int W, void *stream) {
    return stem_pool_launch(d_x, d_w, d_bias, d_y, N, H, W, true, stream, "ptocr_conv7x7s2_stem_relu_pool_nchw_f32");
}
