// ResNet stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels (+ folded BN, ReLU) on fp32 MFMA.
// Replaces the ATen conv2d + batch_norm + relu of det_resnet.py:193-196,284-286 for the 3-channel input.
//
// Why a kernel of its own: the generic implicit GEMM (conv_mfma.hip) gathers 16-byte taps, so the RGB input travels padded to
// 4 channels and K = 7*7*4 = 196 (208 with the k-step padding) although only 147 products are real -- and fp32 MFMA time is
// what bounds the layer.  Here the K axis skips the padding channel: K = 7 rows x (7 pixels x 3 channels + 1 zero) = 154.
//
// A persistent workgroup (256 threads, 2 per CU) keeps the packed weights W[154][64] in LDS (39 KB) for all its tiles and
// walks 8x32-pixel output tiles.  The input patch of a tile (21 x 69 pixels) is fetched with coalesced 16-byte loads from the
// NHWC4 image into registers while the previous tile computes, and stored to LDS as 3 floats per pixel (double-buffered), so
// that one kernel row of one output pixel is 21 contiguous floats: the MFMA operand of output pixel c, k-pair s is the scalar
// LDS read patch[row][6c + 2s + h] with a compile-time offset (LDS reads cost nothing next to fp32 MFMAs; stores and VALU do).
// MFMA roles: A = weights (rows = 32 output channels), B = pixels (columns = 32 pixels of one output row), so a lane ends up
// with 4 x 4 consecutive output channels of ONE pixel and stores 16 bytes at a time straight from the accumulators.
// Wave w owns output rows 2w, 2w+1 of the tile: 2 x 2 MFMA tiles (64 accumulator registers), 77 k-steps of v_mfma_f32_32x32x2_f32.
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 8, ST_TW = 32;           // output tile (rows x columns)
constexpr int ST_PR = 2 * ST_TH + 5;           // patch rows: 21
constexpr int ST_PC = 2 * ST_TW + 5;           // patch columns: 69
constexpr int ST_RS = ST_PC * 3 + 1;           // floats per patch row in LDS (207 + one zero the 22nd k of the last pixel reads)
constexpr int ST_KR = 22;                      // k per kernel row: 7 pixels x 3 channels + 1 zero weight
constexpr int ST_K = 7 * ST_KR;                // 154
constexpr int ST_PATCH = ST_PR * ST_RS;        // floats per patch buffer
constexpr int ST_PIECES = (ST_PR * ST_PC + 255) / 256;     // 16-byte pieces (pixels) per thread: 6

struct StemArgs {
    const float *x, *w, *bias;
    float *y;
    int N, H, W, Ho, Wo, tiles_x, tiles_y, total, relu;
    long x_bytes;
};

template <bool NCHW>                          // input f32[N,3,H,W] planes (the model's own input tensor) instead of f32[N,H,W,4]
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                           // [ST_K][64]
    float *Pb = smem + ST_K * 64;               // [2][ST_PR][ST_RS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;

    for (int i = tid; i < ST_K * 64 / 4; i += 256) reinterpret_cast<f32x4 *>(Wl)[i] = reinterpret_cast<const f32x4 *>(p.w)[i];
    if (tid < 2 * ST_PR) Pb[(tid / ST_PR) * ST_PATCH + (tid % ST_PR) * ST_RS + ST_RS - 1] = 0.f;

    int n, oy0, ox0;
    auto decode = [&](int t) {
        const int per_img = p.tiles_x * p.tiles_y;
        n = t / per_img;
        const int r = t - n * per_img;
        oy0 = (r / p.tiles_x) * ST_TH; ox0 = (r % p.tiles_x) * ST_TW;
    };
    f32x4 preg[ST_PIECES];
    auto gload = [&]() {                        // patch of the tile last decoded; outside the image -> zeros (range-checked load)
#pragma unroll
        for (int r = 0; r < ST_PIECES; r++) {
            const int f = tid + 256 * r;
            const int pr = f / ST_PC, pc = f - pr * ST_PC;
            const int iy = 2 * oy0 - 3 + pr, ix = 2 * ox0 - 3 + pc;
            const bool ok = f < ST_PR * ST_PC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (NCHW) {                         // three plane reads per pixel, consecutive lanes on consecutive x: coalesced
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
        for (int r = 0; r < ST_PIECES; r++) {
            const int f = tid + 256 * r;
            if (f < ST_PR * ST_PC) {
                const int pr = f / ST_PC, pc = f - pr * ST_PC;
                float *d = Pb + buf * ST_PATCH + pr * ST_RS + pc * 3;
                d[0] = preg[r][0]; d[1] = preg[r][1]; d[2] = preg[r][2];
            }
        }
    };

    const int c = lane & 31, kh = lane >> 5;
    const int wb = kh * 64 + c;                                  // weights: W[2s + kh][32 mt + c]
    const int xb = (4 * wave) * ST_RS + 6 * c + kh;              // pixels: patch[2 (2w + nt) + ky][6c + 2s' + kh]

    f32x4 bias4[2][4];                                           // this lane's 8 channel quads (loaded once: no load latency per tile)
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int g = 0; g < 4; g++) bias4[mt][g] = *reinterpret_cast<const f32x4 *>(p.bias + 32 * mt + 8 * g + 4 * kh);

    int tile = blockIdx.x;                                       // host launches gridDim.x <= total
    decode(tile);
    gload();
    lstore(0);
    __syncthreads();
    for (int it = 0;; it++) {
        const int buf = it & 1;
        const int c_n = n, c_oy0 = oy0, c_ox0 = ox0;
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < p.total;
        if (has_next) { decode(next); gload(); }                 // in flight during this tile's MFMAs

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
        const float *xp = Pb + buf * ST_PATCH + xb;
        const float *wp = Wl + wb;
#pragma unroll 1
        for (int ky = 0; ky < 7; ky++) {
#pragma unroll
            for (int s = 0; s < ST_KR / 2; s++) {
                const float a0 = wp[(2 * s) * 64], a1 = wp[(2 * s) * 64 + 32];
                const float b0 = xp[2 * s], b1 = xp[2 * ST_RS + 2 * s];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            xp += ST_RS;
            wp += ST_KR * 64;
        }

        // epilogue: lane holds, for pixel (row 2w + nt, column c), output channels 32 mt + 8 g + 4 kh + {0..3}
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int oy = c_oy0 + 2 * wave + nt, ox = c_ox0 + c;
            if (oy < p.Ho && ox < p.Wo) {
                float *yp = p.y + (((long)c_n * p.Ho + oy) * p.Wo + ox) * 64;
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = 32 * mt + 8 * g + 4 * kh;
                        f32x4 v = f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]} + bias4[mt][g];
                        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        *reinterpret_cast<f32x4 *>(yp + co) = v;
                    }
            }
        }
        if (!has_next) break;
        lstore(buf ^ 1);                                         // every wave finished reading buf^1 before the previous barrier
        __syncthreads();
        tile = next;
    }
}

}  // namespace ptocr

using namespace ptocr;

static int stem_launch(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, int relu, bool nchw,
                       void *stream, const char *who) {
    PT_CHECK(d_x && d_w && d_bias && d_y, "%s: null argument", who);
    PT_CHECK(N > 0 && H > 0 && W > 0, "%s: empty tensor", who);
    PT_CHECK(relu == 0 || relu == 1, "%s: activation must be none or ReLU", who);
    StemArgs a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_x = cdiv(a.Wo, ST_TW); a.tiles_y = cdiv(a.Ho, ST_TH);
    const long total = (long)N * a.tiles_x * a.tiles_y;
    a.x_bytes = (long)N * H * W * (nchw ? 12 : 16);
    PT_CHECK(total < (1L << 31) && a.x_bytes < (1L << 31), "%s: tensor larger than 2 GiB", who);
    a.total = (int)total; a.relu = relu;
    const size_t lds = sizeof(float) * (ST_K * 64 + 2 * ST_PATCH);
    static DynLds dyn0, dyn1;
    if (int e_ = raise_dyn_lds(dyn0, reinterpret_cast<const void *>(&stem_conv_kernel<false>), (int)lds)) return e_;
    if (int e_ = raise_dyn_lds(dyn1, reinterpret_cast<const void *>(&stem_conv_kernel<true>), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int grid = a.total < 2 * n_cu ? a.total : 2 * n_cu;    // two persistent workgroups per CU
    if (nchw) hipLaunchKernelGGL(stem_conv_kernel<true>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stem_conv_kernel<false>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return launch_ok("stem_conv_kernel");
}

// d_x: f32[N,H,W,4] (RGB + one ignored channel); d_w: f32[7][22][64], w[ky][kx*3 + c][cout] with BN folded, [ky][21][*] = 0;
// d_y: f32[N,Ho,Wo,64], Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1.
extern "C" int ptocr_conv7x7s2_stem_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                        int relu, void *stream) {
    return stem_launch(d_x, d_w, d_bias, d_y, N, H, W, relu, false, stream, "ptocr_conv7x7s2_stem_f32");
}

// the same layer straight from the model's input tensor d_x f32[N,3,H,W] (no NCHW -> NHWC boundary pass)
extern "C" int ptocr_conv7x7s2_stem_nchw_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                             int relu, void *stream) {
    return stem_launch(d_x, d_w, d_bias, d_y, N, H, W, relu, true, stream, "ptocr_conv7x7s2_stem_nchw_f32");
}
