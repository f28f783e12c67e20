// Fused DB head tail: ConvTranspose2d(64,64,2,2)+BN+ReLU -> ConvTranspose2d(64,1,2,2)+bias -> sigmoid in ONE kernel.
// Replaces det_db_head.py:13-17 (binarize[3..7]).  The 64-channel half-resolution tensor (1.9 GB at 32x736x1280) is never
// written: each input pixel (1/4 resolution) yields its 4x4 patch of probabilities directly.
//
// GEMM 1 runs on v_mfma_f32_32x32x2_f32 with the roles swapped (A = weights, B = pixels) so the result tile D[co][pixel]
// keeps the pixel on the lane and the 32 mid channels of a tile in registers: the second contraction (over co, only 4 outputs)
// is then register-local FMAs plus one lane^32 exchange -- no LDS round trip for the intermediate.
// Bound: fp32 MFMA for GEMM 1 (1.93 GFLOP/image), HBM for the input read (60 MB/image) and the map write (3.8 MB/image).
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HT_C = 64;                 // channels in and mid
constexpr int HT_LD = HT_C + 4;          // LDS row stride (floats)
constexpr int HT_PIX = 128;              // pixels per tile (4 waves x 32)

// w1: f32[256][64] (row (a*2+b)*64 + co, BN folded), b1: f32[256]; w2: f32[4][64] (a'*2+b'), b2 scalar
// Persistent workgroups, TWO per CU: the 256 x 64 weights sit in LDS (70 KB per workgroup), the pixels do not pass through LDS
// at all -- with B = pixels a lane's operand of a k step is one of four consecutive channels of ITS pixel, so the lane reads its
// 64 channels straight from global memory (eight 16-byte loads, the next tile's while this one computes).  With one wave per SIMD
// (round 2: input tiles double-buffered in LDS, 139 KB) the ~6 k cycles of epilogue per tile -- bias, ReLU, the second contraction,
// sigmoid -- ran with the matrix pipe idle (56 % busy); with two, one workgroup's epilogue runs under the other's MFMAs.
__global__ __launch_bounds__(256, 2) void db_head_tail_kernel(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                                                              const float *__restrict__ w2, float b2, float *__restrict__ maps, int H, int W, long npix,
                                                              int ntiles, long x_bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ws = smem;                                        // [256 col][HT_LD]
    float *w2s = ws + 256 * HT_LD;                           // [64][4]
    float *b1s = w2s + 4 * HT_C;                             // [256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)x_bytes, 0x00020000);
    for (int i = tid; i < 256 * 16; i += 256) {
        const int r = i >> 4, c4 = i & 15;
        *reinterpret_cast<f32x4 *>(&ws[r * HT_LD + c4 * 4]) = *reinterpret_cast<const f32x4 *>(w1 + (long)r * HT_C + c4 * 4);
    }
    w2s[(tid & 63) * 4 + (tid >> 6)] = w2[tid];               // [co][a'b']: the four weights of a mid channel are ONE 16-byte read
    b1s[tid] = b1[tid];
    const int j = lane & 31, h = lane >> 5;                  // pixel within the wave's 32, k half
    f32x4 xcur[8], xnext[8];                                 // B operand: pixel j, k = 8 kk + 4 h + t
    auto gload = [&](int tile, f32x4 *dst) {                 // beyond npix: out of range -> zeros
        const long m = (long)tile * HT_PIX + wave * 32 + j;
#pragma unroll
        for (int kk = 0; kk < 8; kk++)
            dst[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(m * (HT_C * 4) + (kk * 8 + 4 * h) * 4), 0, 0));
    };
    int tile = blockIdx.x;                                   // host launches gridDim.x <= ntiles
    gload(tile, xcur);
    __syncthreads();
  for (;;) {
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) gload(next, xnext);                        // in flight during this tile's MFMAs
    const long p0 = (long)tile * HT_PIX;
    const long pix = p0 + wave * 32 + j;
    const bool store = h == 0 && pix < npix;
    float *o = maps;
    if (store) {
        const long n = pix / ((long)H * W);
        const int rem = (int)(pix - n * (long)H * W);
        const int y = rem / W, xx = rem - y * W;
        o = maps + (n * 4 * H + 4 * y) * (long)(4 * W) + 4 * xx;
    }
    // (a, b) groups two at a time -- the pair (a, 0), (a, 1) fills two output rows; the loop over a stays rolled: at two waves per SIMD
    // the register budget is 256, and four unrolled groups spilled 1.8 KB per lane
#pragma unroll 1
    for (int a = 0; a < 2; a++) {
        float outv[2][4];                                    // [b][a'b'] partial sums of this lane
#pragma unroll
        for (int bb = 0; bb < 2; bb++) {
            const int g = a * 2 + bb;
            asm volatile("" ::: "memory");                      // the epilogue's 192 LDS operands are read here, per group: hoisted out of the tile loop they spill
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
            const float *aw0 = ws + (g * 64 + j) * HT_LD + 4 * h;    // A operand: row (co) j of tile 0
            const float *aw1 = aw0 + 32 * HT_LD;
#pragma unroll
            for (int kk = 0; kk < HT_C / 8; kk++) {
                const f32x4 b = xcur[kk];
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(aw0 + kk * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(aw1 + kk * 8);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b[t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b[t], acc1, 0, 0, 0);
                }
            }
            // D[row = co][col = pixel j]: register r holds co = (r&3) + 8*(r>>2) + 4*h (+32 for tile 1)
            // bias + ReLU + second contraction in packed fp32 (pairs of mid channels for the bias add, pairs of outputs for the FMAs); the
            // four w2 of a channel and four consecutive b1 are one 16-byte LDS read each: 40 reads per group instead of 160 four-byte ones
            // (round 4: this epilogue, ~350 VALU + 160 LDS instructions per group, was what kept the matrix pipe at 64 %)
            f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
                const int cq0 = 8 * rq + 4 * h;                      // co of register 4 rq (tile 0); tile 1: + 32
                const f32x4 bq0 = *reinterpret_cast<const f32x4 *>(b1s + g * 64 + cq0), bq1 = *reinterpret_cast<const f32x4 *>(b1s + g * 64 + 32 + cq0);
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    const f32x2 s0 = f32x2{acc0[4 * rq + t], acc0[4 * rq + t + 1]} + f32x2{bq0[t], bq0[t + 1]};
                    const f32x2 s1 = f32x2{acc1[4 * rq + t], acc1[4 * rq + t + 1]} + f32x2{bq1[t], bq1[t + 1]};
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const float m0 = fmaxf(s0[u], 0.f), m1 = fmaxf(s1[u], 0.f);
                        const f32x4 wa = *reinterpret_cast<const f32x4 *>(w2s + (cq0 + t + u) * 4), wb = *reinterpret_cast<const f32x4 *>(w2s + (32 + cq0 + t + u) * 4);
                        o01 = __builtin_elementwise_fma(f32x2{m0, m0}, f32x2{wa[0], wa[1]}, o01);
                        o23 = __builtin_elementwise_fma(f32x2{m0, m0}, f32x2{wa[2], wa[3]}, o23);
                        o01 = __builtin_elementwise_fma(f32x2{m1, m1}, f32x2{wb[0], wb[1]}, o01);
                        o23 = __builtin_elementwise_fma(f32x2{m1, m1}, f32x2{wb[2], wb[3]}, o23);
                    }
                }
            }
            outv[bb][0] = o01[0]; outv[bb][1] = o01[1]; outv[bb][2] = o23[0]; outv[bb][3] = o23[1];
            // the other half of the mid channels lives in lane ^ 32
#pragma unroll
            for (int oo = 0; oo < 4; oo++) outv[bb][oo] += __shfl_xor(outv[bb][oo], 32);
        }
        // The next tile's pixels (requested at the top of this tile) are taken over HERE, in front of the tile's last stores (round 5b): the
        // vector-memory counter is in order and the loop is rolled, so at the end of the tile the wait for them was a wait for everything,
        // the stores just issued included.
        if (a == 1 && has_next) {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) xcur[kk] = xnext[kk];
        }
        if (store) {
            // output pixel (4y + 2a + a', 4x + 2b + b')
#pragma unroll
            for (int ap = 0; ap < 2; ap++) {
                f32x4 row;
                row[0] = 1.f / (1.f + expf(-(outv[0][ap * 2 + 0] + b2)));
                row[1] = 1.f / (1.f + expf(-(outv[0][ap * 2 + 1] + b2)));
                row[2] = 1.f / (1.f + expf(-(outv[1][ap * 2 + 0] + b2)));
                row[3] = 1.f / (1.f + expf(-(outv[1][ap * 2 + 1] + b2)));
                *reinterpret_cast<f32x4 *>(o + (long)(2 * a + ap) * (4 * W)) = row;
            }
        }
    }
    if (!has_next) break;
    tile = next;
  }
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_db_head_tail_f32(const float *d_x, const float *d_w1, const float *d_b1, const float *d_w2, float b2, float *d_maps,
                                      int N, int H, int W, int C, void *stream) {
    PT_CHECK(d_x && d_w1 && d_b1 && d_w2 && d_maps && N >= 1, "ptocr_db_head_tail_f32: bad arguments");
    PT_CHECK(C == HT_C, "ptocr_db_head_tail_f32: the fused tail is specialised for 64 channels (got %d)", C);
    const long npix = (long)N * H * W;
    const long x_bytes = npix * HT_C * 4;
    PT_CHECK(x_bytes < (1L << 31), "ptocr_db_head_tail_f32: tensor larger than 2 GiB");
    const int ntiles = (int)((npix + HT_PIX - 1) / HT_PIX);
    const size_t lds = sizeof(float) * (256 * HT_LD + 4 * HT_C + 256);
    static DynLds dyn;
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(db_head_tail_kernel), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    hipLaunchKernelGGL(db_head_tail_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, d_x, d_w1, d_b1, d_w2, b2, d_maps, H, W,
                       npix, ntiles, x_bytes);
    return launch_ok("db_head_tail_kernel");
}
