// DB++ Adaptive Scale Fusion around the MFMA conv: attention_type "scale_channel_spatial" (the yml default), "scale_spatial"
// (the same data flow without the channel gate) and "scale_channel" (four softmax weights per image).
// Replaces pytocr/modeling/necks/asf.py:32-75 (ScaleChannelSpatialAttention.forward), :78-107 (ScaleSpatialAttention.forward),
// :9-29 (ScaleChannelAttention.forward) and :146-162 (ScaleFeatureSelection.forward after its 3x3 conv, which runs on conv_mfma).  All four kernels are
// memory-bound (HBM roofline): y = conv(fuse) is f32[N,H,W,64], fuse is f32[N,H,W,256].
//   1. asf_pool_kernel      partial sums of y over pixel chunks            (reads y once)
//   2. asf_channel_kernel   ca = sigmoid(W2 relu(W1 mean(y)))              (one block per image, tiny)
//   3. asf_mean_kernel      s = mean_c(y + ca)   (the reference ADDS the attention, asf.py:67)  -> f32[N,H,W]
//   4. asf_apply_kernel     sa = sigmoid(w1 * relu(conv3x3(s))); g = y + ca + sa (asf.py:72, added again);
//                           score = sigmoid(Wa g) (4 values); fuse[..., 64i:64i+64] *= score_i  (in place)
#include "common.h"

namespace ptocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ASF_C = 64;          // inter channels
constexpr int ASF_MID = 16;
constexpr int ASF_F = 4;           // pyramid levels
constexpr int POOL_PIX = 2048;     // pixels per partial-sum block

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// Where the four 64-channel features come from.  pyr == nullptr: the concat tensor itself (fuse[n, y, x, 64 i + c], re-weighted in place).
// Otherwise a PYRAMID (round 6; ops.Pyramid / ptocr_conv3x3_wino4r_pyramid_f32): level i stored at its own resolution,
// pyr[off[i] + ((n (H >> shift[i]) + (y >> shift[i])) (W >> shift[i]) + (x >> shift[i])) 64 + c] -- the nearest-upsampled copies of
// fpn.py:118-131 are never written; the re-weighting reads the planes and writes the concat ONCE.
struct AsfSrc { const float *pyr; long long off[ASF_F]; int shift[ASF_F]; };
// 16 consecutive channels [16 q, 16 q + 16) of level `level` at pixel (n, py, px) -> address of the source
__device__ __forceinline__ const float *asf_src(const AsfSrc &src, const float *fuse, long pix, int n, int py, int px, int H, int W, int level, int q) {
    if (!src.pyr) return fuse + pix * (ASF_F * ASF_C) + level * ASF_C + q * 16;
    const int sh = src.shift[level];
    return src.pyr + src.off[level] + (((long)n * (H >> sh) + (py >> sh)) * (W >> sh) + (px >> sh)) * ASF_C + q * 16;
}

// block (256 threads) = 16 channel-quads x 16 pixel lanes; deterministic tree reduction
__global__ __launch_bounds__(256) void asf_pool_kernel(const float *__restrict__ y, float *__restrict__ partial, int HW, int nblk) {
    const int n = blockIdx.y, blk = blockIdx.x;
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long base = (long)n * HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int p0 = blk * POOL_PIX, p1 = min(p0 + POOL_PIX, HW);
    for (int p = p0 + pl; p < p1; p += 16) s += *reinterpret_cast<const f32x4 *>(y + (base + p) * ASF_C + cq * 4);
    __shared__ f32x4 sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 16; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 16) *reinterpret_cast<f32x4 *>(partial + ((long)n * nblk + blk) * ASF_C + threadIdx.x * 4) = sh[threadIdx.x];
}

__global__ __launch_bounds__(64) void asf_channel_kernel(const float *__restrict__ partial, const float *__restrict__ w1 /*[16][64]*/,
                                                         const float *__restrict__ w2 /*[64][16]*/, float *__restrict__ ca, int HW, int nblk) {
    const int n = blockIdx.x, c = threadIdx.x;
    __shared__ float mean[ASF_C], mid[ASF_MID];
    float s = 0.f;
    for (int b = 0; b < nblk; b++) s += partial[((long)n * nblk + b) * ASF_C + c];
    mean[c] = s / (float)HW;
    __syncthreads();
    if (c < ASF_MID) {
        float a = 0.f;
        for (int k = 0; k < ASF_C; k++) a += w1[c * ASF_C + k] * mean[k];
        mid[c] = fmaxf(a, 0.f);
    }
    __syncthreads();
    float a = 0.f;
    for (int k = 0; k < ASF_MID; k++) a += w2[c * ASF_MID + k] * mid[k];
    ca[n * ASF_C + c] = sigm(a);
}

// 16 lanes per pixel
__global__ __launch_bounds__(256) void asf_mean_kernel(const float *__restrict__ y, const float *__restrict__ ca, float *__restrict__ smean,
                                                       int HW, long npix) {
    const int sub = threadIdx.x & 15;
    const long pix = blockIdx.x * 16L + (threadIdx.x >> 4);
    float s = 0.f;
    if (pix < npix) {
        const int n = (int)(pix / HW);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(y + pix * ASF_C + sub * 4);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ca + n * ASF_C + sub * 4);
        s = (v[0] + a[0]) + (v[1] + a[1]) + (v[2] + a[2]) + (v[3] + a[3]);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
    if (pix < npix && sub == 0) smean[pix] = s / (float)ASF_C;
}

__global__ __launch_bounds__(256) void asf_apply_kernel(const float *__restrict__ y, const float *__restrict__ ca, const float *__restrict__ smean,
                                                        const float *__restrict__ w3 /*[9]*/, float w1x1, const float *__restrict__ wa /*[4][64]*/,
                                                        float *__restrict__ fuse, int H, int W, long npix, AsfSrc src) {
    const int sub = threadIdx.x & 15;
    const long pix = blockIdx.x * 16L + (threadIdx.x >> 4);
    const int HW = H * W;
    float sc[ASF_F] = {0.f, 0.f, 0.f, 0.f};
    int n = 0, py = 0, px = 0;
    if (pix < npix) {
        n = (int)(pix / HW);
        const int rem = (int)(pix - (long)n * HW);
        py = rem / W; px = rem - py * W;
        float conv = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int yy = py + dy, xx = px + dx;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    conv += w3[(dy + 1) * 3 + dx + 1] * smean[(long)n * HW + (long)yy * W + xx];
            }
        const float sa = sigm(w1x1 * fmaxf(conv, 0.f));
        const f32x4 v = *reinterpret_cast<const f32x4 *>(y + pix * ASF_C + sub * 4);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ca + n * ASF_C + sub * 4);
        f32x4 g;
#pragma unroll
        for (int k = 0; k < 4; k++) g[k] = sa + (a[k] + v[k]);            // spatial_atten + (channel_atten + x)
#pragma unroll
        for (int i = 0; i < ASF_F; i++) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wa + i * ASF_C + sub * 4);
            sc[i] = w[0] * g[0] + w[1] * g[1] + w[2] * g[2] + w[3] * g[3];
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1)
#pragma unroll
        for (int i = 0; i < ASF_F; i++) sc[i] += __shfl_xor(sc[i], o, 16);
    if (pix < npix) {
        // lane `sub` scales channels [16*sub, 16*sub+16) of the 256-channel pixel: level = sub / 4
        const float score = sigm(sc[sub >> 2]);
        float *f = fuse + pix * (ASF_F * ASF_C) + sub * 16;
        const float *g = asf_src(src, fuse, pix, n, py, px, H, W, sub >> 2, sub & 3);
        f32x4 t[4];                                             // all four loads before the first store: a load behind a store waits for its acknowledgement
#pragma unroll
        for (int k = 0; k < 4; k++) t[k] = *reinterpret_cast<const f32x4 *>(g + 4 * k);
#pragma unroll
        for (int k = 0; k < 4; k++) { t[k] *= score; *reinterpret_cast<f32x4 *>(f + 4 * k) = t[k]; }
    }
}

// "scale_channel" (asf.py:9-29): mean over the image -> fc1 (64 -> 32, BatchNorm folded into w1 / b1) -> ReLU -> fc2 (32 -> 4) ->
// softmax over the four levels.  One block per image.
constexpr int ASF_MID2 = 32;
__global__ __launch_bounds__(64) void asf_channel_softmax_kernel(const float *__restrict__ partial, const float *__restrict__ w1 /*[32][64]*/,
                                                                 const float *__restrict__ b1 /*[32]*/, const float *__restrict__ w2 /*[4][32]*/,
                                                                 float *__restrict__ score /*[N][4]*/, int HW, int nblk) {
    const int n = blockIdx.x, c = threadIdx.x;
    __shared__ float mean[ASF_C], mid[ASF_MID2], logit[ASF_F];
    float s = 0.f;
    for (int b = 0; b < nblk; b++) s += partial[((long)n * nblk + b) * ASF_C + c];
    mean[c] = s / (float)HW;
    __syncthreads();
    if (c < ASF_MID2) {
        float a = b1[c];
        for (int k = 0; k < ASF_C; k++) a += w1[c * ASF_C + k] * mean[k];
        mid[c] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (c < ASF_F) {
        float a = 0.f;
        for (int k = 0; k < ASF_MID2; k++) a += w2[c * ASF_MID2 + k] * mid[k];
        logit[c] = a;
    }
    __syncthreads();
    if (c < ASF_F) {
        const float m = fmaxf(fmaxf(logit[0], logit[1]), fmaxf(logit[2], logit[3]));
        const float e0 = expf(logit[0] - m), e1 = expf(logit[1] - m), e2 = expf(logit[2] - m), e3 = expf(logit[3] - m);
        score[n * ASF_F + c] = expf(logit[c] - m) / (e0 + e1 + e2 + e3);
    }
}

// fuse[n, :, :, 64 i : 64 i + 64] *= score[n][i]: the bilinear interpolation of a 1x1 score map (asf.py:154) is that constant
__global__ __launch_bounds__(256) void asf_scale_levels_kernel(float *__restrict__ fuse, const float *__restrict__ score, int H, int W, long nquads,
                                                               AsfSrc src) {
    const long q = blockIdx.x * 256L + threadIdx.x;             // one 16-byte piece of a 256-channel pixel
    if (q >= nquads) return;
    const long pix = q >> 6;
    const int level = (int)(q & 63) >> 4, piece = (int)(q & 15);
    const int HW = H * W;
    const int n = (int)(pix / HW);
    const float sc = score[n * ASF_F + level];
    const float *g = fuse + q * 4;
    if (src.pyr) {
        const int rem = (int)(pix - (long)n * HW);
        const int py = rem / W, px = rem - py * W;
        g = asf_src(src, fuse, pix, n, py, px, H, W, level, 0) + piece * 4;
    }
    f32x4 t = *reinterpret_cast<const f32x4 *>(g);
    t *= sc;
    *reinterpret_cast<f32x4 *>(fuse + q * 4) = t;
}

}  // namespace ptocr

using namespace ptocr;

// fills `src` from the host arrays of a pyramid (nullptr: the concat tensor in place) and checks it against the allocation
static int asf_source(AsfSrc *src, const float *d_pyr, const long long *off, const int *shift, long long pyr_floats, int N, int H, int W) {
    src->pyr = d_pyr;
    for (int j = 0; j < ASF_F; j++) { src->off[j] = 0; src->shift[j] = 0; }
    if (!d_pyr) return 0;
    PT_CHECK(off && shift && pyr_floats > 0, "ptocr_asf_*_pyramid_f32: null plane tables");
    for (int j = 0; j < ASF_F; j++) {
        const int sh = shift[j];
        PT_CHECK(sh >= 0 && sh <= 3 && H % (1 << sh) == 0 && W % (1 << sh) == 0, "ptocr_asf_*_pyramid_f32: plane %d: shift %d does not divide %d x %d", j, sh, H, W);
        const long long need = (long long)N * (H >> sh) * (W >> sh) * ASF_C;
        PT_CHECK(off[j] >= 0 && off[j] % 4 == 0 && off[j] + need <= pyr_floats, "ptocr_asf_*_pyramid_f32: plane %d lies outside the allocation", j);
        src->off[j] = off[j]; src->shift[j] = sh;
    }
    return 0;
}

static int asf_channel_spatial(const float *d_y, float *d_fuse, const AsfSrc &src, const float *d_w_cw1, const float *d_w_cw2,
                               const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work, int N, int H, int W, void *stream) {
    PT_CHECK(d_y && d_fuse && d_w_cw1 && d_w_cw2 && d_w_sp3 && d_w_att && d_work && N >= 1 && N <= 65535, "ptocr_asf: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    const int nblk = cdiv(HW, POOL_PIX);
    const long npix = (long)N * HW;
    float *partial = d_work;                                  // [N][nblk][64]
    float *ca = partial + (long)N * nblk * ASF_C;             // [N][64]
    float *smean = ca + (long)N * ASF_C;                      // [N][H][W]
    hipLaunchKernelGGL(asf_pool_kernel, dim3(nblk, N), dim3(256), 0, s, d_y, partial, HW, nblk);
    hipLaunchKernelGGL(asf_channel_kernel, dim3(N), dim3(64), 0, s, partial, d_w_cw1, d_w_cw2, ca, HW, nblk);
    hipLaunchKernelGGL(asf_mean_kernel, dim3((unsigned)cdiv((int)((npix + 15) / 16 * 16), 16)), dim3(256), 0, s, d_y, ca, smean, HW, npix);
    hipLaunchKernelGGL(asf_apply_kernel, dim3((unsigned)cdiv((int)((npix + 15) / 16 * 16), 16)), dim3(256), 0, s, d_y, ca, smean, d_w_sp3, w_sp1,
                       d_w_att, d_fuse, H, W, npix, src);
    return launch_ok("asf kernels");
}

// attention_type "scale_spatial" (asf.py:78-107): the data flow above with no channel gate (ca = 0: mean_c(y), g = sa + y)
static int asf_spatial(const float *d_y, float *d_fuse, const AsfSrc &src, const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work,
                       int N, int H, int W, void *stream) {
    PT_CHECK(d_y && d_fuse && d_w_sp3 && d_w_att && d_work && N >= 1 && N <= 65535, "ptocr_asf_scale_spatial_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    const long npix = (long)N * HW;
    float *ca = d_work + (long)N * cdiv(HW, POOL_PIX) * ASF_C;  // same layout as ptocr_asf_scale_channel_spatial_f32
    float *smean = ca + (long)N * ASF_C;
    PT_HIP(hipMemsetAsync(ca, 0, sizeof(float) * N * ASF_C, s));
    hipLaunchKernelGGL(asf_mean_kernel, dim3((unsigned)cdiv((int)((npix + 15) / 16 * 16), 16)), dim3(256), 0, s, d_y, ca, smean, HW, npix);
    hipLaunchKernelGGL(asf_apply_kernel, dim3((unsigned)cdiv((int)((npix + 15) / 16 * 16), 16)), dim3(256), 0, s, d_y, ca, smean, d_w_sp3, w_sp1,
                       d_w_att, d_fuse, H, W, npix, src);
    return launch_ok("asf scale_spatial kernels");
}

// attention_type "scale_channel" (asf.py:9-29, 146-162): d_w1 f32[32][64] and d_b1 f32[32] = fc1 with its BatchNorm folded in, d_w2 f32[4][32]
static int asf_channel(const float *d_y, float *d_fuse, const AsfSrc &src, const float *d_w1, const float *d_b1, const float *d_w2, float *d_work,
                       int N, int H, int W, void *stream) {
    PT_CHECK(d_y && d_fuse && d_w1 && d_b1 && d_w2 && d_work && N >= 1 && N <= 65535, "ptocr_asf_scale_channel_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    const int nblk = cdiv(HW, POOL_PIX);
    float *partial = d_work;
    float *score = partial + (long)N * nblk * ASF_C;            // [N][4] in the ca slot
    hipLaunchKernelGGL(asf_pool_kernel, dim3(nblk, N), dim3(256), 0, s, d_y, partial, HW, nblk);
    hipLaunchKernelGGL(asf_channel_softmax_kernel, dim3(N), dim3(64), 0, s, partial, d_w1, d_b1, d_w2, score, HW, nblk);
    const long nquads = (long)N * HW * (ASF_F * ASF_C / 4);
    PT_CHECK(nquads < (1L << 31) * 256L, "ptocr_asf_scale_channel_f32: tensor too large");
    hipLaunchKernelGGL(asf_scale_levels_kernel, dim3((unsigned)((nquads + 255) / 256)), dim3(256), 0, s, d_fuse, score, H, W, nquads, src);
    return launch_ok("asf scale_channel kernels");
}

extern "C" int ptocr_asf_scale_channel_spatial_f32(const float *d_y, float *d_fuse, const float *d_w_cw1, const float *d_w_cw2,
                                                   const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work,
                                                   int N, int H, int W, void *stream) {
    AsfSrc src;
    (void)asf_source(&src, nullptr, nullptr, nullptr, 0, N, H, W);
    return asf_channel_spatial(d_y, d_fuse, src, d_w_cw1, d_w_cw2, d_w_sp3, w_sp1, d_w_att, d_work, N, H, W, stream);
}

extern "C" int ptocr_asf_scale_spatial_f32(const float *d_y, float *d_fuse, const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work,
                                           int N, int H, int W, void *stream) {
    AsfSrc src;
    (void)asf_source(&src, nullptr, nullptr, nullptr, 0, N, H, W);
    return asf_spatial(d_y, d_fuse, src, d_w_sp3, w_sp1, d_w_att, d_work, N, H, W, stream);
}

extern "C" int ptocr_asf_scale_channel_f32(const float *d_y, float *d_fuse, const float *d_w1, const float *d_b1, const float *d_w2, float *d_work,
                                           int N, int H, int W, void *stream) {
    AsfSrc src;
    (void)asf_source(&src, nullptr, nullptr, nullptr, 0, N, H, W);
    return asf_channel(d_y, d_fuse, src, d_w1, d_b1, d_w2, d_work, N, H, W, stream);
}

// The same three on a PYRAMID (round 6): the four features are read from their own-resolution planes (d_pyr, off[4], shift[4], pyr_floats as
// ptocr_conv3x3_wino4r_pyramid_f32, which computes d_y from the same planes) and the re-weighted concat is WRITTEN to d_out f32[N,H,W,256];
// bit-identical to the in-place forms on the materialised concat.  type: 0 scale_channel_spatial (w_a = cw1, w_b = cw2, w_sp3, w_sp1, w_att),
// 1 scale_spatial (w_sp3, w_sp1, w_att), 2 scale_channel (w_a = w1, w_b = b1, w_att = w2).
extern "C" int ptocr_asf_pyramid_f32(int type, const float *d_y, const float *d_pyr, const long long *off, const int *shift, long long pyr_floats,
                                     float *d_out, const float *d_w_a, const float *d_w_b, const float *d_w_sp3, float w_sp1, const float *d_w_att,
                                     float *d_work, int N, int H, int W, void *stream) {
    PT_CHECK(d_pyr && d_out && type >= 0 && type <= 2, "ptocr_asf_pyramid_f32: bad arguments");
    AsfSrc src;
    if (int e = asf_source(&src, d_pyr, off, shift, pyr_floats, N, H, W)) return e;
    if (type == 0) return asf_channel_spatial(d_y, d_out, src, d_w_a, d_w_b, d_w_sp3, w_sp1, d_w_att, d_work, N, H, W, stream);
    if (type == 1) return asf_spatial(d_y, d_out, src, d_w_sp3, w_sp1, d_w_att, d_work, N, H, W, stream);
    return asf_channel(d_y, d_out, src, d_w_a, d_w_b, d_w_att, d_work, N, H, W, stream);
}

extern "C" long ptocr_asf_work_floats(int N, int H, int W) {
    const long HW = (long)H * W;
    return (long)N * cdiv((int)HW, POOL_PIX) * ASF_C + (long)N * ASF_C + (long)N * HW + 64;
}
