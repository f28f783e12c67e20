// MobileNetV3 pieces that are not GEMMs: depthwise conv and Squeeze-Excitation.  Both are memory-bound (HBM roofline).
// Replaces nn.Conv2d(groups=C)+BN+act (det_mobilenet_v3.py:123-126) and SqueezeExcitation (:67-85).
#include "common.h"

namespace ptocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act1(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return v;
}

// one thread per (output pixel, 4 channels); consecutive threads walk channels, so loads/stores are 16 B coalesced
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                     float *__restrict__ y, int H, int W, int C4, int k, int stride, int stride_w, int Ho, int Wo, int act, long total) {
    const int pad = (k - 1) / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long n = r / Ho;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(bias + c * 4);
        for (int a = 0; a < k; a++) {
            const int iy = oy * stride - pad + a;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int b = 0; b < k; b++) {
                const int ix = ox * stride_w - pad + b;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((n * H + iy) * W + ix) * (long)C4 + c) * 4);
                const f32x4 ww = *reinterpret_cast<const f32x4 *>(w + ((long)(a * k + b) * C4 + c) * 4);
                acc += v * ww;
            }
        }
        acc[0] = act1(acc[0], act); acc[1] = act1(acc[1], act); acc[2] = act1(acc[2], act); acc[3] = act1(acc[3], act);
        *reinterpret_cast<f32x4 *>(y + i * 4) = acc;
    }
}

constexpr int SE_PIX = 2048;

// partial sums over pixel chunks: block = (C/4 channel quads) x (L = 256 / (C/4) pixel lanes); lane pl sums pixels p0 + pl,
// p0 + pl + L, ... (coalesced along channels), the L partial sums meet in LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void se_pool_kernel(const float *__restrict__ x, float *__restrict__ partial, int HW, int C, int nblk) {
    __shared__ f32x4 red[256];
    const int n = blockIdx.y, blk = blockIdx.x;
    const int C4 = C >> 2;
    const int p0 = blk * SE_PIX, p1 = min(p0 + SE_PIX, HW);
    if (C4 <= 256) {
        const int L = 256 / C4;
        const int q = threadIdx.x % C4, pl = threadIdx.x / C4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (pl < L)
            for (int p = p0 + pl; p < p1; p += L) s += *reinterpret_cast<const f32x4 *>(x + ((long)n * HW + p) * C + q * 4);
        red[threadIdx.x] = s;
        __syncthreads();
        if (pl == 0) {
            for (int l = 1; l < L; l++) s += red[l * C4 + q];
            *reinterpret_cast<f32x4 *>(partial + ((long)n * nblk + blk) * C + q * 4) = s;
        }
        return;
    }
    // more than 1024 channels: thread t handles channel quads q = t, t+256, ... over all pixels of the chunk
    for (int q = threadIdx.x; q < C4; q += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int p = p0; p < p1; p++) s += *reinterpret_cast<const f32x4 *>(x + ((long)n * HW + p) * C + q * 4);
        *reinterpret_cast<f32x4 *>(partial + ((long)n * nblk + blk) * C + q * 4) = s;
    }
}

__global__ __launch_bounds__(256) void se_fc_kernel(const float *__restrict__ partial, const float *__restrict__ w1, const float *__restrict__ b1,
                                                    const float *__restrict__ w2, const float *__restrict__ b2, float *__restrict__ scale,
                                                    int HW, int C, int S, int nblk) {
    const int n = blockIdx.x;
    __shared__ float mean[1024], mid[256];
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int b = 0; b < nblk; b++) s += partial[((long)n * nblk + b) * C + c];
        mean[c] = s / (float)HW;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < S; j += 256) {
        float a = b1[j];
        for (int c = 0; c < C; c++) a += w1[(long)j * C + c] * mean[c];
        mid[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = b2[c];
        for (int j = 0; j < S; j++) a += w2[(long)c * S + j] * mid[j];
        scale[(long)n * C + c] = fminf(fmaxf(a + 3.f, 0.f), 6.f) * (1.f / 6.f);       // hardsigmoid
    }
}

// the same gate from TRANSPOSED weights (w1t f32[C][S], w2t f32[S][C]): a thread per output whose loads are coalesced across the
// threads and independent of each other -- with the row-major layouts above a thread walks its own row (576 dependent strided
// loads: 74 us for a 32 x 576 vector); this form takes a few microseconds
template <int NT>                                              // threads per block: 256, or 1024 for the wide layers (C >= 256: more K slices per output)
__global__ __launch_bounds__(NT) void se_fc_t_kernel(const float *__restrict__ partial, const float *__restrict__ w1t, const float *__restrict__ b1,
                                                      const float *__restrict__ w2t, const float *__restrict__ b2, float *__restrict__ scale,
                                                      int HW, int C, int S, int nblk) {
    const int n = blockIdx.x;
    __shared__ float mean[1024], mid[256];
    for (int c = threadIdx.x; c < C; c += NT) {                  // block sums in their fixed order, sixteen loads in flight
        const float *pp = partial + (long)n * nblk * C + c;
        float s = 0.f;
        int b = 0;
        for (; b + 15 < nblk; b += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = pp[(long)(b + u) * C];
#pragma unroll
            for (int u = 0; u < 16; u++) s += v[u];
        }
        for (; b + 7 < nblk; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = pp[(long)(b + u) * C];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; b < nblk; b++) s += pp[(long)b * C];
        mean[c] = s / (float)HW;
    }
    __syncthreads();
    // Latency, not bandwidth, is what these tiny products cost (one block per image, a handful of waves per CU).  Round 6: an output is
    // split over K SLICES (thread = (output, slice), NT / outputs of them; the slices meet in LDS and are added in a fixed order) with
    // sixteen independent loads in flight per thread: a 240 -> 64 product was fifteen dependent rounds of L2 loads for 64 busy threads,
    // now four for 256 (se_fc_t_kernel 8-16 us -> see DESIGN 3.5).
    __shared__ float red[NT];
    {
        const int KS = NT / S;                                       // (S <= 256 <= NT: host check)
        const int jl = threadIdx.x % S, ks = threadIdx.x / S;
        float a = 0.f;
        if (ks < KS) {
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u] = 0.f;
            int c = ks;
            for (; c + 15 * KS < C; c += 16 * KS) {
#pragma unroll
                for (int u = 0; u < 16; u++) acc[u] += w1t[(c + u * KS) * S + jl] * mean[c + u * KS];
            }
            for (; c < C; c += KS) acc[0] += w1t[c * S + jl] * mean[c];
            #pragma unroll
            for (int u = 8; u > 0; u >>= 1)
#pragma unroll
                for (int v = 0; v < u; v++) acc[v] += acc[v + u];
            a = acc[0];
        }
        red[threadIdx.x] = a;
        __syncthreads();
        if ((int)threadIdx.x < S) {
            float sum = 0.f;
            for (int k = 0; k < KS; k++) sum += red[k * S + threadIdx.x];
            mid[threadIdx.x] = fmaxf(sum + b1[threadIdx.x], 0.f);
        }
        __syncthreads();
    }
    for (int c0 = 0; c0 < C; c0 += NT) {                        // gate channels [c0, c0 + nc)
        const int nc = C - c0 < NT ? C - c0 : NT;
        const int KS = NT / nc;
        const int cl = threadIdx.x % nc, ks = threadIdx.x / nc;
        float a = 0.f;
        if (ks < KS) {
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u] = 0.f;
            int j = ks;
            for (; j + 15 * KS < S; j += 16 * KS) {
#pragma unroll
                for (int u = 0; u < 16; u++) acc[u] += w2t[(j + u * KS) * C + c0 + cl] * mid[j + u * KS];
            }
            for (; j < S; j += KS) acc[0] += w2t[j * C + c0 + cl] * mid[j];
            #pragma unroll
            for (int u = 8; u > 0; u >>= 1)
#pragma unroll
                for (int v = 0; v < u; v++) acc[v] += acc[v + u];
            a = acc[0];
        }
        __syncthreads();                                         // (red: the previous round's readers are done)
        red[threadIdx.x] = a;
        __syncthreads();
        if ((int)threadIdx.x < nc) {
            float sum = 0.f;
            for (int k = 0; k < KS; k++) sum += red[k * nc + threadIdx.x];
            scale[(long)n * C + c0 + threadIdx.x] = fminf(fmaxf(sum + b2[c0 + threadIdx.x] + 3.f, 0.f), 6.f) * (1.f / 6.f);       // hardsigmoid
        }
    }
}

// Wide layers (C >= 256): one block per image leaves 224 of 256 CUs idle for the 30-50 us the two dependent matrix-vector products
// take (dependent rounds of L2 loads).  Two kernels of SE_SPLIT blocks per image instead: the first recomputes the pooled vector
// (cheap, same fixed order) and produces a slice of the hidden units, the second a slice of the gate; inside a block every output
// is split over K slices (thread = (output, slice)) and the slices are summed in a fixed order.
constexpr int SE_SPLIT = 8;
__global__ __launch_bounds__(256) void se_fc1_split_kernel(const float *__restrict__ partial, const float *__restrict__ w1t, const float *__restrict__ b1,
                                                           float *__restrict__ hidden, int HW, int C, int S, int nblk) {
    const int n = blockIdx.x, part = blockIdx.y;
    __shared__ float mean[1024], red[256];
    for (int c = threadIdx.x; c < C; c += 256) {
        const float *pp = partial + (long)n * nblk * C + c;
        float s = 0.f;
        int b = 0;
        for (; b + 7 < nblk; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = pp[(long)(b + u) * C];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; b < nblk; b++) s += pp[(long)b * C];
        mean[c] = s / (float)HW;
    }
    __syncthreads();
    const int per = (S + SE_SPLIT - 1) / SE_SPLIT;               // hidden units of this block: [j0, j0 + nj)
    const int j0 = part * per, nj = min(per, S - j0);
    if (nj <= 0) return;
    const int KS = 256 / per;                                    // K slices per output (per <= 32 for S <= 256)
    const int jl = threadIdx.x % per, ks = threadIdx.x / per;
    float a = 0.f;
    if (jl < nj && ks < KS) {
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u] = 0.f;
        int c = ks;
        for (; c + 7 * KS < C; c += 8 * KS) {
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] += w1t[(long)(c + u * KS) * S + j0 + jl] * mean[c + u * KS];
        }
        for (; c < C; c += KS) acc[0] += w1t[(long)c * S + j0 + jl] * mean[c];
        a = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < nj) {
        float s = 0.f;
        for (int k = 0; k < KS; k++) s += red[k * per + threadIdx.x];
        hidden[(long)n * S + j0 + threadIdx.x] = fmaxf(s + b1[j0 + threadIdx.x], 0.f);
    }
}

__global__ __launch_bounds__(256) void se_fc2_split_kernel(const float *__restrict__ hidden, const float *__restrict__ w2t, const float *__restrict__ b2,
                                                           float *__restrict__ scale, int C, int S) {
    const int n = blockIdx.x, part = blockIdx.y;
    __shared__ float mid[256], red[256];
    for (int j = threadIdx.x; j < S; j += 256) mid[j] = hidden[(long)n * S + j];
    __syncthreads();
    const int per = (C + SE_SPLIT - 1) / SE_SPLIT;               // gate channels of this block (per <= 128 for C <= 1024)
    const int c0 = part * per, nc = min(per, C - c0);
    if (nc <= 0) return;
    const int KS = 256 / per;
    const int cl = threadIdx.x % per, ks = threadIdx.x / per;
    float a = 0.f;
    if (cl < nc && ks < KS) {
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u] = 0.f;
        int j = ks;
        for (; j + 7 * KS < S; j += 8 * KS) {
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] += w2t[(long)(j + u * KS) * C + c0 + cl] * mid[j + u * KS];
        }
        for (; j < S; j += KS) acc[0] += w2t[(long)j * C + c0 + cl] * mid[j];
        a = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < nc) {
        float s = 0.f;
        for (int k = 0; k < KS; k++) s += red[k * per + threadIdx.x];
        scale[(long)n * C + c0 + threadIdx.x] = fminf(fmaxf(s + b2[c0 + threadIdx.x] + 3.f, 0.f), 6.f) * (1.f / 6.f);       // hardsigmoid
    }
}

__global__ __launch_bounds__(256) void se_apply_kernel(float *__restrict__ x, const float *__restrict__ scale, int HW, int C4, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const long n = i / ((long)HW * C4);
        f32x4 v = *reinterpret_cast<f32x4 *>(x + i * 4);
        v *= *reinterpret_cast<const f32x4 *>(scale + (n * C4 + c) * 4);
        *reinterpret_cast<f32x4 *>(x + i * 4) = v;
    }
}

static inline int grid_cap(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g < 2048 ? (g > 0 ? g : 1) : 2048);
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_dwconv2_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, int C, int k,
                                 int stride_h, int stride_w, int act, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y && C % 4 == 0 && (k == 2 || k == 3 || k == 5) && (stride_h == 1 || stride_h == 2) && (stride_w == 1 || stride_w == 2) &&
             act >= 0 && act <= 2, "ptocr_dwconv2_f32: need C %% 4 == 0, k in {2,3,5} (padding (k-1)/2: none for k = 2), strides in {1,2}");
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride_h + 1, Wo = (W + 2 * pad - k) / stride_w + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_cap(total, 256)), dim3(256), 0, (hipStream_t)stream, d_x, d_w, d_bias, d_y, H, W, C / 4, k,
                       stride_h, stride_w, Ho, Wo, act, total);
    return launch_ok("dwconv_kernel");
}

extern "C" int ptocr_dwconv_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, int C, int k,
                                int stride, int act, void *stream) {
    return ptocr_dwconv2_f32(d_x, d_w, d_bias, d_y, N, H, W, C, k, stride, stride, act, stream);
}

// ClsHead (pytocr/modeling/heads/cls_head.py:16-29) behind the recognition-style MobileNetV3's AvgPool2d(2, 2)
// (rec_mobilenet_v3.py:221,266): mean over the 2x2 blocks that fit (a trailing odd row / column is dropped, as the pool's floor
// does), then over the blocks -- both are equal-weight means, so one mean over the cropped map -- then Linear + softmax.
// one workgroup per image; x f32[N][H][W][C] (C % 4 == 0), w f32[K][C], b f32[K], K <= 64 -> y f32[N][K]
__global__ __launch_bounds__(256) void cls_head_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                       float *__restrict__ y, int H, int W, int C, int K) {
    __shared__ float mean[1024], logit[64];
    const int n = blockIdx.x;
    const int Hc = H & ~1, Wc = W & ~1;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int yy = 0; yy < Hc; yy++)
            for (int xx = 0; xx < Wc; xx++) s += x[(((long)n * H + yy) * W + xx) * C + c];
        mean[c] = s / (float)(Hc * Wc);
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float a = b[threadIdx.x];
        for (int c = 0; c < C; c++) a += w[(long)threadIdx.x * C + c] * mean[c];
        logit[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float m = -INFINITY, s = 0.f;
        for (int k2 = 0; k2 < K; k2++) m = fmaxf(m, logit[k2]);
        for (int k2 = 0; k2 < K; k2++) s += expf(logit[k2] - m);
        y[(long)n * K + threadIdx.x] = expf(logit[threadIdx.x] - m) / s;
    }
}

extern "C" int ptocr_cls_head_f32(const float *d_x, const float *d_w, const float *d_b, float *d_y, int N, int H, int W, int C, int K,
                                  void *stream) {
    PT_CHECK(d_x && d_w && d_b && d_y && N >= 1 && H >= 2 && W >= 2 && C >= 1 && C <= 1024 && K >= 1 && K <= 64,
             "ptocr_cls_head_f32: need H, W >= 2, C <= 1024, K <= 64");
    hipLaunchKernelGGL(cls_head_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, d_x, d_w, d_b, d_y, H, W, C, K);
    return launch_ok("cls_head_kernel");
}

extern "C" int ptocr_se_scale_f32(float *d_x, const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, float *d_work,
                                  int N, int H, int W, int C, int S, void *stream) {
    PT_CHECK(d_x && d_w1 && d_b1 && d_w2 && d_b2 && d_work && C % 4 == 0 && C <= 1024 && S <= 256 && N <= 65535, "ptocr_se_scale_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W, nblk = cdiv(HW, SE_PIX);
    float *partial = d_work, *scale = d_work + (long)N * nblk * C;
    hipLaunchKernelGGL(se_pool_kernel, dim3(nblk, N), dim3(256), 0, s, d_x, partial, HW, C, nblk);
    hipLaunchKernelGGL(se_fc_kernel, dim3(N), dim3(256), 0, s, partial, d_w1, d_b1, d_w2, d_b2, scale, HW, C, S, nblk);
    const long total = (long)N * HW * (C / 4);
    hipLaunchKernelGGL(se_apply_kernel, dim3(grid_cap(total, 256)), dim3(256), 0, s, d_x, scale, HW, C / 4, total);
    return launch_ok("se kernels");
}

// Squeeze-Excitation gate alone (det_mobilenet_v3.py:76-85 without the pooling pass and without the multiply): d_partial
// f32[N][nblk][C] per-chunk channel sums (left by ptocr_dwconv_bf16) -> d_scale f32[N][C] = hardsigmoid(fc2(relu(fc1(mean)))).
// the same from transposed weights d_w1t f32[C][S], d_w2t f32[S][C] (coalesced, a few microseconds; same result up to fp32 summation order)
extern "C" int ptocr_se_fc_t_f32(const float *d_partial, const float *d_w1t, const float *d_b1, const float *d_w2t, const float *d_b2,
                                 float *d_scale, int N, int HW, int C, int S, int nblk, void *stream) {
    PT_CHECK(d_partial && d_w1t && d_b1 && d_w2t && d_b2 && d_scale && C <= 1024 && S <= 256 && N >= 1 && HW >= 1 && nblk >= 1,
             "ptocr_se_fc_t_f32: bad arguments (C <= 1024, S <= 256)");
    // (round 6: ONE block of 1024 threads per image from 64 channels on -- sixteen K slices per hidden unit at 240 -> 64, seven at 576 -> 144: a product is one or
    // two rounds of loads -- instead of 256 threads, and instead of the split pair of launches for the wide layers)
    if (C >= 64) hipLaunchKernelGGL(se_fc_t_kernel<1024>, dim3(N), dim3(1024), 0, (hipStream_t)stream, d_partial, d_w1t, d_b1, d_w2t, d_b2, d_scale, HW, C, S, nblk);
    else hipLaunchKernelGGL(se_fc_t_kernel<256>, dim3(N), dim3(256), 0, (hipStream_t)stream, d_partial, d_w1t, d_b1, d_w2t, d_b2, d_scale, HW, C, S, nblk);
    return launch_ok("se_fc_t_kernel");
}

// the same gate for wide layers, SE_SPLIT blocks per image in two launches; d_hidden: workspace f32[N][S] (the caller's: no state here)
extern "C" int ptocr_se_fc_split_f32(const float *d_partial, const float *d_w1t, const float *d_b1, const float *d_w2t, const float *d_b2,
                                     float *d_hidden, float *d_scale, int N, int HW, int C, int S, int nblk, void *stream) {
    PT_CHECK(d_partial && d_w1t && d_b1 && d_w2t && d_b2 && d_hidden && d_scale && C <= 1024 && S <= 256 && N >= 1 && N <= 65535 && HW >= 1 && nblk >= 1,
             "ptocr_se_fc_split_f32: bad arguments (C <= 1024, S <= 256)");
    hipLaunchKernelGGL(se_fc1_split_kernel, dim3(N, SE_SPLIT), dim3(256), 0, (hipStream_t)stream, d_partial, d_w1t, d_b1, d_hidden, HW, C, S, nblk);
    hipLaunchKernelGGL(se_fc2_split_kernel, dim3(N, SE_SPLIT), dim3(256), 0, (hipStream_t)stream, d_hidden, d_w2t, d_b2, d_scale, C, S);
    return launch_ok("se_fc_split_kernel");
}

extern "C" int ptocr_se_fc_f32(const float *d_partial, const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2,
                               float *d_scale, int N, int HW, int C, int S, int nblk, void *stream) {
    PT_CHECK(d_partial && d_w1 && d_b1 && d_w2 && d_b2 && d_scale && C <= 1024 && S <= 256 && N >= 1 && HW >= 1 && nblk >= 1,
             "ptocr_se_fc_f32: bad arguments (C <= 1024, S <= 256)");
    hipLaunchKernelGGL(se_fc_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, d_partial, d_w1, d_b1, d_w2, d_b2, d_scale, HW, C, S, nblk);
    return launch_ok("se_fc_kernel");
}
