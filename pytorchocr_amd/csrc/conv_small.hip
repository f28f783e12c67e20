// CRNN's first layer, fused: conv 3x3 / s1 / p1 with 1..4 input channels -> 64 channels + bias + ReLU + 2x2/s2 max pool.
// Replaces conv0 + relu0 + pooling0 of rec_vgg.py:78-88 (ATen conv2d, relu_, max_pool2d).
//
// With K = 9 the layer has no matrix work to speak of (6.7 GFLOP for 512 lines) -- it is bound by writing its output, and
// the unfused form (implicit GEMM with the gray channel padded to 4, then the pool kernel) writes the 1.34 GB full-resolution
// tensor and reads it back: 0.94 ms.  Here a thread owns one POOLED pixel and 4 output channels: it reads the 4x4 input patch
// (16-byte pixels of the NHWC4 image, shared by the 16 lanes of the pixel), evaluates the four convolution outputs of the
// pool window on the VALU (weights in LDS) and stores 16 bytes; 16 lanes write one pixel's 256 contiguous bytes.
#include "common.h"

namespace ptocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// w: f32[Cin*9][64] ((ci*3 + ky)*3 + kx major, cout minor; BN folded), bias f32[64]
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_pool_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ bias, float *__restrict__ y,
                                                                 int N, int H, int W, int Hp, int Wp) {
    __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * 64];
    for (int i = threadIdx.x; i < CIN * 9 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    // grid = (pooled-row segments, pooled rows, images): no division per thread (the flat 64-bit index of the first version cost three
    // 64-bit divisions, as many instructions as the 36 multiply-adds x 4 channels of the thread)
    const int ix16 = blockIdx.x * 256 + threadIdx.x;
    const int cq = ix16 & 15, px = ix16 >> 4, py = blockIdx.y, n = blockIdx.z;
    if (px >= Wp) return;
    // 4x4 input patch: rows 2py-1 .. 2py+2, columns 2px-1 .. 2px+2 (zero outside the image); clamped address + select, so that the
    // sixteen loads are issued together instead of one per divergent block
    float in[CIN][4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int iy = 2 * py - 1 + r, ix = 2 * px - 1 + c;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const f32x4 t = *reinterpret_cast<const f32x4 *>(x + (((long)n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * 4);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v = ok ? t : z;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) in[ci][r][c] = v[ci];
        }
    f32x4 acc[2][2];
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + cq * 4);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = b4;
#pragma unroll
    for (int ci = 0; ci < CIN; ci++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(&ws[((ci * 3 + ky) * 3 + kx) * 64 + cq * 4]);
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] += wv * in[ci][a + ky][b + kx];
            }
    f32x4 m;
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = fmaxf(fmaxf(fmaxf(acc[0][0][k], acc[0][1][k]), fmaxf(acc[1][0][k], acc[1][1][k])), 0.f);
    *reinterpret_cast<f32x4 *>(y + (((long)n * Hp + py) * Wp + px) * 64 + cq * 4) = m;
}

// Round 4: SIXTEEN channels per thread (four threads per pooled pixel instead of sixteen) and the multiply / add pairs as packed fp32
// instructions on channel pairs.  The first form was bound by instruction issue: sixteen threads each loaded the same 4x4 input patch
// (sixteen 16-byte loads per thread) and spent 288 scalar v_mul / v_add on their four channels -- 301 us for the CRNN's first layer
// (512 lines), whose output is 336 MB.  Same products, same order, still a separate multiply and add (the file is built with
// -ffp-contract=off and the oracle tolerances were set on that): bit-identical outputs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CS_WIDE
#define CS_WIDE 1
#endif
#ifndef CS_ROWS
#define CS_ROWS 1           // pooled rows a thread of the wide form walks (4, with the next row prefetched: 170 registers, 258 us against 202)
#endif
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_pool16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                   const float *__restrict__ bias, float *__restrict__ y,
                                                                   int N, int H, int W, int Hp, int Wp) {
    __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * 64];
    for (int i = threadIdx.x; i < CIN * 9 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    // (CS_ROWS > 1: a thread walks several pooled rows of its column with the next patch requested before this one is multiplied -- an
    // experiment: the kernel's time is the SUM of its loads, arithmetic and stores (77 us of packed arithmetic, ~84 of stores, ~35 of
    // loads), but four rows per thread cost more in occupancy than the overlap returned)
    const int ix4 = blockIdx.x * 256 + threadIdx.x;
    const int c16 = ix4 & 3, px = ix4 >> 2, py0 = blockIdx.y * CS_ROWS, n = blockIdx.z;
    if (px >= Wp) return;
    float inb[2][CIN][4][4];
    auto gload = [&](int py, float (&in)[CIN][4][4]) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int iy = 2 * py - 1 + r, ix = 2 * px - 1 + c;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const f32x4 t = *reinterpret_cast<const f32x4 *>(x + (((long)n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * 4);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v = ok ? t : z;
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) in[ci][r][c] = v[ci];
            }
    };
    // the thread's sixteen channels are the quads c16, c16 + 4, c16 + 8, c16 + 12: store q4 of the four threads of a pixel is then 64
    // contiguous bytes (consecutive quads) instead of four 16-byte pieces 64 bytes apart
    f32x4 bq[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) bq[q4] = *reinterpret_cast<const f32x4 *>(bias + (c16 + 4 * q4) * 4);
    auto compute = [&](int py, const float (&in)[CIN][4][4]) {
        f32x2 acc[2][2][8];                                      // [a][b][2 q4 + pair of the quad]
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) { acc[a][b][2 * q4] = f32x2{bq[q4][0], bq[q4][1]}; acc[a][b][2 * q4 + 1] = f32x2{bq[q4][2], bq[q4][3]}; }
#pragma unroll
        for (int ci = 0; ci < CIN; ci++)
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const float *wp = &ws[((ci * 3 + ky) * 3 + kx) * 64 + c16 * 4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(wp + 16 * q4);
                        const f32x2 w0 = {wv[0], wv[1]}, w1 = {wv[2], wv[3]};
#pragma unroll
                        for (int a = 0; a < 2; a++)
#pragma unroll
                            for (int b = 0; b < 2; b++) {
                                const float v = in[ci][a + ky][b + kx];
                                const f32x2 vv = {v, v};
                                acc[a][b][2 * q4] = acc[a][b][2 * q4] + w0 * vv;          // v_pk_mul_f32 + v_pk_add_f32: two roundings, as the scalar form
                                acc[a][b][2 * q4 + 1] = acc[a][b][2 * q4 + 1] + w1 * vv;
                            }
                    }
                }
        float *dst = y + (((long)n * Hp + py) * Wp + px) * 64 + c16 * 4;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            f32x4 m;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int pr = 2 * q4 + (k >> 1), e = k & 1;
                m[k] = fmaxf(fmaxf(fmaxf(acc[0][0][pr][e], acc[0][1][pr][e]), fmaxf(acc[1][0][pr][e], acc[1][1][pr][e])), 0.f);
            }
            *reinterpret_cast<f32x4 *>(dst + 16 * q4) = m;
        }
    };
    gload(py0, inb[0]);
#pragma unroll
    for (int it = 0; it < CS_ROWS; it++) {
        const int py = py0 + it;
        if (py >= Hp) break;                                     // block-uniform
        if (it + 1 < CS_ROWS) gload(min(py + 1, Hp - 1), inb[(it + 1) & 1]);
        compute(py, inb[it & 1]);
    }
}

}  // namespace ptocr

using namespace ptocr;

// d_x: f32[N,H,W,4] (channels >= Cin ignored); d_w: f32[Cin*9][64], row (ci*3 + ky)*3 + kx; d_y: f32[N,H/2,W/2,64] =
// maxpool2x2(relu(conv3x3(x) + bias)).  Cin in 1..4.
extern "C" int ptocr_conv3x3_small_relu_pool_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                                 int Cin, void *stream) {
    PT_CHECK(d_x && d_w && d_bias && d_y, "ptocr_conv3x3_small_relu_pool_f32: null argument");
    PT_CHECK(N > 0 && H >= 2 && W >= 2 && Cin >= 1 && Cin <= 4, "ptocr_conv3x3_small_relu_pool_f32: need H, W >= 2 and 1 <= Cin <= 4");
    const int Hp = H / 2, Wp = W / 2;
    PT_CHECK(N <= 65535 && Hp <= 65535, "ptocr_conv3x3_small_relu_pool_f32: batch or pooled height > 65535");
    hipStream_t s = (hipStream_t)stream;
    static const bool wide = CS_WIDE && !(getenv("PTOCR_CONV_SMALL_WIDE") && atoi(getenv("PTOCR_CONV_SMALL_WIDE")) == 0);
    if (wide) {
        const dim3 grid4((unsigned)((Wp * 4 + 255) / 256), (unsigned)((Hp + CS_ROWS - 1) / CS_ROWS), (unsigned)N);
        switch (Cin) {
            case 1: hipLaunchKernelGGL(conv3x3_small_pool16_kernel<1>, grid4, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
            case 2: hipLaunchKernelGGL(conv3x3_small_pool16_kernel<2>, grid4, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
            case 3: hipLaunchKernelGGL(conv3x3_small_pool16_kernel<3>, grid4, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
            default: hipLaunchKernelGGL(conv3x3_small_pool16_kernel<4>, grid4, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
        }
        return launch_ok("conv3x3_small_pool16_kernel");
    }
    const dim3 grid((unsigned)((Wp * 16 + 255) / 256), (unsigned)Hp, (unsigned)N);
    switch (Cin) {
        case 1: hipLaunchKernelGGL(conv3x3_small_pool_kernel<1>, grid, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
        case 2: hipLaunchKernelGGL(conv3x3_small_pool_kernel<2>, grid, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
        case 3: hipLaunchKernelGGL(conv3x3_small_pool_kernel<3>, grid, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
        default: hipLaunchKernelGGL(conv3x3_small_pool_kernel<4>, grid, dim3(256), 0, s, d_x, d_w, d_bias, d_y, N, H, W, Hp, Wp); break;
    }
    return launch_ok("conv3x3_small_pool_kernel");
}
