// CRNN recognition kernels: BiLSTM recurrence, row softmax, CTC greedy (argmax + max-prob) reductions.
//
// Replaces (reference file:line):
//   nn.LSTM(bidirectional=True) of pytocr/modeling/necks/rnn.py:22,29-36   -> lstm_bidir_kernel (the input
//       projections x@W_ih^T + b_ih + b_hh are one MFMA GEMM per layer, ptocr_linear_f32)
//   F.softmax(predicts, dim=2) of pytocr/modeling/heads/rec_ctc_head.py:32-36 -> softmax_rows_kernel
//   preds.argmax(axis=2) / preds.max(axis=2) of pytocr/postprocess/rec_postprocess.py:83-84 -> ctc_greedy_kernel
//
// Layout: sequences are kept batch-major, row = b*T + t, so Im2Seq's permute (rnn.py:9-15) and the decode's
// transpose (rec_postprocess.py:82) disappear; only the drop-in `model(x)` contract re-creates [T,B,C].
//
// LSTM: the recurrence is independent per batch row, so a workgroup owns 16 batch rows of one direction for all T
// steps (no inter-workgroup synchronisation at all).  Per step it computes gates[16,1024] = h[16,256] @ W_hh^T with
// v_mfma_f32_16x16x4_f32: wave w owns hidden units [64w, 64w+64) of all four gates, so i,f,g,o of a unit meet in one
// lane and the cell update is register-local; h lives in LDS (double-buffered), c in registers, W_hh streams from L2.
#include "common.h"

namespace ptocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LH = 256;             // hidden size
constexpr int LROWS = 16;           // batch rows per workgroup
constexpr int HLD = LH + 4;         // LDS row stride of h (floats): conflict-free ds_read_b128 over 16 rows

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_bidir_kernel(const float *__restrict__ xproj, const float *__restrict__ whh,
                                                         float *__restrict__ out, int T, int B, long x_bytes) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][LROWS][HLD];
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * LROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jc = lane & 15, kq = lane >> 4;             // MFMA 16x16x4: column / k-quarter (A,B), rows 4*kq + r (C/D)
    const float *W = whh + (long)dir * 4 * LH * LH;

    for (int i = tid; i < 2 * LROWS * HLD; i += 256) (&hbuf[0][0][0])[i] = 0.f;
    float c[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) c[q][r] = 0.f;
    __syncthreads();

    // Buffer addressing keeps the address registers few: one lane offset per operand, the (gate, sub-tile) part in the
    // scalar offset, the k-block in the instruction's immediate.
    // weights: gate g, sub-tile q -> gate column g*256 + 64*wave + 16*q + jc, this lane reads 4 consecutive k
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, 4 * LH * LH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xproj), 0, (int)x_bytes, 0x00020000);
    const unsigned w_voff = (unsigned)(((64 * wave + jc) * LH + 4 * kq) * 4);
    auto load_w = [&](f32x4 (&w)[4][4], int kb) {
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                w[g][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, w_voff + (unsigned)(kb * 64), (g * LH + 16 * q) * LH * 4, 0));
    };
    // input projection: row (b, t, dir) of xproj[B][T][2][4H], element g*256 + 64*wave + 16*q + jc
    auto load_x = [&](f32x4 (&xp)[4][4], int step) {
        const int t = dir ? (T - 1 - step) : step;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int b = b0 + 4 * kq + r;
            b = b < B ? b : B - 1;
            const unsigned voff = (unsigned)((((long)b * T + t) * 2 + dir) * (4 * LH) + 64 * wave + jc) * 4u;
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    xp[g][q][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff + (unsigned)((g * LH + 16 * q) * 4), 0, 0));
        }
    };
    f32x4 w0[4][4], w1[4][4], xnext[4][4];
    load_x(xnext, 0);
    load_w(w0, 0);
    for (int step = 0; step < T; step++) {
        const int t = dir ? (T - 1 - step) : step;
        const int cur = step & 1;
        f32x4 acc[4][4];
        // accumulators start from the input projection of this time step
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[g][q] = xnext[g][q];
        if (step + 1 < T) load_x(xnext, step + 1);
        const float *hrow = &hbuf[cur][jc][4 * kq];       // A operand: row jc, k = 16*kb + 4*kq + t
#pragma unroll
        for (int kb = 0; kb < LH / 16; kb += 2) {
            load_w(w1, kb + 1);
            {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(hrow + 16 * kb);
#pragma unroll
                for (int tt = 0; tt < 4; tt++)             // k innermost-last: 16 independent accumulators between dependent MFMAs
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], w0[g][q][tt], acc[g][q], 0, 0, 0);
            }
            load_w(w0, (kb + 2) & (LH / 16 - 1));             // the last one fetches k-block 0 for the next time step
            {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(hrow + 16 * (kb + 1));
#pragma unroll
                for (int tt = 0; tt < 4; tt++)             // k innermost-last: 16 independent accumulators between dependent MFMAs
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            acc[g][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], w1[g][q][tt], acc[g][q], 0, 0, 0);
            }
        }
        // cell update: torch gate order i, f, g, o
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float ig = sigmoidf_(acc[0][q][r]);
                const float fg = sigmoidf_(acc[1][q][r]);
                const float gg = tanhf(acc[2][q][r]);
                const float og = sigmoidf_(acc[3][q][r]);
                c[q][r] = fg * c[q][r] + ig * gg;
                const float h = og * tanhf(c[q][r]);
                const int row = 4 * kq + r, unit = 64 * wave + 16 * q + jc;
                hbuf[cur ^ 1][row][unit] = h;
                const int b = b0 + row;
                if (b < B) out[((long)b * T + t) * (2 * LH) + dir * LH + unit] = h;
            }
        __syncthreads();
    }
}

// one wave per row: online max / sum-exp / first arg-max
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float *__restrict__ x, int rows, int C, int ld, int is_prob,
                                                         int *__restrict__ idx_out, float *__restrict__ prob_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + (long)row * ld;
    float m = -INFINITY, s = 0.f;
    int idx = 0x7fffffff;
    const int C4 = C & ~3;
    for (int i = lane * 4; i < C4; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + i);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float xv = v[k];
            if (xv > m) { s = s * __expf(m - xv) + 1.f; m = xv; idx = i + k; }
            else s += __expf(xv - m);
        }
    }
    for (int i = C4 + lane; i < C; i += 64) {
        const float xv = p[i];
        if (xv > m) { s = s * __expf(m - xv) + 1.f; m = xv; idx = i; }
        else s += __expf(xv - m);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        const int i2 = __shfl_xor(idx, o);
        const float mm = fmaxf(m, m2);
        const float sa = (m == -INFINITY) ? 0.f : s * __expf(m - mm);
        const float sb = (m2 == -INFINITY) ? 0.f : s2 * __expf(m2 - mm);
        idx = (m2 > m || (m2 == m && i2 < idx)) ? i2 : idx;
        m = mm; s = sa + sb;
    }
    if (lane == 0) {
        idx_out[row] = idx;
        prob_out[row] = is_prob ? m : 1.f / s;
    }
}

// row softmax, one wave per row (two passes; the row is L2-resident for the second)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ x, int rows, int C, int ld,
                                                           float *__restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + (long)row * ld;
    float m = -INFINITY;
    for (int i = lane; i < C; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += expf(p[i] - m);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    float *q = y + (long)row * ldy;
    for (int i = lane; i < C; i += 64) q[i] = expf(p[i] - m) / s;
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_lstm_bidir_f32(const float *d_xproj, const float *d_whh, float *d_out, int T, int B, int H, void *stream) {
    PT_CHECK(d_xproj && d_whh && d_out && T >= 1 && B >= 1, "ptocr_lstm_bidir_f32: bad arguments");
    PT_CHECK(H == LH, "ptocr_lstm_bidir_f32: hidden size must be %d (got %d)", LH, H);
    const long x_bytes = (long)B * T * 2 * 4 * LH * 4;
    PT_CHECK(x_bytes < (1L << 31), "ptocr_lstm_bidir_f32: B*T too large (projection tensor must stay below 2 GiB)");
    hipLaunchKernelGGL(lstm_bidir_kernel, dim3(cdiv(B, LROWS), 2), dim3(256), 0, (hipStream_t)stream, d_xproj, d_whh, d_out, T, B, x_bytes);
    return launch_ok("lstm_bidir_kernel");
}

extern "C" int ptocr_ctc_greedy_f32(const float *d_x, int rows, int C, int ld, int is_prob, int32_t *d_idx, float *d_prob,
                                    void *stream) {
    PT_CHECK(d_x && d_idx && d_prob && rows >= 1 && C >= 1 && ld >= C && ld % 4 == 0, "ptocr_ctc_greedy_f32: bad arguments (ld %% 4 == 0)");
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_x, rows, C, ld, is_prob, d_idx, d_prob);
    return launch_ok("ctc_greedy_kernel");
}

extern "C" int ptocr_softmax_rows_f32(const float *d_x, int rows, int C, int ld, float *d_y, int ldy, void *stream) {
    PT_CHECK(d_x && d_y && rows >= 1 && C >= 1 && ld >= C && ldy >= C, "ptocr_softmax_rows_f32: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, d_x, rows, C, ld, d_y, ldy);
    return launch_ok("softmax_rows_kernel");
}
