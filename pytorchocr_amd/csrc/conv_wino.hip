// Winograd F(2x2, 3x3) convolution on fp32 MFMA for the 3x3 / stride 1 / pad 1 layers (85 % of DBNet-r18's FLOPs).
//
// Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A : 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e. 2.25x
// fewer MFMAs than the direct implicit GEMM, at fp32 accuracy (transform constants are 0, +-1, +-1/2).
// Replaces the same ATen conv2d calls as conv_mfma.hip (det_resnet.py:66-82, fpn.py:59-82, det_db_head.py:10).
//
// Work item ("patch"): 64 Winograd tiles (TXN x TYN tiles = 2TXN x 2TYN outputs; 16x16 or, for short images, 32 rows x 8
// columns) of one image x 64 output channels, one 512-thread workgroup per patch (8 waves = two per SIMD, 1 workgroup/CU).
// Consecutive blockIdx share the output-channel block, so the weights stay in the XCD's L2.
// The 16 "frequencies" xi = (i, j) are 16 independent GEMMs [64 tiles x Cin] x [Cin x 64]; wave w owns xi = 2w, 2w+1:
// 2 xi x (2x2 MFMA tiles of 32x32) = 128 accumulator registers.
// K loop in chunks of 4 channels (two v_mfma_f32_32x32x2_f32 per tile pair): per chunk each wave issues 16 MFMAs and a
// 1/8 share of the side work: (a) input transform of the next chunk (B^T d B from an LDS copy of the raw patch, 16 channels
// deep; a thread pair per (tile, channel), each half producing two of the four output rows), (b) copy of the next chunk of
// pre-transformed weights U (packed contiguously on the host) to LDS, (c) every fourth chunk the refill of the raw patch.
// V, U and the raw patch are all double-buffered in LDS (~145 KB).
// Measured on MI355X (tools/wino_timing.py, s_memtime probes): fp32 MFMAs do not overlap the same SIMD's VALU / LDS-store /
// VMEM issue, whichever wave issues it -- a chunk costs 2048 MFMA cycles + ~500 cycles of side work (2560 measured; 2060
// with the side work compiled out) -- so the side work is kept minimal rather than "hidden"; two waves per SIMD mainly
// shorten the output transform and the barrier skew (4 waves/workgroup: 2740 cycles per chunk, 14.7k-cycle epilogue; 8
// waves: 2560 and 8.9k).
// Output transform: wave w reduces its two M_ij over j in registers (its part of T_ib = sum_j M_ij A_jb); the sum over the
// eight partial tiles (A^T over i) goes through LDS, two passes (b = 0, 1), followed by bias / residual / ReLU and 16-byte
// channel-contiguous stores (optionally replicated up x up: the FPN's nearest upsample into the concat buffer).
#include "common.h"
#include <cstdlib>

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WLD = 6;                     // LDS row stride (floats) of V / U rows holding 4 channels (8-byte aligned, conflict-free)
constexpr int W_V = 16 * 64 * WLD;         // floats per V (or U) buffer
constexpr int WPX = 18;                    // LDS pixel stride (floats) of the raw patch: 16 channels + 2, so that the transform's
                                           // ds_read_b32 (banks mod 32, 32-lane groups) of 8 tiles x 4 channels is conflict-free
constexpr int W_EL = 68;                   // exchange tile row stride
// patch geometry: TN images x TYN x TXN tiles = 64 tiles; raw patch = TN sub-patches of (2 TYN + 2) x (2 TXN + 2) pixels
constexpr int wino_raw_floats(int txn, int tyn) { return ((64 / (txn * tyn)) * (2 * txn + 2) * (2 * tyn + 2) + 1) * WPX + 2; }   // + one dump pixel

struct WinoArgs {
    const float *x, *u, *bias, *res;
    float *y;
    int N, H, W, Cin, Cout;                // stride 1, pad 1: output H x W
    int tiles_x, tiles_y;                  // patches per image group (TN images)
    int relu, res_mode, out_ldc, out_coff, res_ldc, up;
    int cout_store;                        // output channels actually stored (weights may be zero-padded to a multiple of 64)
    int total;                             // patches x image groups x (Cout / 64)
    long x_bytes, u_bytes;
    unsigned long long *dbg;               // timing probe (ptocr_wino_set_timing_buffer): 4 clock samples per workgroup, or null
};

template <int TXN, int TYN, int DBG = 0>
__global__ __launch_bounds__(512) void conv_wino_kernel(WinoArgs p) {
    constexpr int TN = 64 / (TXN * TYN);            // images per patch (short images: several images share a patch)
    constexpr int PW = 2 * TXN + 2, PH = 2 * TYN + 2, NPX = TN * PW * PH;
    constexpr int W_RAW = wino_raw_floats(TXN, TYN);
    constexpr int NPIECE = (NPX * 4 + 511) / 512;   // 16-byte pieces of the raw patch per thread (3 or 4)
    static_assert(TN * TXN * TYN == 64 && NPIECE <= 4, "unsupported patch geometry");
    static_assert(4 * W_V + 2 * W_RAW <= 40960, "LDS budget (160 KB)");
    static_assert(8 * 64 * W_EL <= 4 * W_V + 2 * W_RAW, "exchange tiles must fit the LDS allocation");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Vb = smem;                       // [2][16][64][WLD]
    float *Ub = smem + 2 * W_V;             // [2][16][64][WLD]
    float *Rb = smem + 4 * W_V;             // [2][NPX + 1][WPX]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7: two waves per SIMD
    const int patches = p.tiles_x * p.tiles_y;
    const int per_cb = ((p.N + TN - 1) / TN) * patches;
    const int nS = p.Cin >> 4;              // super-steps of 16 channels
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.u_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;             // tensors are < 2 GiB: stays out of range after adding a channel offset

    // ---- patch decode (uniform) + raw patch loader: NPX pixels x 4 float4 (16 channels); thread handles pieces f = tid + 512 r
    const int id = blockIdx.x;
    const int cb = id / per_cb, rem = id - cb * per_cb;
    const int n_base = (rem / patches) * TN;           // first image of the patch
    const int pr = rem - (rem / patches) * patches;
    const int pty = pr / p.tiles_x, ptx = pr - pty * p.tiles_x;
    const int oy0 = pty * (2 * TYN), ox0 = ptx * (2 * TXN), n0 = cb * 64;
    const unsigned u_base = (unsigned)cb * (unsigned)(p.Cin >> 2) * 16384u;
    unsigned r_off[NPIECE];                 // byte offset of the piece for channel block 0
    unsigned r_valid = 0;                   // bit r: piece r lies inside an image (else it reads as zeros)
#pragma unroll
    for (int r = 0; r < NPIECE; r++) {
        const int f = tid + 512 * r;
        const int px = f >> 2, cq = f & 3;
        const int img = px / (PW * PH), pq = px - img * (PW * PH);
        const int py = pq / PW, pxx = pq - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx, nn = n_base + img;
        const bool ok = px < NPX && nn < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        r_off[r] = ok ? (unsigned)((((nn * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : 0u;
        r_valid |= (unsigned)ok << r;
    }
    // Loads past the last channel block / chunk are not predicated: they read the neighbouring pixel's channels or the next
    // weight block (or zeros beyond the buffer) into LDS buffers that are never consumed.
    f32x4 rreg[NPIECE];
    auto raw_gload1 = [&](int S, int r) {
        const unsigned off = ((r_valid >> r) & 1u) ? r_off[r] + (unsigned)(S * 64) : oob;
        rreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    };
    auto raw_lstore1 = [&](int buf, int r) {
        const int f = tid + 512 * r;
        const int px = f >> 2 < NPX ? f >> 2 : NPX;                                // pieces beyond the patch land in the dump pixel
        float *d = Rb + buf * W_RAW + px * WPX + (f & 3) * 4;
        *reinterpret_cast<f32x2 *>(d) = f32x2{rreg[r][0], rreg[r][1]};
        *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rreg[r][2], rreg[r][3]};
    };
    // ---- U loader: chunk = [16 xi][64 cout][4] floats contiguous (16 KB); thread handles float4 f = tid + 512 r
    f32x4 ureg[2];
    auto u_gload1 = [&](int chunk, int r) {
        const unsigned off = u_base + (unsigned)(chunk * 16384 + (tid + 512 * r) * 16);
        ureg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, off, 0, 0));
    };
    auto u_lstore1 = [&](int buf, int r) {
        float *d = Ub + buf * W_V + (tid + 512 * r) * WLD;
        *reinterpret_cast<f32x2 *>(d) = f32x2{ureg[r][0], ureg[r][1]};
        *reinterpret_cast<f32x2 *>(d + 2) = f32x2{ureg[r][2], ureg[r][3]};
    };
    // ---- input transform B^T d B of one (tile, channel) item per thread pair: tile 16 (w & 3) + 8 (lane >> 5) + (lane & 7),
    // channel (lane >> 3) & 3 of the chunk; waves 0-3 produce output rows a = 0, 1 (from patch rows 0..2), waves 4-7 rows
    // a = 3, 2 (from patch rows 1..3).  With B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] and k0..k2 the three rows read:
    //   first half:  row0 = k0 - k2 (a=0), row1 = k1 + k2 (a=1);   second half: row0 = k0 - k2 (a=3), row1 = k1 - k0 (a=2)
    // i.e. row1 = k1 + sgn * (half ? k0 : k2): one select + one exact fma, no divergent code.
    const int th = wave >> 2;
    const int tt = (wave & 3) * 16 + (lane >> 5) * 8 + (lane & 7);
    const int tch = (lane >> 3) & 3;
    const int t_img = tt / (TXN * TYN), t_ty = (tt / TXN) % TYN, t_tx = tt % TXN;
    const int t_roff = ((t_img * PH + 2 * t_ty + th) * PW + 2 * t_tx) * WPX + tch;
    const int t_voff0 = ((th ? 3 : 0) * 4 * 64 + tt) * WLD + tch;          // V row block of row0
    const int t_voff1 = ((th ? 2 : 1) * 4 * 64 + tt) * WLD + tch;          // V row block of row1
    const float t_sgn = th ? -1.f : 1.f;
    float td[3][4], tq[2][4];
    auto tr_read = [&](const float *rp, int k, int b) { td[k][b] = rp[(k * PW + b) * WPX]; };
    auto tr_rows = [&]() {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            tq[0][b] = td[0][b] - td[2][b];
            tq[1][b] = __builtin_fmaf(t_sgn, th ? td[0][b] : td[2][b], td[1][b]);
        }
    };
    auto tr_cols = [&](float *vp, int a) {
        vp[0 * 64 * WLD] = tq[a][0] - tq[a][2];
        vp[1 * 64 * WLD] = tq[a][1] + tq[a][2];
        vp[2 * 64 * WLD] = tq[a][2] - tq[a][1];
        vp[3 * 64 * WLD] = tq[a][1] - tq[a][3];
    };
    auto raw_ptr = [&](int chunk) { return Rb + ((chunk >> 2) & 1) * W_RAW + t_roff + (chunk & 3) * 4; };

    // ---- MFMA: wave w owns frequencies xi = 2w, 2w+1: 2 xi x (2x2 MFMA tiles of 32x32) = 128 accumulator registers.
    // Operand fragments in two register sets: set (chunk & 1) is read right after the barrier that publishes the chunk's
    // V / U, while the last MFMAs of the previous chunk are still executing.
    const int frow = lane & 31, fh = lane >> 5;
    const int f_off = (wave * 2 * 64 + frow) * WLD + 2 * fh;      // xi = 2w + e, row frow, k = 2h + t
    f32x16 acc[2][2][2];
    f32x2 fa0[2][2], fa1[2][2], fb0[2][2], fb1[2][2];
    auto frag_load = [&](int set, int buf) {
        const float *va = Vb + buf * W_V + f_off;
        const float *ub = Ub + buf * W_V + f_off;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            fa0[set][e] = *reinterpret_cast<const f32x2 *>(va + e * 64 * WLD);
            fa1[set][e] = *reinterpret_cast<const f32x2 *>(va + e * 64 * WLD + 32 * WLD);
            fb0[set][e] = *reinterpret_cast<const f32x2 *>(ub + e * 64 * WLD);
            fb1[set][e] = *reinterpret_cast<const f32x2 *>(ub + e * 64 * WLD + 32 * WLD);
        }
    };
    auto mfma_g = [&](int set, int g) {                        // MFMA g of 16 of a chunk: xi e, k step t, tile (ma, nb)
        const int e = g >> 3, t = (g >> 2) & 1, ma = (g >> 1) & 1, nb = g & 1;
        acc[e][ma][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma ? fa1[set][e][t] : fa0[set][e][t],
                                                               nb ? fb1[set][e][t] : fb0[set][e][t], acc[e][ma][nb], 0, 0, 0);
    };

    // ---- head: raw patch of super-step 0, weights and transform of chunk 0, weights of chunk 1 in flight
#pragma unroll
    for (int r = 0; r < NPIECE; r++) raw_gload1(0, r);
#pragma unroll
    for (int r = 0; r < 2; r++) u_gload1(0, r);
#pragma unroll
    for (int r = 0; r < NPIECE; r++) raw_lstore1(0, r);
#pragma unroll
    for (int r = 0; r < 2; r++) { u_lstore1(0, r); u_gload1(1, r); }
    __syncthreads();
    {
        const float *rp = raw_ptr(0);
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) tr_read(rp, k, b);
        tr_rows();
        tr_cols(Vb + t_voff0, 0);
        tr_cols(Vb + t_voff1, 1);
    }
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[e][a][b][r] = 0.f;
    __syncthreads();
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
    frag_load(0, 0);

    for (int S = 0; S < nS; S++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int chunk = 4 * S + q;
            const int cur = q & 1, nxt = cur ^ 1;                 // chunk parity == q parity
            const float *rp = raw_ptr(chunk + 1);
            // 12 slots of one MFMA plus a share of the side work for chunk+1/+2, in program order (sched_barrier pins it).
            // A wave's own VALU / LDS-write / VMEM instructions do not overlap its MFMAs on this hardware; they run in the
            // shadow of the SIMD's other wave, so the slotting only has to keep load -> use distances long:
            //   0-1   weights of chunk+1 (loaded a chunk ago) to LDS, weights of chunk+2 into the same registers
            //   0-3   raw patch refill for the next 16 channels (load at q=0, LDS store at q=1)
            //   2-7   input transform: LDS reads;  8 row pass;  9-10 column pass + LDS writes
#pragma unroll
            for (int g = 0; g < 12; g++) {
                mfma_g(cur, g);
                if (g < 2 && !(DBG & 2)) { u_lstore1(nxt, g); u_gload1(chunk + 2, g); }
                if (g < NPIECE && !(DBG & 2)) {
                    if (q == 0) raw_gload1(S + 1, g);
                    if (q == 1) raw_lstore1((S + 1) & 1, g);
                }
                if (!(DBG & 1)) {
                    if (g >= 2 && g < 8) { tr_read(rp, (g - 2) >> 1, 2 * (g & 1)); tr_read(rp, (g - 2) >> 1, 2 * (g & 1) + 1); }
                    if (g == 8) tr_rows();
                    if (g == 9) tr_cols(Vb + nxt * W_V + t_voff0, 0);
                    if (g == 10) tr_cols(Vb + nxt * W_V + t_voff1, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(DBG & 4)) __syncthreads();
            // chunk+1 is published: fetch its fragments under the last four MFMAs of this chunk
            frag_load(nxt, nxt);
#pragma unroll
            for (int g = 12; g < 16; g++) mfma_g(cur, g);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                            // every wave is done with V / U: the exchange tiles reuse them
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Wave w holds M_ij for i = w >> 1, j = 2 (w & 1) + {0, 1}.
    // Column pass in registers: its part of T_ib = sum_j M_ij A_jb is  b=0: M_i0 + M_i1 | M_i2 ;  b=1: M_i1 | -M_i2 - M_i3.
    // The eight partial tiles go through LDS; Y_0b = sum of the partials of waves 0..5, Y_1b = (w2 + w3) - (w4 + w5 + w6 + w7).
    float *ex = smem;                                           // [8 waves][64 tiles][W_EL]
    const int HW = p.H * p.W;
    const int up = p.up;
    const int jh = wave & 1;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        // residual rows for this pass, requested before the exchange so that their latency hides behind it
        f32x4 rres[2][2];
        if (p.res_mode == PTOCR_RES_ADD_PRE_RELU) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int item = tid + 512 * it;
                const int tile = item >> 4, cq = item & 15;
                const int ox = ox0 + 2 * (tile % TXN) + b;
                const int n = n_base + tile / (TXN * TYN);
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int oy = oy0 + 2 * ((tile / TXN) % TYN) + a;
                    const bool ok = oy < p.H && ox < p.W && n < p.N;
                    const long m = ok ? (long)n * HW + (long)oy * p.W + ox : 0;
                    rres[it][a] = *reinterpret_cast<const f32x4 *>(p.res + m * p.res_ldc + n0 + cq * 4);
                }
            }
        }
        const float c0 = b == 0 ? 1.f : (jh ? -1.f : 0.f);      // partial = c0 * acc[0] + c1 * acc[1] (exact: coefficients 0, +-1)
        const float c1 = b == 0 ? (jh ? 0.f : 1.f) : (jh ? -1.f : 1.f);
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float tv = __builtin_fmaf(c1, acc[1][mt][nt][r], c0 * acc[0][mt][nt][r]);
                    const int tile = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    ex[(wave * 64 + tile) * W_EL + nt * 32 + frow] = tv;
                }
                __builtin_amdgcn_sched_barrier(0);               // keeps the accumulator reads from piling up in VGPRs
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int item = tid + 512 * it;                    // items = 64 tiles x 16 channel quads
            const int tile = item >> 4, cq = item & 15;
            f32x4 t[8];
#pragma unroll
            for (int w = 0; w < 8; w++) t[w] = *reinterpret_cast<const f32x4 *>(ex + (w * 64 + tile) * W_EL + cq * 4);
            const int col = n0 + cq * 4;
            if (col >= p.cout_store) continue;                  // channels of the zero padding
            const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(p.bias + col);
            const f32x4 mid = t[2] + t[3];
            const f32x4 y0 = ((t[0] + t[1]) + mid) + (t[4] + t[5]);
            const f32x4 y1 = (mid - (t[4] + t[5])) - (t[6] + t[7]);
            const int ox = ox0 + 2 * (tile % TXN) + b;
            const int n = n_base + tile / (TXN * TYN);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const int oy = oy0 + 2 * ((tile / TXN) % TYN) + a;
                if (oy >= p.H || ox >= p.W || n >= p.N) continue;
                f32x4 v = (a == 0 ? y0 : y1) + bias4;
                if (p.res_mode == PTOCR_RES_ADD_PRE_RELU) v += rres[it][a];
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (up == 1) {
                    const long m = (long)n * HW + (long)oy * p.W + ox;
                    *reinterpret_cast<f32x4 *>(p.y + m * p.out_ldc + p.out_coff + col) = v;
                } else {                                        // nearest upsample: up x up replicas
                    const long Wu = (long)p.W * up;
                    float *yb = p.y + (((long)n * p.H * up + (long)oy * up) * Wu + (long)ox * up) * p.out_ldc + p.out_coff + col;
                    for (int dy = 0; dy < up; dy++)
                        for (int dx = 0; dx < up; dx++) *reinterpret_cast<f32x4 *>(yb + (dy * Wu + dx) * p.out_ldc) = v;
                }
            }
        }
        __syncthreads();
    }
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}

template <int TXN, int TYN, int DBG = 0>
static int launch_wino(WinoArgs a, hipStream_t stream) {
    constexpr int TN = 64 / (TXN * TYN);
    a.tiles_x = cdiv(a.W, 2 * TXN); a.tiles_y = cdiv(a.H, 2 * TYN);
    const long total = (long)cdiv(a.N, TN) * a.tiles_x * a.tiles_y * (a.Cout / 64);
    PT_CHECK(total < (1L << 31), "ptocr_conv3x3_wino_f32: too many patches");
    a.total = (int)total;
    const size_t lds = sizeof(float) * (4 * W_V + 2 * wino_raw_floats(TXN, TYN));
    static DynLds dyn;
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&conv_wino_kernel<TXN, TYN, DBG>), (int)lds)) return e_;
    hipLaunchKernelGGL((conv_wino_kernel<TXN, TYN, DBG>), dim3((unsigned)a.total), dim3(512), lds, stream, a);
    return launch_ok("conv_wino_kernel");
}

}  // namespace ptocr

using namespace ptocr;

static unsigned long long *g_wino_dbg = nullptr;
// debug: device buffer of 4 clock samples per workgroup (start, main loop start, main loop end, end); null switches it off
extern "C" void ptocr_wino_set_timing_buffer(void *d_buf) { g_wino_dbg = (unsigned long long *)d_buf; }

// d_u: weights transformed on the host, packed f32[Cout/64][Cin/4][16][64][4] (U = G g G^T per (cout, cin), BN folded); Cout may be
// zero-padded to a multiple of 64, then only channels < cout_store (a multiple of 4; 0 = Cout) are stored and d_bias has Cout entries.
// up > 1 writes every output pixel up x up times (nearest upsample) into y[N][H*up][W*up][out_ldc].
extern "C" int ptocr_conv3x3_wino_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                      int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                                      int out_ldc, int out_coff, int up, void *stream) {
    PT_CHECK(d_x && d_u && d_bias && d_y, "ptocr_conv3x3_wino_f32: null argument");
    PT_CHECK(N > 0 && H > 0 && W > 0, "ptocr_conv3x3_wino_f32: empty tensor");
    PT_CHECK(Cin % 16 == 0 && Cout % 64 == 0, "ptocr_conv3x3_wino_f32: need Cin %% 16 == 0 and Cout %% 64 == 0");
    PT_CHECK(relu == 0 || relu == 1, "ptocr_conv3x3_wino_f32: activation must be none or ReLU");
    PT_CHECK(res_mode == PTOCR_RES_NONE || (res_mode == PTOCR_RES_ADD_PRE_RELU && d_res), "ptocr_conv3x3_wino_f32: only the pre-ReLU residual add is fused");
    PT_CHECK(up >= 1 && up <= 8 && (up == 1 || res_mode == PTOCR_RES_NONE), "ptocr_conv3x3_wino_f32: up must be 1..8 and excludes the residual");
    if (cout_store <= 0) cout_store = Cout;
    PT_CHECK(cout_store <= Cout && cout_store % 4 == 0, "ptocr_conv3x3_wino_f32: cout_store must be a multiple of 4 and <= Cout");
    PT_CHECK(out_ldc % 4 == 0 && out_coff % 4 == 0 && out_ldc >= out_coff + cout_store && (res_mode == 0 || res_ldc % 4 == 0), "ptocr_conv3x3_wino_f32: channel strides must be multiples of 4");
    WinoArgs a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.relu = relu; a.res_mode = res_mode; a.out_ldc = out_ldc; a.out_coff = out_coff; a.res_ldc = res_ldc > 0 ? res_ldc : Cout;
    a.up = up; a.cout_store = cout_store;
    a.dbg = g_wino_dbg;
    a.x_bytes = (long)N * H * W * Cin * 4;
    a.u_bytes = (long)Cout * Cin * 16 * 4;
    PT_CHECK(a.x_bytes < (1L << 31) && a.u_bytes < (1L << 31), "ptocr_conv3x3_wino_f32: tensor larger than 2 GiB");
    // patch geometry (64 tiles): 16x16 outputs, 32 rows x 8 columns, or -- for short images (text lines) and small maps -- 8x16
    // outputs of 2 images / 4x16 or 8x8 outputs of 4 images; the one that covers the batch at the lowest cost wins (multi-image patches carry a
    // larger halo and one more raw piece per thread: measured 6 % / 12 % more time per patch)
    const long cnt[5] = {100 * (long)N * cdiv(H, 16) * cdiv(W, 16), 100 * (long)N * cdiv(H, 32) * cdiv(W, 8),
                         106 * (long)cdiv(N, 2) * cdiv(H, 8) * cdiv(W, 16), 112 * (long)cdiv(N, 4) * cdiv(H, 4) * cdiv(W, 16),
                         112 * (long)cdiv(N, 4) * cdiv(H, 8) * cdiv(W, 8)};
    int geo = 0;
    for (int g = 1; g < 5; g++)
        if (cnt[g] < cnt[geo]) geo = g;
#ifdef PTOCR_WINO_EXPERIMENT
    static const int dbgm = getenv("PTOCR_WINO_DBG") ? atoi(getenv("PTOCR_WINO_DBG")) : 0;
    if (geo == 0) switch (dbgm) {
        case 1: return launch_wino<8, 8, 1>(a, (hipStream_t)stream);
        case 2: return launch_wino<8, 8, 2>(a, (hipStream_t)stream);
        case 3: return launch_wino<8, 8, 3>(a, (hipStream_t)stream);
        case 4: return launch_wino<8, 8, 4>(a, (hipStream_t)stream);
        default: break;
    }
#endif
    switch (geo) {
        case 1: return launch_wino<4, 16>(a, (hipStream_t)stream);
        case 2: return launch_wino<8, 4>(a, (hipStream_t)stream);
        case 3: return launch_wino<8, 2>(a, (hipStream_t)stream);
        case 4: return launch_wino<4, 4>(a, (hipStream_t)stream);
        default: return launch_wino<8, 8>(a, (hipStream_t)stream);
    }
}
