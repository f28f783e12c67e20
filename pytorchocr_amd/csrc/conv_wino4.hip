// Winograd F(4x4, 3x3) convolution on fp32 MFMA for the large 3x3 / stride 1 / pad 1 layers.
//
// Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A with 6x6 transform tiles: 36 multiplies per 4x4 output tile and channel pair
// instead of 144, i.e. 4x fewer MFMAs than the direct implicit GEMM and 1.78x fewer than F(2x2, 3x3) (conv_wino.hip).  The
// transform constants (B: 0, +-1, +-2, +-4, +-5; A: 0, +-1, 2, 4, 8; G folded into the weights in fp64 on the host) cost
// accuracy: measured through the whole DBNet-r18 the probability maps move by 1e-6 and the features by 2e-6 of their
// maximum (bar 1e-4).  Replaces the same ATen conv2d calls as conv_wino.hip (det_resnet.py:66-82, fpn.py:59-82,
// det_db_head.py:10); the host side picks per layer whichever of the two kernels covers the map at the lower cost.
//
// Work item ("patch"): 32 Winograd tiles (TN images x TYN x TXN tiles, 4x4 outputs each) x 64 output channels, one
// 768-thread workgroup (12 waves = three per SIMD, 1 workgroup/CU).  The 36 "frequencies" xi = (i, j) are 36 independent
// GEMMs [32 tiles x Cin] x [Cin x 64]; wave w owns xi = 3w .. 3w+2: 3 xi x (1x2 MFMA tiles of 32x32) = 96 accumulators.
// K runs in chunks of 4 channels.  Per chunk a wave issues 12 v_mfma_f32_32x32x2_f32 and its 1/12 share of the side work:
//   * input transform of the next chunk: thread = (tile, channel, output row a) -- 128 items x 6 rows = 768 threads, the
//     row index is uniform per wave, so the row formulas are scalar-selected coefficients, no divergence; 24 ds_read_b32
//     from the LDS copy of the raw patch (pixel stride 17 floats: 8 tiles x 4 channels of a 32-lane group hit 32 banks),
//     row pass, column pass, 6 ds_write_b32 into V (laid out [xi][k pair][tile][2] so that the MFMA A fragment of a lane
//     is one aligned 8-byte read, 64 lanes contiguous);
//   * the weight (B) fragments come STRAIGHT FROM GLOBAL MEMORY into registers: every U element is consumed by exactly
//     one wave, so staging U in LDS would only add LDS writes; the host packs U so that a wave's fragments of a chunk are
//     three contiguous 1 KB rows (one 16-byte load per lane each), prefetched one chunk ahead;
//   * every fourth chunk the refill of the raw patch (16 channels deep, double-buffered).
// Output transform: the j-sum over a wave's three frequencies in registers (one of the four output columns b per pass),
// the i-sum over the twelve partial tiles through LDS, then bias / residual / ReLU and 16-byte channel-contiguous stores
// (optionally replicated up x up: the FPN's nearest upsample into the concat buffer).
#include "common.h"

#ifndef W4_RAW_SLOT
#define W4_RAW_SLOT 0       // MFMA slot of a chunk at which the raw-patch refill (global loads / LDS stores of two pieces) is issued
#endif
#ifndef W4_PRIO
#define W4_PRIO 0            // experiment, OFF: wave priority raised around every MFMA issue.  Nothing alone on the chip (3 058 vs 3 074 cycles per chunk),
                            // +0.35 % on the headline where the post-process of the previous batch shares the CUs (2 104 / 2 112 -> 2 115 / 2 117
                            // images/s) -- but the first full default bench run with it ended in a GPU memory access fault that no run before
                            // it (and none of the test suites) had shown: it changes which kernel's waves run when several share a CU, and
                            // was withdrawn rather than trusted
#endif
#ifndef W4_RESPF
#define W4_RESPF 1          // 1: the residual rows of an output pass are requested one pass ahead
#endif
#ifndef W4_DBG
#define W4_DBG 0            // timing experiments only (PTOCR_EXTRA_HIPCC_FLAGS=-DW4_DBG=n): 1 no global stores, 2 no consumer, 4 no exchange writes,
                            // 8 no input transform, 16 no weight-fragment loads, 32 no raw refill, 64 no transform math (LDS traffic kept)
#endif

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// (bf16(x), bf16(x)) in one dword, round to nearest even
__device__ __forceinline__ unsigned cvt_pk_same(float x) { const bf16x2_t v = {(__bf16)x, (__bf16)x}; return __builtin_bit_cast(unsigned, v); }

// packed fp32 VALU ops (two lanes of math per instruction); the compiler scalarises <2 x float> arithmetic whose halves come from
// separate LDS reads, so these are spelled out.  c: wave-uniform coefficient pair in SGPRs.
__device__ __forceinline__ f32x2 pk_mul_s(f32x2 c, f32x2 x) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(c), "v"(x)); return d; }
__device__ __forceinline__ f32x2 pk_fma_s(f32x2 c, f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(y)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y)); return d; }
// (p.hi + p.lo, p.hi - p.lo)
__device__ __forceinline__ f32x2 pk_hi_pm_lo(f32x2 p) { f32x2 d; asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p)); return d; }
// (p.hi + c p.lo, p.hi - c p.lo)
__device__ __forceinline__ f32x2 pk_hi_pm_clo(f32x2 c, f32x2 p) {
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(d) : "s"(c), "v"(p)); return d;
}

constexpr int W4_VH = 80;                      // floats per (xi, k pair) block of V: 32 tiles x 2 channels + 16 (second pair 16 banks off)
constexpr int W4_V = 36 * 2 * W4_VH;           // floats per V buffer
constexpr int W4_PX = 17;                      // LDS pixel stride (floats) of the raw patch: 16 channels + 1
constexpr int W4_EL = 68;                      // exchange tile row stride (72, which puts the two halves of a writing wave on disjoint banks, measures the same: 13.3 k cycles)
constexpr int W4_THREADS = 768;
// raw buffer: room for every 16-byte piece the 768 threads store (pieces past the patch are never read), so that piece r of a
// thread sits at a compile-time offset (192 pixels) behind its piece 0
constexpr int w4_raw_floats(int txn, int tyn, int tn) { return ((tn * (4 * txn + 2) * (4 * tyn + 2) * 4 + W4_THREADS - 1) / W4_THREADS) * 192 * W4_PX + 4; }

struct Wino4Args {
    const float *x, *u, *bias, *res;
    float *y;
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y;                      // patches per image group
    int relu, res_mode, out_ldc, out_coff, res_ldc, up;
    int cout_store;
    int pool;                                  // MODE 3: ReLU + MaxPool2d(2, 2) in the epilogue, y is [N, H/2, W/2, ldc]
    int total;
    long x_bytes, u_bytes, y_bytes, res_bytes;
    unsigned long long *dbg;
};

// MODE: 0 plain, 1 pre-ReLU residual add, 2 nearest-upsample replication, 3 ReLU + MaxPool2d(2, 2) (compile-time: the epilogue stays
// free of dead code).  MODE 3 (round 4, CRNN conv1 + pooling1, rec_vgg.py:33-35): a 4x4 output tile holds 2x2 whole pool windows (tile
// origins are multiples of four, H and W even), and a consumer thread's columns of the two passes of a channel half -- 2 c_bs and
// 2 c_bs + 1 -- are one window's: the passes run (even, odd) columns per channel half, the even pass parks its four rows in registers,
// the odd pass takes the maxima and stores two pooled rows.  The full-resolution tensor (671 MB for 512 lines) is never written.
// SPLIT (experiment, ptocr_conv3x3_wino4_split_f32): the 36 GEMMs on the bf16 matrix pipe with two-piece operands.  x = h + m,
// h = bf16(x), m = bf16(x - h) (16 mantissa bits in all); a product a b becomes a_h (b_h + b_m) + a_m (b_h + b_m), fp32 accumulate:
// two v_mfma_f32_32x32x8_bf16_1k (32 cycles each) per 4-channel chunk and accumulator tile instead of two v_mfma_f32_32x32x2_f32
// (64 cycles each).  A lane half's four k slots are (c0, c0, c1, c1) of its two channels: the A operand is (h0, h0, h1, h1) in the
// first pass and (m0, m0, m1, m1) in the second, the B operand (b_h(c0), b_m(c0), b_h(c1), b_m(c1)) in both -- the host packs the
// weights that way, two bf16 in the place of each fp32, so the fragment loads and their registers do not change.  What is left out
// (the third pieces: relative 2^-17 per operand) is 30x the rounding of an fp32 product: NOT the fp32 path, an opt-in.
template <int TXN, int TYN, int TN, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(Wino4Args p) {
    constexpr int NTV = TN * TXN * TYN;             // tiles in use (<= 32)
    constexpr int PW = 4 * TXN + 2, PH = 4 * TYN + 2, NPX = TN * PW * PH;
    constexpr int W_RAW = w4_raw_floats(TXN, TYN, TN);
    constexpr int NPIECE = (NPX * 4 + W4_THREADS - 1) / W4_THREADS;
    static_assert(NTV <= 32 && NTV > 24 && NPIECE <= 4, "unsupported patch geometry");
    static_assert((2 * W4_V + 2 * W_RAW) * 4 <= 160 * 1024, "LDS budget (160 KB)");
    static_assert(12 * 32 * W4_EL <= 2 * W4_V + 2 * W_RAW, "exchange tiles must fit the LDS allocation");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Vb = smem;                           // [2][36][2][W4_VH]
    float *Rb = smem + 2 * W4_V;                // [2][NPIECE * 192][W4_PX]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..11: three waves per SIMD
    const int patches = p.tiles_x * p.tiles_y;
    const int per_cb = ((p.N + TN - 1) / TN) * patches;
    const int nS = p.Cin >> 4;
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.u_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;

    // ---- patch decode (uniform) + raw patch loader: NPX pixels x 4 float4 (16 channels); thread handles pieces f = tid + 768 r
    const int id = blockIdx.x;
    const int cb = id / per_cb, rem = id - cb * per_cb;
    const int n_base = (rem / patches) * TN;
    const int pr = rem - (rem / patches) * patches;
    const int pty = pr / p.tiles_x, ptx = pr - pty * p.tiles_x;
    const int oy0 = pty * (4 * TYN), ox0 = ptx * (4 * TXN), n0 = cb * 64;
    unsigned r_off[NPIECE];
#pragma unroll
    for (int r = 0; r < NPIECE; r++) {
        const int f = tid + W4_THREADS * r;
        const int px = f >> 2, cq = f & 3;
        const int img = px / (PW * PH), pq = px - img * (PW * PH);
        const int py = pq / PW, pxx = pq - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx, nn = n_base + img;
        const bool ok = px < NPX && nn < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        r_off[r] = ok ? (unsigned)((((nn * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : oob;     // stays out of range with a channel offset added
    }
    // Loads past the last channel block / chunk are not predicated: they read the neighbouring pixel's channels or the next
    // weight block (or zeros beyond the buffer) into buffers / registers that are never consumed.
    // The refill slots of the LAST super-step have nothing left to fetch for this patch (round 3 let them read the neighbouring pixel's
    // channels into a buffer nobody consumes).  They now fetch the first 16 channels of the patch 256 ids ahead -- the one a workgroup
    // dispatched to this XCD (ids go round the eight XCDs) will open with, most likely on this very CU once this one retires: its cold
    // HBM misses, the bulk of a workgroup's ~10 k-cycle head, turn into L2 hits.  The other patch's pieces are this patch's shifted by
    // one uniform byte offset (same geometry); where that lands outside the tensor the buffer check returns zeros, where it lands on a
    // pixel the other patch does not need it is a wasted line -- either way the data goes where it always went: nowhere.
#ifndef W4_WARM
#define W4_WARM 256
#endif
    unsigned warm_delta = 0;
    {
        const int id2 = (int)blockIdx.x + W4_WARM;
        if (W4_WARM > 0 && id2 < p.total) {
            const int rem2 = id2 % per_cb;
            const int nb2 = (rem2 / patches) * TN, pr2 = rem2 % patches;
            const int pty2 = pr2 / p.tiles_x, ptx2 = pr2 - pty2 * p.tiles_x;
            const int rem1 = (int)blockIdx.x % per_cb, pr1 = rem1 % patches;
            const int d_n = nb2 - (rem1 / patches) * TN, d_y = (pty2 - pr1 / p.tiles_x) * (4 * TYN), d_x = (ptx2 - pr1 % p.tiles_x) * (4 * TXN);
            warm_delta = (unsigned)(((d_n * p.H + d_y) * p.W + d_x) * p.Cin * 4);
        }
    }
    const int nS_ = p.Cin >> 4;
    f32x4 rreg[NPIECE];
    auto raw_gload1 = [&](int S, int r) {
        const unsigned so = S < nS_ ? (unsigned)(S * 64) : warm_delta;           // uniform
        rreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, r_off[r] + so, 0, 0));
    };
    float *const r_l0 = Rb + (tid >> 2) * W4_PX + (tid & 3) * 4;                   // piece r: 192 pixels further
    auto raw_lstore1 = [&](int buf, int r) {
        float *d = r_l0 + buf * W_RAW + r * 192 * W4_PX;
        d[0] = rreg[r][0]; d[1] = rreg[r][1]; d[2] = rreg[r][2]; d[3] = rreg[r][3];
    };

    // ---- weight fragments straight from global memory: packed [Cout/64][Cin/4][12 waves][3 xi][64 lanes][4]
    const unsigned u_base = (unsigned)cb * (unsigned)(p.Cin >> 2) * 36864u + (unsigned)wave * 3072u + (unsigned)lane * 16u;
    // ONE register set, refreshed in place: the fragment of xi e for chunk+1 is requested right after the chunk's last MFMA on
    // xi e has issued (an MFMA reads its operands when it issues) and is needed eight MFMA slots + a barrier later.
    f32x4 fb[3];                                // {n block 0: k step 0, 1; n block 1: k step 0, 1}
    auto u_gload = [&](int e, int chunk) {
        fb[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_base + (unsigned)(chunk * 36864 + e * 1024), 0, 0));
    };

    // ---- input transform B^T d B, thread = (tile, channel, output row a); a is uniform per wave.
    // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]: every row combines at
    // most four patch rows, so the row pass is out = c0 x[k0] + c1 x[k1] + c2 x[k2] + c3 x[k3] with scalar (k, c) per wave.
    const int ta = wave % 6;
    const int tt = 16 * (wave / 6) + 8 * (lane >> 5) + (lane & 7);
    const int tch = (lane >> 3) & 3;
    const int ttc = tt < NTV ? tt : 0;                                             // unused tile slots transform tile 0 again
    const int t_img = ttc / (TXN * TYN), t_ty = (ttc / TXN) % TYN, t_tx = ttc % TXN;
    const int t_roff = ((t_img * PH + 4 * t_ty) * PW + 4 * t_tx) * W4_PX + tch;
    const int t_voff = (ta * 6 * 2 + (tch >> 1)) * W4_VH + tt * 2 + (tch & 1);     // + b' * 2 * W4_VH
    const float *tb0 = Rb + t_roff + (ta == 0 ? 0 : 1) * PW * W4_PX;              // the four patch rows of this wave's output row;
    const float *tb1 = Rb + t_roff + (ta == 5 ? 3 : 2) * PW * W4_PX;              // everything else is a compile-time offset
    const float *tb2 = Rb + t_roff + (ta == 0 ? 4 : ta == 5 ? 5 : 3) * PW * W4_PX;
    const float *tb3 = Rb + t_roff + 4 * PW * W4_PX;                               // weight 0 for a = 0, 5
    const float tc0 = ta == 0 ? 4.f : ta == 1 ? -4.f : ta == 2 ? 4.f : ta == 3 ? -2.f : ta == 4 ? 2.f : 4.f;
    const float tc1 = ta == 0 ? -5.f : ta == 1 ? -4.f : ta == 2 ? -4.f : ta == 3 ? -1.f : ta == 4 ? -1.f : -5.f;
    const float tc2 = ta == 0 ? 1.f : ta == 1 ? 1.f : ta == 2 ? -1.f : ta == 3 ? 2.f : ta == 4 ? -2.f : 1.f;
    const float tc3 = (ta == 0 || ta == 5) ? 0.f : 1.f;
    const f32x2 tc0v = {tc0, tc0}, tc1v = {tc1, tc1}, tc2v = {tc2, tc2}, tc3v = {tc3, tc3};
    const f32x2 k_m4 = {-4.f, -4.f}, k_2 = {2.f, 2.f};
    f32x2 tq[3];                                                                    // row-pass values of the patch columns (1,2) (3,4) (0,5)
    auto tr_col2 = [&](int off, int pr) {                                           // row pass of two patch columns, packed
        const int o0 = off + (pr == 2 ? 0 : 2 * pr + 1) * W4_PX, o1 = off + (pr == 2 ? 5 : 2 * pr + 2) * W4_PX;
        const f32x2 x0 = {tb0[o0], tb0[o1]}, x1 = {tb1[o0], tb1[o1]}, x2 = {tb2[o0], tb2[o1]}, x3 = {tb3[o0], tb3[o1]};
        if (W4_DBG & 64) { tq[pr] = x0 + x1 + x2 + x3; return; }
        tq[pr] = pk_fma_s(tc0v, x0, pk_fma_s(tc1v, x1, pk_fma_s(tc2v, x2, pk_mul_s(tc3v, x3))));
    };
    auto tr_out = [&](float *vp) {                                                  // column pass (the same B^T along the columns)
        const f32x2 q12 = tq[0], q34 = tq[1], q05 = tq[2];
        if (W4_DBG & 64) { vp[0] = q12[0]; vp[2 * W4_VH] = q12[1]; vp[4 * W4_VH] = q34[0]; vp[6 * W4_VH] = q34[1]; vp[8 * W4_VH] = q05[0]; vp[10 * W4_VH] = q05[1]; return; }
        const f32x2 nm = pk_fma_s(k_m4, q12, q34);                                  // (q3 - 4 q1, q4 - 4 q2)
        const f32x2 nm2 = pk_sub(q34, q12);                                         // (q3 - q1, q4 - q2)
        const f32x2 o12 = pk_hi_pm_lo(nm);                                          // b' = 1 | 2
        const f32x2 o34 = pk_hi_pm_clo(k_2, nm2);                                   // b' = 3 | 4
        vp[0 * 2 * W4_VH] = __builtin_fmaf(4.f, q05[0], __builtin_fmaf(-5.f, q12[1], q34[1]));
        vp[1 * 2 * W4_VH] = o12[0];
        vp[2 * 2 * W4_VH] = o12[1];
        vp[3 * 2 * W4_VH] = o34[0];
        vp[4 * 2 * W4_VH] = o34[1];
        vp[5 * 2 * W4_VH] = __builtin_fmaf(4.f, q12[0], __builtin_fmaf(-5.f, q34[0], q05[1]));
    };

    // ---- MFMA: wave w owns xi = 3w + e; A = V rows (tiles), B = U columns (output channels)
    const int frow = lane & 31, fh = lane >> 5;
    const int f_off = (wave * 3 * 2 + fh) * W4_VH + frow * 2;
    f32x16 acc[3][2];
    f32x2 fa[3];                                // likewise one set: xi 0, 1 refreshed after the barrier, xi 2 after its last MFMA
    auto frag_load = [&](int e, int buf) { fa[e] = *reinterpret_cast<const f32x2 *>(Vb + buf * W4_V + f_off + e * 2 * W4_VH); };
    u32x2 sa_h, sa_m;                           // SPLIT: the two pieces of the current xi's A fragment
    auto mfma_g = [&](int g) {                                 // MFMA g of 12 of a chunk: xi e, k step t (SPLIT: piece t), n block nb
        const int e = g >> 2, t = (g >> 1) & 1, nb = g & 1;
        if (!SPLIT) {
            if (W4_PRIO) __builtin_amdgcn_s_setprio(W4_PRIO);
            acc[e][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e][t], fb[e][2 * nb + t], acc[e][nb], 0, 0, 0);
            if (W4_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            if ((g & 3) == 0) {
                sa_h[0] = cvt_pk_same(fa[e][0]); sa_h[1] = cvt_pk_same(fa[e][1]);
                sa_m[0] = cvt_pk_same(fa[e][0] - __builtin_bit_cast(float, sa_h[0] & 0xffff0000u));
                sa_m[1] = cvt_pk_same(fa[e][1] - __builtin_bit_cast(float, sa_h[1] & 0xffff0000u));
            }
            const f32x2 bq = {fb[e][2 * nb], fb[e][2 * nb + 1]};
            acc[e][nb] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, t ? sa_m : sa_h), __builtin_bit_cast(s16x4, bq),
                                                                 acc[e][nb], 0, 0, 0);
        }
    };

    // ---- head
#pragma unroll
    for (int r = 0; r < NPIECE; r++) raw_gload1(0, r);
#pragma unroll
    for (int e = 0; e < 3; e++) u_gload(e, 0);
#pragma unroll
    for (int r = 0; r < NPIECE; r++) raw_lstore1(0, r);
    __syncthreads();
#pragma unroll
    for (int pr = 0; pr < 3; pr++) tr_col2(0, pr);
    tr_out(Vb + t_voff);
#pragma unroll
    for (int e = 0; e < 3; e++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[e][b][r] = 0.f;
    __syncthreads();
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
#pragma unroll
    for (int e = 0; e < 3; e++) frag_load(e, 0);

    // The super-step loop is unrolled by two: the raw-buffer parity, hence every LDS address offset, is a compile-time constant.
    for (int S2 = 0; S2 < nS; S2 += 2) {
#pragma unroll
        for (int sq = 0; sq < 8; sq++) {
            const int sp = sq >> 2, q = sq & 3;                    // super-step parity, chunk within the super-step
            const int S = S2 + sp;
            const int chunk = 4 * S + q;
            if (sp == 1 && q == 0 && S >= nS) break;               // odd number of super-steps (uniform)
            const int nxt = (q & 1) ^ 1;
            const int roff = (((sq + 1) >> 2) & 1) * W_RAW + ((q + 1) & 3) * 4;      // raw patch of chunk+1
            float *vp = Vb + nxt * W4_V + t_voff;
            // nine slots of one MFMA plus a share of the side work for chunk+1, in program order:
            //   3, 7  weight fragments of xi 0, 1 for chunk+1 (global -> the registers their last MFMA has just read)
            //   0-1   raw patch refill for the next 16 channels: pieces 0, 1 load at q=0 / store at q=1, pieces 2, 3 at q=1 / q=2
            //   2,4,6 input transform: LDS reads + packed row pass of two patch columns each;  8 column pass + LDS writes
#pragma unroll
            for (int g = 0; g < 9; g++) {
                mfma_g(g);
                if (g == 3 && !(W4_DBG & 16)) u_gload(0, chunk + 1);
                if (g == 7 && !(W4_DBG & 16)) u_gload(1, chunk + 1);
                if (g >= W4_RAW_SLOT && g < W4_RAW_SLOT + 2 && !(W4_DBG & 32)) {
                    const int gi = g - W4_RAW_SLOT;
                    if (q == 0) raw_gload1(S + 1, gi);
                    if (q == 1) { raw_lstore1(sp ^ 1, gi); if (gi + 2 < NPIECE) raw_gload1(S + 1, gi + 2); }
                    if (q == 2 && gi + 2 < NPIECE) raw_lstore1(sp ^ 1, gi + 2);
                }
                if (!(W4_DBG & 8)) {
                    if (g == 2 || g == 4 || g == 6) tr_col2(roff, (g - 2) >> 1);
                    if (g == 8) tr_out(vp);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            frag_load(0, nxt);                                    // chunk+1 is published: its A fragments under the last three MFMAs
            frag_load(1, nxt);
#pragma unroll
            for (int g = 9; g < 12; g++) mfma_g(g);
            frag_load(2, nxt);
            if (!(W4_DBG & 16)) u_gload(2, chunk + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                            // every wave is done with V / the raw patch: the exchange tiles reuse them
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();

    // ---- output transform Y = A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
    // Wave w holds M_ij for i = w >> 1, j = 3 (w & 1) + e.  Column pass (over j) in registers; a pass handles two output
    // columns b of 32 output channels, so the shared sums are formed once:
    //   j = 0,1,2:  s = M1 + M2, d = M1 - M2:   b0: M0 + s   b2: s        |  b1: d    b3: d
    //   j = 3,4,5:  s = M3 + M4, d = M3 - M4:   b0: s        b2: 4 s      |  b1: 2 d  b3: 8 d + M5
    // The twelve partial tiles go through LDS; T_i = partial(2i) + partial(2i+1), Y_a = sum_i A^T[a][i] T_i.
    float *ex = smem;                                           // [12 waves][32 tiles][W4_EL]: columns = 2 b x 32 channels
    const int up = MODE == 2 ? p.up : 1;
    const int jh = wave & 1;
    const int item = tid;                                       // consumer items = 32 tiles x 2 b x 8 channel quads (threads 0..511)
    const int c_tile = (item >> 4) & 31, c_bs = (item >> 3) & 1, cq = item & 7;
    const int c_img = c_tile / (TXN * TYN), c_ty = (c_tile / TXN) % TYN, c_tx = c_tile % TXN;
    const int c_n = n_base + c_img, c_oy = oy0 + 4 * c_ty;
    const bool c_on = tid < 512 && c_tile < NTV && c_n < p.N;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res), 0, (int)p.res_bytes, 0x00020000);
    const unsigned y_row = MODE == 3 ? (unsigned)((p.W >> 1) * p.out_ldc * 4) : (unsigned)(p.W * up * p.out_ldc * 4);                  // bytes per output row
    const unsigned y_pix0 = MODE == 3 ? (unsigned)(c_n * (p.H >> 1) + (c_oy >> 1)) * y_row + (unsigned)(p.out_coff * 4)
                                      : (unsigned)((c_n * p.H + c_oy) * up) * y_row + (unsigned)(p.out_coff * 4);
    f32x4 hold[4];                                              // MODE 3: the even column's four rows, kept for the odd pass
    const unsigned r_row = (unsigned)(p.W * p.res_ldc * 4);
    const unsigned r_pix0 = (unsigned)(c_n * p.H + c_oy) * r_row;
    // MODE 1: the residual rows of a pass are requested one pass ahead (pass 0's before the first partial tiles are written, pass p + 1's
    // once pass p's stores are issued): requested at the top of the consumer section their round trip sat between the barrier and the
    // first LDS read of every pass.
    f32x4 rres[4];
    auto res_gload = [&](int pass) {
        const int odd = pass >> 1, nt = pass & 1;
        const int ox = ox0 + 4 * c_tx + odd + 2 * c_bs, col = n0 + nt * 32 + cq * 4;
        const unsigned ro = (c_on && col < p.cout_store && ox < p.W) ? r_pix0 + (unsigned)((ox * p.res_ldc + col) * 4) : oob;
#pragma unroll
        for (int a = 0; a < 4; a++)                              // rows below the image read as zeros (beyond the buffer) or a later image: not stored
            rres[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro + (ro == oob ? 0u : (unsigned)a * r_row), 0, 0));
    };
    if (MODE == 1 && W4_RESPF) res_gload(0);
    // the bias of the two channel halves, requested here: as a global load at the top of every pass's consumer section its round trip
    // (the consumer knock-out: 7 k of the epilogue's 13.3 k cycles) stood between the barrier and the stores four times per workgroup
    f32x4 bias2[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) bias2[nt] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + nt * 32 + cq * 4);     // Cout is a multiple of 64
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int odd = MODE == 3 ? (pass & 1) : (pass >> 1), nt = MODE == 3 ? (pass >> 1) : (pass & 1);     // the pair of output columns (0, 2) or (1, 3); channel half
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float a0 = acc[0][nt][r], a1 = acc[1][nt][r], a2 = acc[2][nt][r];
            float lo, hi;                                       // partials of columns b = odd, odd + 2
            if (jh == 0) {
                if (odd) { lo = a1 - a2; hi = lo; } else { hi = a1 + a2; lo = a0 + hi; }
            } else {
                if (odd) { const float d = a0 - a1; lo = 2.f * d; hi = __builtin_fmaf(8.f, d, a2); }
                else { lo = a0 + a1; hi = 4.f * lo; }
            }
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (W4_DBG & 4) { if (lo == 123.456f && hi == 1.f) ex[tile] = lo; continue; }
            ex[(wave * 32 + tile) * W4_EL + frow] = lo;
            ex[(wave * 32 + tile) * W4_EL + 32 + frow] = hi;
        }
        __syncthreads();
        const int ox = ox0 + 4 * c_tx + odd + 2 * c_bs;
        const int col = n0 + nt * 32 + cq * 4;
        if (!(W4_DBG & 2) && c_on && col < p.cout_store && ox < p.W) {
            const float *ep = ex + c_tile * W4_EL + c_bs * 32 + cq * 4;
            auto T = [&](int i) {
                return *reinterpret_cast<const f32x4 *>(ep + (2 * i) * 32 * W4_EL) + *reinterpret_cast<const f32x4 *>(ep + (2 * i + 1) * 32 * W4_EL);
            };
            if (MODE == 1 && !W4_RESPF) res_gload(pass);
            const f32x4 bias4 = bias2[nt];
            f32x4 yv[4];
            {
                const f32x4 t1 = T(1), t2 = T(2);
                const f32x4 s12 = t1 + t2, d12 = t1 - t2;
                yv[0] = T(0) + s12; yv[1] = d12; yv[2] = s12; yv[3] = d12 + T(5);
                const f32x4 t3 = T(3), t4 = T(4);
                const f32x4 s34 = t3 + t4, d34 = t3 - t4;
                yv[0] += s34; yv[1] += 2.f * d34; yv[2] += 4.f * s34; yv[3] += 8.f * d34;
            }
            if (MODE == 3) {
                f32x4 v[4];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    v[a] = yv[a] + bias4;
                    v[a][0] = fmaxf(v[a][0], 0.f); v[a][1] = fmaxf(v[a][1], 0.f); v[a][2] = fmaxf(v[a][2], 0.f); v[a][3] = fmaxf(v[a][3], 0.f);
                }
                if (odd == 0) {
#pragma unroll
                    for (int a = 0; a < 4; a++) hold[a] = v[a];
                } else {
                    const unsigned po = y_pix0 + (unsigned)(((ox >> 1) * p.out_ldc + col) * 4);
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {             // pooled rows (c_oy >> 1) + h2: rows 2 h2, 2 h2 + 1 of the tile (both inside the map or both below it: H is even)
                        f32x4 m;
#pragma unroll
                        for (int k = 0; k < 4; k++) m[k] = fmaxf(fmaxf(hold[2 * h2][k], hold[2 * h2 + 1][k]), fmaxf(v[2 * h2][k], v[2 * h2 + 1][k]));
                        if ((W4_DBG & 1) && m[0] != 123.456f) continue;
                        const unsigned rowo = c_oy + 2 * h2 < p.H ? po + (unsigned)h2 * y_row : oob;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, m), yr, rowo, 0, 0);
                    }
                }
            }
            const unsigned yo = y_pix0 + (unsigned)((ox * up * p.out_ldc + col) * 4);
#pragma unroll
            for (int a = 0; a < 4 && MODE != 3; a++) {
                f32x4 v = yv[a] + bias4;
                if (MODE == 1) v += rres[a];
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if ((W4_DBG & 1) && v[0] != 123.456f) continue;
                const unsigned rowo = c_oy + a < p.H ? yo + (unsigned)(a * up) * y_row : oob;      // rows below the image: dropped by the range check
                if (MODE != 2) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yr, rowo, 0, 0);
                } else {                                        // nearest upsample: up x up replicas
                    for (int dy = 0; dy < up; dy++)
                        for (int dx = 0; dx < up; dx++)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yr,
                                                                   rowo + (unsigned)dy * y_row + (unsigned)(dx * p.out_ldc * 4), 0, 0);
                }
            }
        }
        if (MODE == 1 && W4_RESPF && pass < 3) res_gload(pass + 1);
        if (pass < 3) __syncthreads();
    }
    if (p.dbg && tid == 0) p.dbg[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
}

template <int TXN, int TYN, int TN, int MODE, bool SPLIT>
static int launch_wino4m(Wino4Args a, hipStream_t stream) {
    a.tiles_x = cdiv(a.W, 4 * TXN); a.tiles_y = cdiv(a.H, 4 * TYN);
    const long total = (long)cdiv(a.N, TN) * a.tiles_x * a.tiles_y * (a.Cout / 64);
    PT_CHECK(total < (1L << 31), "ptocr_conv3x3_wino4_f32: too many patches");
    a.total = (int)total;
    const size_t lds = sizeof(float) * (2 * W4_V + 2 * w4_raw_floats(TXN, TYN, TN));
    static DynLds dyn;
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&conv_wino4_kernel<TXN, TYN, TN, MODE, SPLIT>), (int)lds)) return e_;
    hipLaunchKernelGGL((conv_wino4_kernel<TXN, TYN, TN, MODE, SPLIT>), dim3((unsigned)a.total), dim3(W4_THREADS), lds, stream, a);
    return launch_ok("conv_wino4_kernel");
}

template <int TXN, int TYN, int TN>
static int launch_wino4(const Wino4Args &a, hipStream_t stream, bool split) {
    if (split) {
        if (a.res_mode == PTOCR_RES_ADD_PRE_RELU) return launch_wino4m<TXN, TYN, TN, 1, true>(a, stream);
        if (a.up > 1) return launch_wino4m<TXN, TYN, TN, 2, true>(a, stream);
        return launch_wino4m<TXN, TYN, TN, 0, true>(a, stream);
    }
    if (a.pool) return launch_wino4m<TXN, TYN, TN, 3, false>(a, stream);
    if (a.res_mode == PTOCR_RES_ADD_PRE_RELU) return launch_wino4m<TXN, TYN, TN, 1, false>(a, stream);
    if (a.up > 1) return launch_wino4m<TXN, TYN, TN, 2, false>(a, stream);
    return launch_wino4m<TXN, TYN, TN, 0, false>(a, stream);
}

}  // namespace ptocr

using namespace ptocr;

static unsigned long long *g_wino4_dbg = nullptr;
// debug: device buffer of 4 clock samples per workgroup (start, main loop start, main loop end, end); null switches it off
extern "C" void ptocr_wino4_set_timing_buffer(void *d_buf) { g_wino4_dbg = (unsigned long long *)d_buf; }

// patch geometries {TXN, TYN, TN}: tiles along x, along y, images per patch (TXN * TYN * TN <= 32 tile slots)
static const int W4_GEO[][3] = {{8, 4, 1}, {4, 8, 1}, {5, 6, 1}, {4, 4, 2}, {8, 2, 2}, {4, 2, 4}, {7, 1, 4}};
constexpr int W4_NGEO = 7;
static long wino4_patches(int g, int N, int H, int W) {
    return (long)cdiv(N, W4_GEO[g][2]) * cdiv(H, 4 * W4_GEO[g][1]) * cdiv(W, 4 * W4_GEO[g][0]);
}
static int wino4_best_geo(int N, int H, int W) {
    int geo = 0;
    for (int g = 1; g < W4_NGEO; g++)
        if (wino4_patches(g, N, H, W) < wino4_patches(geo, N, H, W)) geo = g;
    return geo;
}

// number of patches the best geometry needs (x Cout / 64 workgroups): lets the host compare with the F(2x2) kernel's count
extern "C" long ptocr_conv3x3_wino4_patches(int N, int H, int W) { return wino4_patches(wino4_best_geo(N, H, W), N, H, W); }

// d_u: weights transformed on the host (U = G g G^T per (cout, cin) in fp64, BN folded), packed
// f32[Cout/64][Cin/4][12][3][64][4]: wave w, xi = 3w + e, lane (n = lane & 31, h = lane >> 5) holds
// {U[xi][c0+2h][n], U[xi][c0+2h+1][n], U[xi][c0+2h][32+n], U[xi][c0+2h+1][32+n]}, c0 = 4 chunk, n relative to the 64-block.
// Everything else as ptocr_conv3x3_wino_f32.
static int wino4_run(bool split, const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                     int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                     int out_ldc, int out_coff, int up, void *stream, int pool = 0) {
    PT_CHECK(d_x && d_u && d_bias && d_y, "ptocr_conv3x3_wino4_f32: null argument");
    PT_CHECK(N > 0 && H > 0 && W > 0, "ptocr_conv3x3_wino4_f32: empty tensor");
    PT_CHECK(Cin % 16 == 0 && Cout % 64 == 0, "ptocr_conv3x3_wino4_f32: need Cin %% 16 == 0 and Cout %% 64 == 0");
    PT_CHECK(relu == 0 || relu == 1, "ptocr_conv3x3_wino4_f32: activation must be none or ReLU");
    PT_CHECK(res_mode == PTOCR_RES_NONE || (res_mode == PTOCR_RES_ADD_PRE_RELU && d_res), "ptocr_conv3x3_wino4_f32: only the pre-ReLU residual add is fused");
    PT_CHECK(up >= 1 && up <= 8 && (up == 1 || res_mode == PTOCR_RES_NONE), "ptocr_conv3x3_wino4_f32: up must be 1..8 and excludes the residual");
    if (cout_store <= 0) cout_store = Cout;
    PT_CHECK(cout_store <= Cout && cout_store % 4 == 0, "ptocr_conv3x3_wino4_f32: cout_store must be a multiple of 4 and <= Cout");
    PT_CHECK(out_ldc % 4 == 0 && out_coff % 4 == 0 && out_ldc >= out_coff + cout_store && (res_mode == 0 || res_ldc % 4 == 0), "ptocr_conv3x3_wino4_f32: channel strides must be multiples of 4");
    Wino4Args a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.relu = relu; a.res_mode = res_mode; a.out_ldc = out_ldc; a.out_coff = out_coff; a.res_ldc = res_ldc > 0 ? res_ldc : Cout;
    a.up = up; a.cout_store = cout_store; a.pool = pool;
    PT_CHECK(!pool || (!split && relu == 1 && res_mode == PTOCR_RES_NONE && up == 1 && H % 2 == 0 && W % 2 == 0),
             "ptocr_conv3x3_wino4_pool2_f32: the fused pool needs ReLU, no residual, no upsample and even H, W");
    a.dbg = g_wino4_dbg;
    a.x_bytes = (long)N * H * W * Cin * 4;
    a.u_bytes = (long)Cout * Cin * 36 * 4;
    a.y_bytes = pool ? (long)N * (H / 2) * (W / 2) * out_ldc * 4 : (long)N * H * up * W * up * out_ldc * 4;
    a.res_bytes = res_mode ? (long)N * H * W * a.res_ldc * 4 : 0;
    PT_CHECK(a.x_bytes < (1L << 31) && a.u_bytes < (1L << 31) && a.y_bytes < (1L << 31) && a.res_bytes < (1L << 31),
             "ptocr_conv3x3_wino4_f32: tensor larger than 2 GiB");
    switch (wino4_best_geo(N, H, W)) {
        case 1: return launch_wino4<4, 8, 1>(a, (hipStream_t)stream, split);
        case 2: return launch_wino4<5, 6, 1>(a, (hipStream_t)stream, split);
        case 3: return launch_wino4<4, 4, 2>(a, (hipStream_t)stream, split);
        case 4: return launch_wino4<8, 2, 2>(a, (hipStream_t)stream, split);
        case 5: return launch_wino4<4, 2, 4>(a, (hipStream_t)stream, split);
        case 6: return launch_wino4<7, 1, 4>(a, (hipStream_t)stream, split);
        default: return launch_wino4<8, 4, 1>(a, (hipStream_t)stream, split);
    }
}

extern "C" int ptocr_conv3x3_wino4_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                       int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                                       int out_ldc, int out_coff, int up, void *stream) {
    return wino4_run(false, d_x, d_u, d_bias, d_res, d_y, N, H, W, Cin, Cout, cout_store, relu, res_mode, res_ldc, out_ldc, out_coff, up, stream);
}

// 3x3 / stride 1 / pad 1 conv + folded BN + ReLU + MaxPool2d(2, 2) in one launch (conv_wino4_kernel MODE 3): y f32[N, H/2, W/2, out_ldc],
// cout_store channels written; H and W even.  Bit-identical to ptocr_conv3x3_wino4_f32 followed by ptocr_maxpool2d_f32 (max is exact).
extern "C" int ptocr_conv3x3_wino4_pool2_f32(const float *d_x, const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cin,
                                             int Cout, int cout_store, int out_ldc, void *stream) {
    return wino4_run(false, d_x, d_u, d_bias, nullptr, d_y, N, H, W, Cin, Cout, cout_store, 1, PTOCR_RES_NONE, 0, out_ldc, 0, 1, stream, 1);
}

// Experiment (PTOCR_WINO_SPLIT=1 on the host side, off by default): the same convolution with two-piece bf16 operands on the bf16
// matrix pipe (see conv_wino4_kernel, SPLIT).  d_u: the same packing as ptocr_conv3x3_wino4_f32 with every fp32 U replaced by the dword
// bf16(U) | bf16(U - bf16(U)) << 16.  Everything else as ptocr_conv3x3_wino4_f32.
extern "C" int ptocr_conv3x3_wino4_split_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                             int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                                             int out_ldc, int out_coff, int up, void *stream) {
    return wino4_run(true, d_x, d_u, d_bias, d_res, d_y, N, H, W, Cin, Cout, cout_store, relu, res_mode, res_ldc, out_ldc, out_coff, up, stream);
}
