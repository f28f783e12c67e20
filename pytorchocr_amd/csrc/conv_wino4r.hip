// Winograd F(4x4, 3x3) convolution on fp32 MFMA, round-5 re-cut of conv_wino4.hip (same arithmetic: Y = A^T [sum_c (G g G^T) .* (B^T d B)] A,
// 6x6 transform tiles, 36 multiplies per 4x4 output tile and channel pair; same layers: det_resnet.py:66-82, fpn.py:59-82,
// det_db_head.py:10, rec_vgg.py:78-91).  What was wrong with the first cut, by its own probes: a workgroup's head cost
// 8.3 k cycles and its epilogue 13.3 k against 16 x 3 k cycles of main loop on the 64-channel layers (matrix pipe busy 38.7 %).
// Three changes, none of them in the multiplies:
//
//  * A wave owns ONE ROW i of the 6x6 frequencies (xi = 6 i + j, j = 0..5) for 32 of the 64 output channels, instead of three
//    frequencies for all 64.  The column pass of the output transform (over j) then happens entirely in the wave's registers, and what
//    goes through LDS is T_i[b] = sum_j A^T[b][j] M_ij -- four values where the first cut exchanged partial sums of six: half the
//    exchange volume, and a consumer adds six tiles (one per i) instead of twelve partial ones.
//  * The MFMA operands are swapped: A = weights (rows = output channels), B = transformed input (columns = tiles).  A lane's accumulator
//    quad is then FOUR CONSECUTIVE OUTPUT CHANNELS of one tile: the exchange is written with 16-byte LDS stores (16 per lane instead of
//    128 four-byte ones), conflict-free through an XOR swizzle of the channel quad with the tile index, no padding.
//  * The workgroup is PERSISTENT (one per CU, patch ids blockIdx.x, + gridDim.x, ...) and the exchange is double-buffered: the
//    producers of output column b + 1 write while the consumers of column b read (4 barriers instead of 7), and the first 16 input
//    channels of the NEXT patch travel global -> registers -> LDS during the epilogue, whose LDS map leaves raw buffer 0 free for
//    exactly that.  A patch's head is then one barrier, one input transform and the first fragment reads.
//
// LDS map: [ V0 V1 : transformed chunks | R0 : raw patch, even super-steps | R1 : raw patch, odd super-steps ]; the exchange buffers
// (12 waves x 32 tiles x 32 channels each) lie over V0 V1 (E0) and over R1 (E1), never over R0.
// Main loop: as conv_wino4.hip (K in chunks of 4 channels, 12 v_mfma_f32_32x32x2_f32 per wave and chunk, the input transform of the
// next chunk by thread = (tile, channel, output row), weight fragments straight from global memory into the registers their last MFMA
// has just read, raw-patch refill every fourth chunk) with six B-operand reads per chunk instead of three A-operand reads.
#include "common.h"

#ifndef R4_STAGGER
#define R4_STAGGER 0        // experiment: workgroup phase p = (blockIdx.x >> 3) & 3 starts p * R4_STAGGER * 8 k cycles late
#endif
#ifndef R4_UAUX
#define R4_UAUX 0           // cache-policy bits of the weight-fragment loads (1 sc0, 2 nt, 3 both): every element is read once per workgroup and patch
#endif
#ifndef R4_YAUX
#define R4_YAUX 2           // cache-policy bits of the output stores (gfx950: 1 sc0, 2 nt, 16 sc1).  Round 6: nt -- nothing in the launch reads the output
                            // again, and streamed past the L2 it leaves the weights and halo rows there: det forward 14.26 -> 14.12 ms (two alternating
                            // runs each, tools/dbg/det_fwd_ab.sh; sc1 / sc0+sc1: no change).  The same policy on the other conv kernels' stores LOSES
                            // (stem 1393 -> 1462 us, 3x3/s2 570 -> 587, in3 lateral 371 -> 426, in2 lateral 725 -> 2149: their stores are pieces of lines)
#endif
#ifndef R4_DBG
#define R4_DBG 0            // timing experiments only (PTOCR_EXTRA_HIPCC_FLAGS=-DR4_DBG=n): 1 no global stores, 2 no consumer, 4 no exchange writes,
                            // 8 no input transform, 16 no weight-fragment loads, 32 no raw refill, 128 no next-patch prefetch, 256 stamps 1 / 2 on the 100 MHz clock (tools/dbg/r4_clock.py)
#endif

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// packed fp32 VALU ops with wave-uniform coefficient pairs in SGPRs (see conv_wino4.hip)
__device__ __forceinline__ f32x2 r4_pk_mul_s(f32x2 c, f32x2 x) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(c), "v"(x)); return d; }
__device__ __forceinline__ f32x2 r4_pk_fma_s(f32x2 c, f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(y)); return d; }
__device__ __forceinline__ f32x2 r4_pk_sub(f32x2 x, f32x2 y) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ f32x2 r4_pk_hi_pm_lo(f32x2 p) { f32x2 d; asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p)); return d; }
__device__ __forceinline__ f32x2 r4_pk_hi_pm_clo(f32x2 c, f32x2 p) {
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=v"(d) : "s"(c), "v"(p)); return d;
}

constexpr int R4_VH = 80;                      // floats per (xi, k pair) block of V: 32 tiles x 2 channels + 16
constexpr int R4_V = 36 * 2 * R4_VH;           // floats per V buffer
constexpr int R4_THREADS = 768;
constexpr int R4_EX = 12 * 32 * 32;            // floats per exchange buffer: [i 6][channel half 2][tile 32][channel 32]
constexpr int R4_BIAS_LDS = 2048;              // floats of bias kept in LDS behind the buffers (larger layers read it from memory)
// Raw patch in LDS (16 channels deep), filled by LDS-DMA (buffer_load_dwordx4 ... lds: a wave-instruction writes 64 consecutive 16-byte
// slots, lane l the slot l -- the SOURCE address is per lane, the destination is not).  A patch row of PW pixels is PW x 4 slots with one
// hole slot behind every fourth pixel: pixel x, channel quad cq at slot 4 x + cq + (x >> 2) of its row.  The hole does what the 17-float
// pixel stride of the first cut did -- neighbouring tiles (four pixels apart) sit 68 floats apart, so the 8 tiles x 4 channels of a
// 32-lane group of the input transform hit 32 different banks -- without a per-lane destination; the row length is kept odd so that tile
// rows shift by 16 banks.  Lanes whose slot is a hole or beyond the patch fetch an out-of-range offset (zeros).
constexpr int r4_row_slots(int txn) { const int pw = 4 * txn + 2, rs = pw * 4 + (pw + 3) / 4; return rs | 1; }
constexpr int r4_ndma(int txn, int tyn, int tn) { return (tn * (4 * tyn + 2) * r4_row_slots(txn) + 63) / 64; }       // wave-instructions per refill
constexpr int r4_raw_floats(int txn, int tyn, int tn) { return r4_ndma(txn, tyn, tn) * 256; }
constexpr int r4_lds_floats(int txn, int tyn, int tn) {       // [ V0 V1 (+ gap up to one exchange buffer) | R0 | R1 (at least one exchange buffer) ]
    const int raw = r4_raw_floats(txn, tyn, tn);
    return R4_EX + raw + (raw > R4_EX ? raw : R4_EX);
}
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
// one LDS-DMA wave-instruction: 64 x 16 bytes from rsrc[voff (per lane)] to LDS bytes [lds_dst, lds_dst + 1024).  Inline asm on purpose:
// hipcc waits for a builtin LDS-DMA before every later LDS read it cannot prove disjoint and before every barrier; this one it does not
// see, its completion is waited for by hand (vmcnt is in order: see the main loop).  M0 is saved and restored in the same statement.
__device__ __forceinline__ void r4_dma16(u32x4s rsrc, unsigned voff, unsigned lds_dst, unsigned soff = 0) {      // soff: uniform byte offset (scalar operand)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

struct Wino4RArgs {
    const float *x, *u, *bias, *res;
    float *y;
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y;                      // patches per image group
    int relu, res_mode, out_ldc, out_coff, res_ldc, up;
    int cout_store;
    int total;                                 // patches x output-channel blocks
    long x_bytes, u_bytes, y_bytes, res_bytes;
    unsigned long long *dbg;
    // PYR: the input is a pyramid of four 64-channel planes in ONE allocation: plane j holds channels 64 j .. 64 j + 63 of the (virtual)
    // [N,H,W,256] input at 1 / 2^shift of its resolution, [N, H >> shift, W >> shift, 64] at byte offset pyr_off[j]; pixel (y, x) of a
    // plane's channels is pixel (y >> shift, x >> shift) of the plane: a nearest-upsampled concat that is never written (FPN, fpn.py:118-131)
    unsigned pyr_off[4];
    int pyr_shift[4];
};

// MODE: 0 plain, 1 pre-ReLU residual add, 2 nearest-upsample replication, 3 ReLU + MaxPool2d(2, 2) (the pool windows of a 4x4 output tile
// are whole: output columns (0, 1) and (2, 3); the even column's four rows wait in registers for the odd one)
template <int TXN, int TYN, int TN, int MODE, bool PYR = false>
__global__ __launch_bounds__(R4_THREADS) void conv_wino4r_kernel(Wino4RArgs p) {
    constexpr int NTV = TN * TXN * TYN;             // tiles in use (<= 32)
    constexpr int PW = 4 * TXN + 2, PH = 4 * TYN + 2;
    constexpr int W_RAW = r4_raw_floats(TXN, TYN, TN);
    constexpr int RS = r4_row_slots(TXN), NROW = TN * PH, NDMA = r4_ndma(TXN, TYN, TN), NK = (NDMA + 11) / 12;
    constexpr int RB0 = R4_EX, RB1 = R4_EX + W_RAW;          // float offsets of the raw buffers (V at 0; the second within the 64 KB a DS offset field reaches from the first)
    static_assert(2 * R4_V <= R4_EX && (W_RAW + 6 * RS * 4) * 4 < 65536, "LDS map");
    static_assert(NTV <= 32 && NTV > 24 && NK <= 5, "unsupported patch geometry");
    static_assert((r4_lds_floats(TXN, TYN, TN) + R4_BIAS_LDS) * 4 <= 160 * 1024, "LDS budget (160 KB)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Vb = smem;                     // [2][36][2][R4_VH]
    float *const Bl = smem + r4_lds_floats(TXN, TYN, TN);            // the layer's bias (Cout <= R4_BIAS_LDS floats): read per pass of the epilogue without a vector-memory wait
    float *const E0 = smem, *const E1 = smem + RB1;      // exchange buffers (epilogue only): over V, over raw buffer 1

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..11: three waves per SIMD
    const int wi = wave % 6, wh = wave / 6;                          // frequency row, output-channel half
    // Everything below that depends on the lane (LDS addresses of the transform and of the fragments, source offsets of the raw pieces, the
    // weight base) is RECOMPUTED at the top of every patch from an opaque copy of tid: kept across the epilogue -- whose 96 accumulators,
    // output rows and residual rows fill the register file -- the allocator spilled them and reloaded them inside the main loop.
    int lane = tid & 63;
    const int patches = p.tiles_x * p.tiles_y;
    const int per_cb = ((p.N + TN - 1) / TN) * patches;
    const int nS = p.Cin >> 4;

    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res), 0, (int)p.res_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;

    // ---- raw patch by LDS-DMA: wave-instruction d = wave + 12 k fills slots [64 d, 64 d + 64); this lane's slot s = 64 d + lane is
    // (row, x, channel quad) or a hole.  Source byte offset per patch; out of range stays out of range with a channel offset added.
    auto slot_off = [&](int d, int nb, int y0, int x0, int pl = 0, int ln = -1) -> unsigned {      // wave-instruction d: this lane's slot (PYR: in plane pl)
        const int sl = d * 64 + (ln >= 0 ? ln : lane);                               // (divisions by compile-time constants; recomputed per patch: kept in five registers
        const int row = sl / RS, rem = sl - row * RS;               //  the slot codes were spilled and their reloads made the head 2 k cycles longer)
        const int grp = rem / 17, w = rem - grp * 17;
        const int x = 4 * grp + (w >> 2), cq = w & 3;
        const int img = row / PH, py = row - img * PH;
        const int iy = y0 - 1 + py, ix = x0 - 1 + x, nn = nb + img;
        const bool ok = w < 16 && x < PW && row < NROW && nn < p.N && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        if (PYR) {
            const int sh = p.pyr_shift[pl], hs = p.H >> sh, ws = p.W >> sh;
            return ok ? p.pyr_off[pl] + (unsigned)((((nn * hs + (iy >> sh)) * ws + (ix >> sh)) * 64 + cq * 4) * 4) : oob;
        }
        return ok ? (unsigned)((((nn * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : oob;
    };
    auto piece_off = [&](int k, int nb, int y0, int x0, int pl = 0, int ln = -1) -> unsigned { return slot_off(wave + 12 * k, nb, y0, x0, pl, ln); };
    auto decode = [&](int id, int &cb, int &nb, int &y0, int &x0) {
        cb = id / per_cb;
        const int rem = id - cb * per_cb;
        nb = (rem / patches) * TN;
        const int pr = rem - (rem / patches) * patches;
        const int pty = pr / p.tiles_x, ptx = pr - pty * p.tiles_x;
        y0 = pty * (4 * TYN); x0 = ptx * (4 * TXN);
    };
    unsigned r_off[NK];
    const u32x4s xrs = {(unsigned)(unsigned long long)p.x, (unsigned)((unsigned long long)p.x >> 32) & 0xffffu, (unsigned)p.x_bytes, 0x00020000u};
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)smem;      // LDS byte address of the allocation
    auto raw_dma_d = [&](int buf, unsigned voff, int d, unsigned soff = 0) {        // wave-instruction d of a raw patch into raw buffer buf
        r4_dma16(xrs, voff, lds0 + (unsigned)(((buf ? RB1 : RB0) + d * 256) * 4), soff);
    };
    auto raw_dma_at = [&](int buf, unsigned voff, int k, unsigned soff = 0) {       // this wave's piece k
        if (k < NK && wave + 12 * k < NDMA) raw_dma_d(buf, voff, wave + 12 * k, soff);     // (uniform)
    };
    // piece k of the refill slot of the main loop: the raw patch of super-step S of the current patch -- or, in the LAST super-step, whose
    // slots have nothing left to fetch, the first 16 channels of the workgroup's NEXT patch (pf_main; they go to the buffer the slot would
    // have filled, which is raw buffer 0 when the number of super-steps is even).  In-order vmcnt makes the main loop the only safe place
    // for these cold reads: every wave waits for its weight fragments once per chunk anyway, and nothing else waits here.
    // (Measured: the LDS-DMA refill costs ~220 cycles per chunk -- a DMA wave-instruction holds its SIMD's issue for 60-100 cycles,
    // MI355X_MICROARCH.md "LDS-DMA piece issue cost", 3.75 of them per SIMD and chunk -- against ~70 for 16-byte loads + LDS stores
    // through registers; but those need 8-16 registers the accumulators do not leave: with them the allocator spilled into the main loop,
    // 4 300 cycles per chunk instead of 3 050.)
    bool pf_main = false;
    int cb2 = 0, nb2 = 0, oy2 = 0, ox2 = 0;
    auto raw_dma = [&](int buf, int S, int k) {
        if (k >= NK) return;
        if (S < nS) raw_dma_at(buf, r_off[k], k, (unsigned)((PYR ? (S & 3) : S) * 64));       // (PYR: r_off is the plane's, see the main loop; the channel offset rides in the scalar operand)
        else if (pf_main) {
            // (from an opaque copy of the lane id, like the per-patch setup: hoisted out of the patch loop the slot decode was spilled, and its
            // reload here -- a scratch load and a wait for EVERYTHING in flight, the weight fragments just requested included -- cost more
            // than the thirty instructions of the decode)
            int t4 = tid;
            asm volatile("" : "+v"(t4));
            raw_dma_at(buf, piece_off(k, nb2, oy2, ox2, 0, t4 & 63), k);
        }
    };

    // ---- weight fragments straight from global memory: packed [Cout/64][Cin/4][12 waves][3][64 lanes][4]; wave (wh, wi), float4 q,
    // lane (n = lane & 31, kh = lane >> 5) holds U[xi = 6 wi + 2 q + jj][channel 4 chunk + 2 kh + t][cout 64 cb + 32 wh + n] at jj * 2 + t
    unsigned u_base = 0;                                            // uniform: output-channel block and wave
    unsigned u_lane = 0;                                            // per lane
    f32x4 fu[3];
    auto u_gload = [&](int q, int chunk) {
        // (the uniform part of the address rides in the instruction's scalar offset: per chunk a scalar add instead of three vector adds -- a
        // vector-ALU instruction is paid in matrix-pipe time, §3.0)
        fu[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_lane, (int)(u_base + (unsigned)(chunk * 36864 + q * 1024)), R4_UAUX));
    };

    // ---- input transform B^T d B, thread = (tile, channel, output row a); a is uniform per wave (conv_wino4.hip)
    const int ta = wave % 6;
    int t_voff = 0;                                                                // + b' * 2 * R4_VH
    const float *tb0 = nullptr, *tb1 = nullptr, *tb2 = nullptr, *tb3 = nullptr;    // the four patch rows of this wave's output row (weight 0 on tb3 for a = 0, 5)
    auto lane_setup_transform = [&]() {
        const int tt = 16 * (wave / 6) + 8 * (lane >> 5) + (lane & 7);
        const int tch = (lane >> 3) & 3;
        const int ttc = tt < NTV ? tt : 0;                                         // unused tile slots transform tile 0 again
        const int t_img = ttc / (TXN * TYN), t_ty = (ttc / TXN) % TYN, t_tx = ttc % TXN;
        const int t_roff = (t_img * PH + 4 * t_ty) * (RS * 4) + t_tx * 68 + tch;  // float offset of the tile's first pixel (x = 4 t_tx: t_tx holes before it)
        t_voff = (ta * 6 * 2 + (tch >> 1)) * R4_VH + tt * 2 + (tch & 1);
        tb0 = smem + RB0 + t_roff + (ta == 0 ? 0 : 1) * (RS * 4);
        tb1 = smem + RB0 + t_roff + (ta == 5 ? 3 : 2) * (RS * 4);
        tb2 = smem + RB0 + t_roff + (ta == 0 ? 4 : ta == 5 ? 5 : 3) * (RS * 4);
        tb3 = smem + RB0 + t_roff + 4 * (RS * 4);
    };
    const float tc0 = ta == 0 ? 4.f : ta == 1 ? -4.f : ta == 2 ? 4.f : ta == 3 ? -2.f : ta == 4 ? 2.f : 4.f;
    const float tc1 = ta == 0 ? -5.f : ta == 1 ? -4.f : ta == 2 ? -4.f : ta == 3 ? -1.f : ta == 4 ? -1.f : -5.f;
    const float tc2 = ta == 0 ? 1.f : ta == 1 ? 1.f : ta == 2 ? -1.f : ta == 3 ? 2.f : ta == 4 ? -2.f : 1.f;
    const float tc3 = (ta == 0 || ta == 5) ? 0.f : 1.f;
    const f32x2 tc0v = {tc0, tc0}, tc1v = {tc1, tc1}, tc2v = {tc2, tc2}, tc3v = {tc3, tc3};
    const f32x2 k_m4 = {-4.f, -4.f}, k_2 = {2.f, 2.f};
    f32x2 tq[3];                                                                    // row-pass values of the patch columns (1,2) (3,4) (0,5)
    auto tr_col2 = [&](int off, int pr) {                                           // row pass of two patch columns, packed
        const int c0 = pr == 2 ? 0 : 2 * pr + 1, c1 = pr == 2 ? 5 : 2 * pr + 2;      // patch columns; column c of a tile: 16 c floats + the hole behind column 3
        const int o0 = off + 16 * c0 + 4 * (c0 >> 2), o1 = off + 16 * c1 + 4 * (c1 >> 2);
        const f32x2 x0 = {tb0[o0], tb0[o1]}, x1 = {tb1[o0], tb1[o1]}, x2 = {tb2[o0], tb2[o1]}, x3 = {tb3[o0], tb3[o1]};
        tq[pr] = r4_pk_fma_s(tc0v, x0, r4_pk_fma_s(tc1v, x1, r4_pk_fma_s(tc2v, x2, r4_pk_mul_s(tc3v, x3))));
    };
    auto tr_out = [&](float *vp) {                                                  // column pass (the same B^T along the columns)
        const f32x2 q12 = tq[0], q34 = tq[1], q05 = tq[2];
        const f32x2 nm = r4_pk_fma_s(k_m4, q12, q34);                               // (q3 - 4 q1, q4 - 4 q2)
        const f32x2 nm2 = r4_pk_sub(q34, q12);                                      // (q3 - q1, q4 - q2)
        const f32x2 o12 = r4_pk_hi_pm_lo(nm);                                       // b' = 1 | 2
        const f32x2 o34 = r4_pk_hi_pm_clo(k_2, nm2);                                // b' = 3 | 4
        vp[0 * 2 * R4_VH] = __builtin_fmaf(4.f, q05[0], __builtin_fmaf(-5.f, q12[1], q34[1]));
        vp[1 * 2 * R4_VH] = o12[0];
        vp[2 * 2 * R4_VH] = o12[1];
        vp[3 * 2 * R4_VH] = o34[0];
        vp[4 * 2 * R4_VH] = o34[1];
        vp[5 * 2 * R4_VH] = __builtin_fmaf(4.f, q12[0], __builtin_fmaf(-5.f, q34[0], q05[1]));
    };

    // ---- MFMA: wave (wi, wh) owns xi = 6 wi + j; A = U rows (output channels 32 wh + n), B = V columns (tiles)
    int f_off = 0;
    auto lane_setup_frag = [&]() { f_off = (wi * 6 * 2 + (lane >> 5)) * R4_VH + (lane & 31) * 2; };
    f32x16 acc[6];
    f32x2 fv[6];
    auto frag_load = [&](int j, int buf) { fv[j] = *reinterpret_cast<const f32x2 *>(Vb + buf * R4_V + f_off + j * 2 * R4_VH); };
    auto mfma_g = [&](int g) {                                 // MFMA g of 12 of a chunk: xi j = g >> 1, k step t = g & 1
        const int j = g >> 1, t = g & 1;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fu[j >> 1][(j & 1) * 2 + t], fv[j][t], acc[j], 0, 0, 0);
    };

#if R4_STAGGER
    // experiment: persistent workgroups do the same work in lockstep (all CUs fetch, then all multiply, then all store); phase them apart
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 3) * R4_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif
    if (p.Cout <= R4_BIAS_LDS)
        for (int i = tid; i < p.Cout; i += R4_THREADS) Bl[i] = p.bias[i];      // (published by the barriers of the first patch)
    // ---- first patch of this workgroup: its first 16 channels
    int id = blockIdx.x;
    int cb, n_base, oy0, ox0;
    decode(id, cb, n_base, oy0, ox0);
#pragma unroll
    for (int k = 0; k < NK; k++) raw_dma_at(0, piece_off(k, n_base, oy0, ox0), k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (landed; the barrier at the top of the loop publishes them)

    for (;;) {
        if (p.dbg && tid == 0) p.dbg[id * 4 + 0] = __builtin_readcyclecounter();
        {
            int t2 = tid;
            asm volatile("" : "+v"(t2));                            // opaque: what follows is per patch (see above)
            lane = t2 & 63;
        }
        lane_setup_transform();
        lane_setup_frag();
        const int next = id + (int)gridDim.x;
        const bool has_next = next < p.total;
        if (has_next) decode(next, cb2, nb2, oy2, ox2);
        pf_main = has_next && !(nS & 1) && !(R4_DBG & 128);
#pragma unroll
        for (int k = 0; k < NK; k++) r_off[k] = piece_off(k, n_base, oy0, ox0);
        u_base = (unsigned)cb * (unsigned)(p.Cin >> 2) * 36864u + (unsigned)wave * 3072u;
        u_lane = (unsigned)lane * 16u;
#pragma unroll
        for (int q = 0; q < 3; q++) u_gload(q, 0);
        // (odd number of super-steps only: waves 8-11 fetched this patch's first 16 channels behind the previous epilogue: landed; their three
        // loads above may be out)
        if ((nS & 1) && wave >= 8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __syncthreads();                                        // raw buffer 0 is published; the last exchange buffer is read
#pragma unroll
        for (int pr = 0; pr < 3; pr++) tr_col2(0, pr);
        tr_out(Vb + t_voff);
#pragma unroll
        for (int j = 0; j < 6; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
        __syncthreads();
        if (p.dbg && tid == 0) p.dbg[id * 4 + 1] = (R4_DBG & 256) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();
#pragma unroll
        for (int j = 0; j < 4; j++) frag_load(j, 0);

        // The super-step loop is unrolled by two: the raw-buffer parity, hence every LDS address offset, is a compile-time constant.
        for (int S2 = 0; S2 < nS; S2 += 2) {
#pragma unroll
            for (int sq = 0; sq < 8; sq++) {
                const int sp = sq >> 2, q = sq & 3;                    // super-step parity, chunk within the super-step
                const int S = S2 + sp;
                const int chunk = 4 * S + q;
                if (sp == 1 && q == 0 && S >= nS) break;               // odd number of super-steps (uniform)
                // (PYR: the refills issued in super-step S fetch super-step S + 1; when that is the first of the next plane -- 16 channels per
                // super-step, 64 per plane -- the pieces' source offsets are recomputed for it at the end of super-step S - 1, whose last chunk
                // issues no refill: the slot decode again, ~35 instructions a piece, once per 4 super-steps)
                const int nxt = (q & 1) ^ 1;
                const int roff = ((((sq + 1) >> 2) & 1) ? W_RAW : 0) + ((q + 1) & 3) * 4;      // raw patch of chunk+1
                float *vp = Vb + nxt * R4_V + t_voff;
                // nine slots of one MFMA plus a share of the side work for chunk+1, in program order:
                //   3, 7  weight fragments of xi pairs 0, 1 for chunk+1 (global -> the registers their last MFMA has just read)
                //   0-1   raw patch refill for the next 16 channels by LDS-DMA: pieces 0, 1 at q = 0, 2, 3 at q = 1, 4 at q = 2
                //   2,4,6 input transform: LDS reads + packed row pass of two patch columns each;  8 column pass + LDS writes
#pragma unroll
                for (int g = 0; g < 9; g++) {
                    mfma_g(g);
                    if (g == 3) { frag_load(4, q & 1); frag_load(5, q & 1); }      // (this chunk's last two fragments: needed from MFMA 8 on)
                    if (g == 3 && !(R4_DBG & 16)) u_gload(0, chunk + 1);
                    if (g == 7 && !(R4_DBG & 16)) u_gload(1, chunk + 1);
                    if (!(R4_DBG & 32)) {
                        // two pieces per wave in chunks q = 0 and 1, the fifth in q = 2: vmcnt retires in order, so what a chunk issues must have landed
                        // when the NEXT chunk waits for its weight fragments (20 KB per CU and chunk at ~11 B per cycle)
                        if (q < 2 && g < 2) raw_dma(sp ^ 1, S + 1, 2 * q + g);
                        if (q == 2 && g == 0) raw_dma(sp ^ 1, S + 1, 4);
                    }
                    if (!(R4_DBG & 8)) {
                        if (g == 2 || g == 4 || g == 6) tr_col2(roff, (g - 2) >> 1);
                        if (g == 8) tr_out(vp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the raw patch of the next super-step is read from the next chunk on: its DMA pieces are older than weight fragments this wave has
                // waited for since (vmcnt retires in order), except the fifth, issued in this chunk; the counted wait states it
                if (q == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                __syncthreads();
                frag_load(0, nxt);                                    // chunk+1 is published: its first fragments under the last three MFMAs
                frag_load(1, nxt);
#pragma unroll
                for (int g = 9; g < 12; g++) mfma_g(g);
                frag_load(2, nxt);
                frag_load(3, nxt);
                if (!(R4_DBG & 16)) u_gload(2, chunk + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (PYR && sp == 0 && q == 3 && (S2 & 2) && S + 2 < nS) {
                    // (from an opaque copy of the lane id: hoisted out of the loop, the slot decode's values were spilled at the head of the
                    // patch and came back here as sixteen scratch loads, one round trip after the other: 3.5 k cycles per plane)
                    int t3 = tid;
                    asm volatile("" : "+v"(t3));
#pragma unroll
                    for (int k = 0; k < NK; k++) r_off[k] = piece_off(k, n_base, oy0, ox0, (S + 2) >> 2, t3 & 63);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                                            // every wave is done with V and the raw patch: the exchange buffers reuse them
        if (p.dbg && tid == 0) p.dbg[id * 4 + 2] = (R4_DBG & 256) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter();

        // ---- odd number of super-steps (Cin = 16, 48, ...: the main loop's spare slots end on the wrong buffer): the next patch's first 16
        // channels go to raw buffer 0 behind the epilogue, which lies over V and raw buffer 1 only.  Waves 8-11 issue all of it, a quarter per
        // consumer phase (where they have nothing to do): a wave that stores outputs or loads residual rows would stall behind these reads.
        const int n0 = cb * 64;
        auto prefetch_next = [&](int b) {
            if (!has_next || pf_main || wave < 8 || (R4_DBG & 128)) return;    // (uniform)
            for (int d = wave - 8 + 4 * b; d < NDMA; d += 16) raw_dma_d(0, slot_off(d, nb2, oy2, ox2), d);
        };

        // ---- output transform Y = A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].
        // The wave holds M_ij, j = 0..5, for its 32 output channels: a lane's register r is channel (r & 3) + 8 (r >> 2) + 4 kh of tile ln.
        // Column pass in registers, one output column b per pass: T_i[b] = sum_j A^T[b][j] M_ij; the six T_i meet in LDS,
        // Y[a][b] = sum_i A^T[a][i] T_i[b].  Exchange layout [i][channel half][tile][32 channels], the channel quad XORed with tile & 7.
        const int up = MODE == 2 ? p.up : 1;
        const unsigned y_row = MODE == 3 ? (unsigned)((p.W >> 1) * p.out_ldc * 4) : (unsigned)(p.W * up * p.out_ldc * 4);                  // bytes per output row
        const unsigned r_row = (unsigned)(p.W * p.res_ldc * 4);
        // The consumer's per-lane coordinates (consumer items = 32 tiles x 16 channel quads, threads 0..511) are RECOMPUTED in every pass from
        // an opaque copy of tid (round 5b): kept across the four passes beside the 96 accumulators they were spilled, and every reload came
        // with a wait for ALL vector-memory operations in flight -- the output rows stored a few instructions earlier, the residual rows
        // requested a pass ahead -- once per store.
#define R4_CONSUMER_COORDS                                                                                                         \
        int t5 = tid;                                                                                                              \
        asm volatile("" : "+v"(t5));                                                                                               \
        const int c_tile = (t5 >> 4) & 31, c_wh = (t5 >> 3) & 1, c_q = t5 & 7;                                                     \
        const int c_img = c_tile / (TXN * TYN), c_ty = (c_tile / TXN) % TYN, c_tx = c_tile % TXN;                                  \
        const int c_n = n_base + c_img, c_oy = oy0 + 4 * c_ty;                                                                     \
        const int col = n0 + c_wh * 32 + c_q * 4;                                                                                  \
        const bool c_on = t5 < 512 && c_tile < NTV && c_n < p.N && col < p.cout_store;
        f32x4 hold[2];                                              // MODE 3: the even column's two row-pair maxima, kept for the odd pass
        f32x4 rres[4];                                              // MODE 1: the residual rows of a pass, requested one pass ahead
        auto res_gload = [&](int b) {
            R4_CONSUMER_COORDS
            const unsigned r_pix0 = (unsigned)(c_n * p.H + c_oy) * r_row;
            const int ox = ox0 + 4 * c_tx + b;
            const unsigned ro = (c_on && ox < p.W) ? r_pix0 + (unsigned)((ox * p.res_ldc + col) * 4) : oob;
#pragma unroll
            for (int a = 0; a < 4; a++)                              // rows below the image read as zeros (beyond the buffer) or a later image: not stored
                rres[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro + (ro == oob ? 0u : (unsigned)a * r_row), 0, 0));
        };
        if (MODE == 1) res_gload(0);
        const int ln = lane & 31, kh = lane >> 5;
        const int e_w = ((wi * 2 + wh) * 32 + ln) * 32;             // producer: this lane's row of its wave's exchange tile
        auto produce = [&](int b, float *Eb) {
            if (R4_DBG & 4) return;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = 4 * g + k;
                    const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r], a4 = acc[4][r], a5 = acc[5][r];
                    if (b == 0) v[k] = (a0 + (a1 + a2)) + (a3 + a4);
                    else if (b == 1) v[k] = __builtin_fmaf(2.f, a3 - a4, a1 - a2);
                    else if (b == 2) v[k] = __builtin_fmaf(4.f, a3 + a4, a1 + a2);
                    else v[k] = __builtin_fmaf(8.f, a3 - a4, a1 - a2) + a5;
                }
                *reinterpret_cast<f32x4 *>(Eb + e_w + 4 * ((2 * g + kh) ^ (ln & 7))) = v;
            }
        };
        auto consume = [&](int b, const float *Eb) {
            R4_CONSUMER_COORDS
            const unsigned y_pix0 = MODE == 3 ? (unsigned)(c_n * (p.H >> 1) + (c_oy >> 1)) * y_row + (unsigned)(p.out_coff * 4)
                                              : (unsigned)((c_n * p.H + c_oy) * up) * y_row + (unsigned)(p.out_coff * 4);
            const int ox = ox0 + 4 * c_tx + b;
            if ((R4_DBG & 2) || !c_on || ox >= p.W) return;
            const f32x4 bias4 = p.Cout <= R4_BIAS_LDS ? *reinterpret_cast<const f32x4 *>(Bl + (col < p.Cout ? col : 0))
                                                      : *reinterpret_cast<const f32x4 *>(p.bias + (col < p.Cout ? col : 0));      // Cout is a multiple of 64
            const float *ep = Eb + (c_wh * 32 + c_tile) * 32 + 4 * (c_q ^ (c_tile & 7));
            auto T = [&](int i) { return *reinterpret_cast<const f32x4 *>(ep + i * 2048); };
            f32x4 yv[4];
            {                                                   // three tiles at a time: all six in flight is 24 registers beside the 96 accumulators
                const f32x4 t0 = T(0), t1 = T(1), t2 = T(2);
                const f32x4 s12 = t1 + t2, d12 = t1 - t2;
                yv[0] = t0 + s12; yv[1] = d12; yv[2] = s12; yv[3] = d12;
                const f32x4 t3 = T(3), t4 = T(4), t5 = T(5);
                const f32x4 s34 = t3 + t4, d34 = t3 - t4;
                yv[0] += s34; yv[1] += 2.f * d34; yv[2] += 4.f * s34; yv[3] += 8.f * d34; yv[3] += t5;
            }
            if (MODE == 3) {
                f32x4 v[4];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    v[a] = yv[a] + bias4;
                    v[a][0] = fmaxf(v[a][0], 0.f); v[a][1] = fmaxf(v[a][1], 0.f); v[a][2] = fmaxf(v[a][2], 0.f); v[a][3] = fmaxf(v[a][3], 0.f);
                }
                if ((b & 1) == 0) {                              // (the row pairs' maxima: max is exact and order-free, half the registers of the four rows)
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++)
#pragma unroll
                        for (int k = 0; k < 4; k++) hold[h2][k] = fmaxf(v[2 * h2][k], v[2 * h2 + 1][k]);
                } else {
                    const unsigned po = y_pix0 + (unsigned)(((ox >> 1) * p.out_ldc + col) * 4);
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) {             // pooled rows (c_oy >> 1) + h2: rows 2 h2, 2 h2 + 1 of the tile (both inside the map or both below it: H is even)
                        f32x4 m;
#pragma unroll
                        for (int k = 0; k < 4; k++) m[k] = fmaxf(hold[h2][k], fmaxf(v[2 * h2][k], v[2 * h2 + 1][k]));
                        if ((R4_DBG & 1) && m[0] != 123.456f) continue;
                        const unsigned rowo = c_oy + 2 * h2 < p.H ? po + (unsigned)h2 * y_row : oob;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), yr, rowo, 0, R4_YAUX);
                    }
                }
                return;
            }
            const unsigned yo = y_pix0 + (unsigned)((ox * up * p.out_ldc + col) * 4);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                yv[a] += bias4;
                if (MODE == 1) yv[a] += rres[a];
                if (p.relu) { yv[a][0] = fmaxf(yv[a][0], 0.f); yv[a][1] = fmaxf(yv[a][1], 0.f); yv[a][2] = fmaxf(yv[a][2], 0.f); yv[a][3] = fmaxf(yv[a][3], 0.f); }
            }
            // the next pass's residual rows are requested HERE, in front of this pass's stores (round 5b): the vector-memory counter is in
            // order, so requested behind them the wait for the rows in the next pass also waited for these stores' acknowledgement
            if (MODE == 1 && b < 3) res_gload(b + 1);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const f32x4 v = yv[a];
                if ((R4_DBG & 1) && v[0] != 123.456f) continue;
                const unsigned rowo = c_oy + a < p.H ? yo + (unsigned)(a * up) * y_row : oob;      // rows below the image: dropped by the range check
                if (MODE != 2) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yr, rowo, 0, R4_YAUX);
                } else {                                        // nearest upsample: up x up replicas
                    for (int dy = 0; dy < up; dy++)
                        for (int dx = 0; dx < up; dx++)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yr, rowo + (unsigned)dy * y_row + (unsigned)(dx * p.out_ldc * 4), 0, R4_YAUX);
                }
            }
        };
        produce(0, E0);
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (b < 3) produce(b + 1, ((b + 1) & 1) ? E1 : E0);
            prefetch_next(b);   // (pass b - 1's consumers, the readers of this buffer, are behind the last barrier)
            consume(b, (b & 1) ? E1 : E0);
            if (b < 3) __syncthreads();
        }
#undef R4_CONSUMER_COORDS
        if (p.dbg && tid == 0) p.dbg[id * 4 + 3] = __builtin_readcyclecounter();
        if (!has_next) break;
        id = next; cb = cb2; n_base = nb2; oy0 = oy2; ox0 = ox2;
    }
}

template <int TXN, int TYN, int TN, int MODE, bool PYR = false>
static int launch_wino4rm(Wino4RArgs a, hipStream_t stream) {
    a.tiles_x = cdiv(a.W, 4 * TXN); a.tiles_y = cdiv(a.H, 4 * TYN);
    const long total = (long)cdiv(a.N, TN) * a.tiles_x * a.tiles_y * (a.Cout / 64);
    PT_CHECK(total < (1L << 29), "ptocr_conv3x3_wino4r_f32: too many patches");
    a.total = (int)total;
    const size_t lds = sizeof(float) * (r4_lds_floats(TXN, TYN, TN) + R4_BIAS_LDS);
    static DynLds dyn;                                            // (one per template instance; per device inside)
    if (int e_ = raise_dyn_lds(dyn, reinterpret_cast<const void *>(&conv_wino4r_kernel<TXN, TYN, TN, MODE, PYR>), (int)lds)) return e_;
    int n_cu = 0;
    if (int e_ = current_device_cus(&n_cu)) return e_;
    // one persistent workgroup per CU (150 KB of LDS each: one fits); consecutive patch ids still go round the eight XCDs
    static const int per_cu = getenv("PTOCR_WINO4R_GRID") ? atoi(getenv("PTOCR_WINO4R_GRID")) : 1;
    const long want = (long)n_cu * (per_cu > 0 ? per_cu : 1);
    const int grid = (int)(total < want ? total : want);
    hipLaunchKernelGGL((conv_wino4r_kernel<TXN, TYN, TN, MODE, PYR>), dim3((unsigned)grid), dim3(R4_THREADS), lds, stream, a);
    return launch_ok("conv_wino4r_kernel");
}

template <int TXN, int TYN, int TN>
static int launch_wino4r(const Wino4RArgs &a, hipStream_t stream, int pool) {
    if (pool == 2) return launch_wino4rm<TXN, TYN, TN, 0, true>(a, stream);        // (pyramid input: plain epilogue)
    if (pool == 1) return launch_wino4rm<TXN, TYN, TN, 3>(a, stream);
    if (a.res_mode == PTOCR_RES_ADD_PRE_RELU) return launch_wino4rm<TXN, TYN, TN, 1>(a, stream);
    if (a.up > 1) return launch_wino4rm<TXN, TYN, TN, 2>(a, stream);
    return launch_wino4rm<TXN, TYN, TN, 0>(a, stream);
}

}  // namespace ptocr

using namespace ptocr;

static unsigned long long *g_wino4r_dbg = nullptr;
// debug: device buffer of 4 clock samples per PATCH (start, main loop start, main loop end, end); null switches it off
extern "C" void ptocr_wino4r_set_timing_buffer(void *d_buf) { g_wino4r_dbg = (unsigned long long *)d_buf; }

// patch geometries {TXN, TYN, TN} as in conv_wino4.hip (the same choice per map: ptocr_conv3x3_wino4_patches)
static const int R4_GEO[][3] = {{8, 4, 1}, {4, 8, 1}, {5, 6, 1}, {4, 4, 2}, {8, 2, 2}, {4, 2, 4}, {7, 1, 4}};
constexpr int R4_NGEO = 7;
static long r4_patches(int g, int N, int H, int W) {
    return (long)cdiv(N, R4_GEO[g][2]) * cdiv(H, 4 * R4_GEO[g][1]) * cdiv(W, 4 * R4_GEO[g][0]);
}
static int r4_best_geo(int N, int H, int W) {
    int geo = 0;
    for (int g = 1; g < R4_NGEO; g++)
        if (r4_patches(g, N, H, W) < r4_patches(geo, N, H, W)) geo = g;
    return geo;
}

// d_u: weights transformed on the host (U = G g G^T per (cout, cin) in fp64, BN folded), packed
// f32[Cout/64][Cin/4][2 wh][6 wi][3 q][2 kh][32 n][2 jj][2 t] = U[xi = 6 wi + 2 q + jj][cin = 4 chunk + 2 kh + t][cout = 64 cb + 32 wh + n].
// Everything else as ptocr_conv3x3_wino4_f32 / ptocr_conv3x3_wino4_pool2_f32 (pool = 1: ReLU + MaxPool2d(2, 2) in the epilogue).
static int wino4r_run(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                      int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                      int out_ldc, int out_coff, int up, void *stream, int pool, const long long *pyr_off = nullptr,
                      const int *pyr_shift = nullptr, long long pyr_floats = 0) {
    PT_CHECK(d_x && d_u && d_bias && d_y, "ptocr_conv3x3_wino4r_f32: null argument");
    PT_CHECK(N > 0 && H > 0 && W > 0, "ptocr_conv3x3_wino4r_f32: empty tensor");
    PT_CHECK(Cin % 16 == 0 && Cout % 64 == 0, "ptocr_conv3x3_wino4r_f32: need Cin %% 16 == 0 and Cout %% 64 == 0");
    PT_CHECK(relu == 0 || relu == 1, "ptocr_conv3x3_wino4r_f32: activation must be none or ReLU");
    PT_CHECK(res_mode == PTOCR_RES_NONE || (res_mode == PTOCR_RES_ADD_PRE_RELU && d_res), "ptocr_conv3x3_wino4r_f32: only the pre-ReLU residual add is fused");
    PT_CHECK(up >= 1 && up <= 8 && (up == 1 || res_mode == PTOCR_RES_NONE), "ptocr_conv3x3_wino4r_f32: up must be 1..8 and excludes the residual");
    if (cout_store <= 0) cout_store = Cout;
    PT_CHECK(cout_store <= Cout && cout_store % 4 == 0, "ptocr_conv3x3_wino4r_f32: cout_store must be a multiple of 4 and <= Cout");
    PT_CHECK(out_ldc % 4 == 0 && out_coff % 4 == 0 && out_ldc >= out_coff + cout_store && (res_mode == 0 || res_ldc % 4 == 0), "ptocr_conv3x3_wino4r_f32: channel strides must be multiples of 4");
    PT_CHECK(pool != 1 || (relu == 1 && res_mode == PTOCR_RES_NONE && up == 1 && H % 2 == 0 && W % 2 == 0),
             "ptocr_conv3x3_wino4r_pool2_f32: the fused pool needs ReLU, no residual, no upsample and even H, W");
    Wino4RArgs a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.relu = relu; a.res_mode = res_mode; a.out_ldc = out_ldc; a.out_coff = out_coff; a.res_ldc = res_ldc > 0 ? res_ldc : Cout;
    a.up = up; a.cout_store = cout_store;
    a.dbg = g_wino4r_dbg;
    a.x_bytes = (long)N * H * W * Cin * 4;
    for (int j = 0; j < 4; j++) { a.pyr_off[j] = 0; a.pyr_shift[j] = 0; }
    if (pyr_off) {
        PT_CHECK(!pool && res_mode == PTOCR_RES_NONE && up == 1 && Cin == 256 && pyr_shift && pyr_floats > 0 && pyr_floats * 4 < (1LL << 31),
                 "ptocr_conv3x3_wino4r_pyramid_f32: four planes of 64 channels (Cin = 256), no residual / upsample, below 2 GiB");
        for (int j = 0; j < 4; j++) {
            const int sh = pyr_shift[j];
            PT_CHECK(sh >= 0 && sh <= 3 && H % (1 << sh) == 0 && W % (1 << sh) == 0, "ptocr_conv3x3_wino4r_pyramid_f32: plane %d: shift %d does not divide %d x %d", j, sh, H, W);
            const long long need = (long long)N * (H >> sh) * (W >> sh) * 64;
            PT_CHECK(pyr_off[j] >= 0 && pyr_off[j] % 4 == 0 && pyr_off[j] + need <= pyr_floats, "ptocr_conv3x3_wino4r_pyramid_f32: plane %d lies outside the allocation", j);
            a.pyr_off[j] = (unsigned)(pyr_off[j] * 4);
            a.pyr_shift[j] = sh;
        }
        a.x_bytes = (long)pyr_floats * 4;
        pool = 2;                                                // (launch_wino4r: the pyramid instance)
    }
    a.u_bytes = (long)Cout * Cin * 36 * 4;
    a.y_bytes = pool == 1 ? (long)N * (H / 2) * (W / 2) * out_ldc * 4 : (long)N * H * up * W * up * out_ldc * 4;
    a.res_bytes = res_mode ? (long)N * H * W * a.res_ldc * 4 : 0;
    PT_CHECK(a.x_bytes < (1L << 31) && a.u_bytes < (1L << 31) && a.y_bytes < (1L << 31) && a.res_bytes < (1L << 31),
             "ptocr_conv3x3_wino4r_f32: tensor larger than 2 GiB");
    switch (r4_best_geo(N, H, W)) {
        case 1: return launch_wino4r<4, 8, 1>(a, (hipStream_t)stream, pool);
        case 2: return launch_wino4r<5, 6, 1>(a, (hipStream_t)stream, pool);
        case 3: return launch_wino4r<4, 4, 2>(a, (hipStream_t)stream, pool);
        case 4: return launch_wino4r<8, 2, 2>(a, (hipStream_t)stream, pool);
        case 5: return launch_wino4r<4, 2, 4>(a, (hipStream_t)stream, pool);
        case 6: return launch_wino4r<7, 1, 4>(a, (hipStream_t)stream, pool);
        default: return launch_wino4r<8, 4, 1>(a, (hipStream_t)stream, pool);
    }
}

extern "C" int ptocr_conv3x3_wino4r_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                        int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                                        int out_ldc, int out_coff, int up, void *stream) {
    return wino4r_run(d_x, d_u, d_bias, d_res, d_y, N, H, W, Cin, Cout, cout_store, relu, res_mode, res_ldc, out_ldc, out_coff, up, stream, 0);
}

extern "C" int ptocr_conv3x3_wino4r_pool2_f32(const float *d_x, const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cin,
                                              int Cout, int cout_store, int out_ldc, void *stream) {
    return wino4r_run(d_x, d_u, d_bias, nullptr, d_y, N, H, W, Cin, Cout, cout_store, 1, PTOCR_RES_NONE, 0, out_ldc, 0, 1, stream, 1);
}

// The same convolution on an input that exists only as a PYRAMID: four planes of 64 channels in one allocation d_pyr (pyr_floats floats),
// plane j = channels 64 j .. 64 j + 63 of the virtual f32[N,H,W,256] input, stored as f32[N, H >> shift[j], W >> shift[j], 64] at float
// offset off[j] (host arrays of four): the kernel reads pixel (y >> shift, x >> shift) of a plane for pixel (y, x) -- DBNet's
// cat(up8(p5), up4(p4), up2(p3), p2) (fpn.py:118-131) without the upsampled copies.  Bit-identical to ptocr_conv3x3_wino4r_f32 on the
// materialised concat.
extern "C" int ptocr_conv3x3_wino4r_pyramid_f32(const float *d_pyr, const long long *off, const int *shift, long long pyr_floats,
                                                const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cout,
                                                int cout_store, int relu, int out_ldc, int out_coff, void *stream) {
    PT_CHECK(off && shift, "ptocr_conv3x3_wino4r_pyramid_f32: null plane tables");
    return wino4r_run(d_pyr, d_u, d_bias, nullptr, d_y, N, H, W, 256, Cout, cout_store, relu, PTOCR_RES_NONE, 0, out_ldc, out_coff, 1, stream, 0,
                      off, shift, pyr_floats);
}
