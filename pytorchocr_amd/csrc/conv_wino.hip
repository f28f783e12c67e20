// Winograd F(2x2, 3x3) convolution on fp32 MFMA for the 3x3 / stride 1 / pad 1 layers (85 % of DBNet-r18's FLOPs).
//
// Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A : 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e. 2.25x
// fewer MFMAs than the direct implicit GEMM, at fp32 accuracy (transform constants are 0, +-1, +-1/2).
// Replaces the same ATen conv2d calls as conv_mfma.hip (det_resnet.py:66-82, fpn.py:59-82, det_db_head.py:10).
//
// Work item ("patch"): 64 Winograd tiles (TXN x TYN tiles = 2TXN x 2TYN outputs) of one image x 64 output channels.  One
// persistent workgroup per CU (4 waves, one per SIMD) walks patches id = blockIdx.x, + gridDim.x, ...; consecutive ids
// share the output-channel block, so the weights stay in the XCD's L2.
// The 16 "frequencies" xi = (i, j) are 16 independent GEMMs [64 tiles x Cin] x [Cin x 64]; wave i owns row i (four xi):
// 4 xi x (2x2 MFMA tiles of 32x32) = 256 accumulator registers.
// K loop in chunks of 4 channels (two v_mfma_f32_32x32x2_f32 per tile pair): per chunk each wave issues 32 MFMAs while,
// in their shadow, it (a) transforms the next chunk's input (B^T d B from an LDS copy of the raw patch, 16 channels deep),
// (b) copies the next chunk of pre-transformed weights U (packed contiguously on the host) to LDS, and (c) every fourth
// chunk refills the raw patch for the next 16 channels.  V, U and the raw patch are all double-buffered in LDS (~147 KB).
// Output transform: wave i reduces its four M_ij to T_ib = sum_j M_ij A_jb in registers; the sum over i (A^T) goes through
// LDS, two passes (b = 0, 1), followed by bias / residual / ReLU and 16-byte channel-contiguous stores (optionally
// replicated up x up: the FPN's nearest upsample into the concat buffer).  The next patch's first loads are issued before
// the output transform so that their latency and the store drain hide behind it.
#include "common.h"

namespace ptocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WLD = 6;                     // LDS row stride (floats) of V / U rows holding 4 channels (8-byte aligned, conflict-free)
constexpr int W_V = 16 * 64 * WLD;         // floats per V (or U) buffer
constexpr int WPX = 18;                    // LDS pixel stride (floats) of the raw patch: 16 channels + 2, so that the transform's
                                           // ds_read_b32 (banks mod 32, 32-lane groups) of 8 tiles x 4 channels is conflict-free
constexpr int W_EL = 68;                   // exchange tile row stride
constexpr int wino_raw_floats(int txn) { return ((2 * txn + 2) * (2 * (64 / txn) + 2) + 1) * WPX + 2; }   // + one dump pixel

struct WinoArgs {
    const float *x, *u, *bias, *res;
    float *y;
    int N, H, W, Cin, Cout;                // stride 1, pad 1: output H x W
    int tiles_x, tiles_y;                  // patches per image
    int relu, res_mode, out_ldc, out_coff, res_ldc, up;
    int total;                             // patches x images x (Cout / 64)
    long x_bytes, u_bytes;
};

template <int TXN>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(WinoArgs p) {
    constexpr int TYN = 64 / TXN;                   // tiles per patch column
    constexpr int PW = 2 * TXN + 2, PH = 2 * TYN + 2, NPX = PW * PH;
    constexpr int W_RAW = wino_raw_floats(TXN);
    static_assert(NPX * 4 <= 6 * 256, "raw patch must fit six 16-byte pieces per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Vb = smem;                       // [2][16][64][WLD]
    float *Ub = smem + 2 * W_V;             // [2][16][64][WLD]
    float *Rb = smem + 4 * W_V;             // [2][NPX + 1][WPX]

    // Thread-derived indices are re-derived from an opaque copy of threadIdx.x at the start of every patch and again before
    // the output transform: otherwise the compiler hoists the main loop's ~100 address registers out of the persistent loop
    // and keeps them alive (spilled) through the output transform.
    int tid, lane, wave, tt, tch, t_roff, t_voff, frow, fh, f_off;
    auto rebase = [&]() {
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        // input transform: lane -> (tile 16 w + 8 (lane >> 5) + (lane & 7), channel (lane >> 3) & 3 of the chunk)
        tt = wave * 16 + (lane >> 5) * 8 + (lane & 7);
        tch = (lane >> 3) & 3;
        t_roff = ((2 * (tt / TXN)) * PW + 2 * (tt % TXN)) * WPX + tch;
        t_voff = tt * WLD + tch;
        frow = lane & 31; fh = lane >> 5;
        f_off = (wave * 4 * 64 + frow) * WLD + 2 * fh;            // MFMA fragments: xi = wave*4 + j, row frow, k = 2h + t
    };
    rebase();
    const int patches = p.tiles_x * p.tiles_y;
    const int per_cb = p.N * patches;
    const int nS = p.Cin >> 4;              // super-steps of 16 channels
    const int nchunks = p.Cin >> 2;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.u_bytes, 0x00020000);
    const unsigned oob = 0x80000000u;             // tensors are < 2 GiB: stays out of range after adding a channel offset

    // ---- patch decode (uniform) + raw patch loader: NPX pixels x 4 float4 (16 channels); thread handles pieces f = tid + 256 r
    int n, oy0, ox0, n0;
    unsigned u_base;
    unsigned r_off[6];                      // byte offset of the piece for channel block 0
    unsigned r_valid;                       // bit r: piece r lies inside the image (else it reads as zeros)
    auto decode = [&](int id) {
        const int cb = id / per_cb, rem = id - cb * per_cb;
        n = rem / patches;
        const int pr = rem - n * patches;
        const int pty = pr / p.tiles_x, ptx = pr - pty * p.tiles_x;
        oy0 = pty * (2 * TYN); ox0 = ptx * (2 * TXN); n0 = cb * 64;
        u_base = (unsigned)cb * (unsigned)nchunks * 16384u;
        r_valid = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int f = tid + 256 * r;
            const int px = f >> 2, cq = f & 3;
            const int py = px / PW, pxx = px - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
            const bool ok = px < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            r_off[r] = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : 0u;
            r_valid |= (unsigned)ok << r;
        }
    };
    // Loads past the last channel block / chunk are not predicated: they read the neighbouring pixel's channels or the next
    // weight block (or zeros beyond the buffer) into LDS buffers that are never consumed.
    f32x4 rreg[6];
    auto raw_gload1 = [&](int S, int r) {
        const unsigned off = ((r_valid >> r) & 1u) ? r_off[r] + (unsigned)(S * 64) : oob;
        rreg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    };
    auto raw_lstore1 = [&](int buf, int r) {
        const int f = tid + 256 * r;
        const int px = f >> 2 < NPX ? f >> 2 : NPX;                                // pieces beyond the patch land in the dump pixel
        float *d = Rb + buf * W_RAW + px * WPX + (f & 3) * 4;
        *reinterpret_cast<f32x2 *>(d) = f32x2{rreg[r][0], rreg[r][1]};
        *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rreg[r][2], rreg[r][3]};
    };
    // ---- U loader: chunk = [16 xi][64 cout][4] floats contiguous (16 KB); thread handles float4 f = tid + 256 r
    f32x4 ureg[4];
    auto u_gload1 = [&](int chunk, int r) {
        const unsigned off = u_base + (unsigned)(chunk * 16384 + (tid + 256 * r) * 16);
        ureg[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, off, 0, 0));
    };
    auto u_lstore1 = [&](int buf, int r) {
        float *d = Ub + buf * W_V + (tid + 256 * r) * WLD;
        *reinterpret_cast<f32x2 *>(d) = f32x2{ureg[r][0], ureg[r][1]};
        *reinterpret_cast<f32x2 *>(d + 2) = f32x2{ureg[r][2], ureg[r][3]};
    };
    // ---- input transform: lane -> (tile 16 w + 8 (lane >> 5) + (lane & 7), channel (lane >> 3) & 3 of the chunk); scalar
    // B^T d B in three phases (LDS reads, row pass, column pass + LDS writes) so that the main loop can spread them between MFMAs
    float td[4][4], tq[4][4];
    auto tr_read = [&](const float *rp, int a, int b) { td[a][b] = rp[(a * PW + b) * WPX]; };
    auto tr_rows = [&](int b) {      // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        tq[0][b] = td[0][b] - td[2][b];
        tq[1][b] = td[1][b] + td[2][b];
        tq[2][b] = td[2][b] - td[1][b];
        tq[3][b] = td[1][b] - td[3][b];
    };
    auto tr_cols = [&](float *vp, int a) {
        vp[(a * 4 + 0) * 64 * WLD] = tq[a][0] - tq[a][2];
        vp[(a * 4 + 1) * 64 * WLD] = tq[a][1] + tq[a][2];
        vp[(a * 4 + 2) * 64 * WLD] = tq[a][2] - tq[a][1];
        vp[(a * 4 + 3) * 64 * WLD] = tq[a][1] - tq[a][3];
    };
    auto raw_ptr = [&](int chunk) { return Rb + ((chunk >> 2) & 1) * W_RAW + t_roff + (chunk & 3) * 4; };

    // MFMA operand fragments, two register sets: set (chunk & 1) is read right after the barrier that publishes the
    // chunk's V / U, while the last MFMAs of the previous chunk are still executing
    f32x16 acc[4][2][2];
    f32x2 fa0[2][4], fa1[2][4], fb0[2][4], fb1[2][4];
    auto frag_load = [&](int set, int buf) {
        const float *va = Vb + buf * W_V + f_off;
        const float *ub = Ub + buf * W_V + f_off;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fa0[set][j] = *reinterpret_cast<const f32x2 *>(va + j * 64 * WLD);
            fa1[set][j] = *reinterpret_cast<const f32x2 *>(va + j * 64 * WLD + 32 * WLD);
            fb0[set][j] = *reinterpret_cast<const f32x2 *>(ub + j * 64 * WLD);
            fb1[set][j] = *reinterpret_cast<const f32x2 *>(ub + j * 64 * WLD + 32 * WLD);
        }
    };
    auto mfma_g = [&](int set, int g) {                        // MFMA g of 32 of a chunk: xi j, k step t, tile (ma, nb)
        const int j = g >> 3, t = (g >> 2) & 1, ma = (g >> 1) & 1, nb = g & 1;
        acc[j][ma][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma ? fa1[set][j][t] : fa0[set][j][t],
                                                               nb ? fb1[set][j][t] : fb0[set][j][t], acc[j][ma][nb], 0, 0, 0);
    };

    decode(blockIdx.x);
#pragma unroll
    for (int r = 0; r < 6; r++) raw_gload1(0, r);
#pragma unroll
    for (int r = 0; r < 4; r++) u_gload1(0, r);

    {
        const int c_n = n, c_oy0 = oy0, c_ox0 = ox0, c_n0 = n0;
        // ---- head: raw patch of super-step 0 and weights of chunk 0 (already in flight) to LDS, transform of chunk 0
#pragma unroll
        for (int r = 0; r < 6; r++) raw_lstore1(0, r);
#pragma unroll
        for (int r = 0; r < 4; r++) { u_lstore1(0, r); u_gload1(1, r); }
        __syncthreads();
        {
            const float *rp = raw_ptr(0);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) tr_read(rp, a, b);
#pragma unroll
            for (int b = 0; b < 4; b++) tr_rows(b);
#pragma unroll
            for (int a = 0; a < 4; a++) tr_cols(Vb + t_voff, a);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[j][a][b][r] = 0.f;
        __syncthreads();
        frag_load(0, 0);

        for (int S = 0; S < nS; S++) {                                                        // nS >= 1: no zero-trip path for the accumulators
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int chunk = 4 * S + q;
                const int cur = q & 1, nxt = cur ^ 1;                 // chunk parity == q parity
                const float *rp = raw_ptr(chunk + 1);
                float *vp = Vb + nxt * W_V + t_voff;
                // 24 slots of one MFMA plus a share of the side work for chunk+1/+2, in program order (sched_barrier pins it):
                //   0-3   weights of chunk+1 (loaded a chunk ago) to LDS, weights of chunk+2 into the same registers;
                //   0-2   raw patch refill for the next 16 channels (load at q=0/1, LDS store at q=1/2)
                //   4-11  input transform: LDS reads;  12-15 row pass;  16-19 column pass + LDS writes;  20-23 MFMA only
#pragma unroll
                for (int g = 0; g < 24; g++) {
                    mfma_g(cur, g);
                    if (g < 4) { u_lstore1(nxt, g); u_gload1(chunk + 2, g); }
                    if (g < 3) {
                        if (q == 0) raw_gload1(S + 1, g);
                        if (q == 1) { raw_lstore1((S + 1) & 1, g); raw_gload1(S + 1, 3 + g); }
                        if (q == 2) raw_lstore1((S + 1) & 1, 3 + g);
                    }
                    if (g >= 4 && g < 12) { tr_read(rp, (g - 4) >> 1, 2 * (g & 1)); tr_read(rp, (g - 4) >> 1, 2 * (g & 1) + 1); }
                    if (g >= 12 && g < 16) tr_rows(g - 12);
                    if (g >= 16 && g < 20) tr_cols(vp, g - 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                // chunk+1 is published: fetch its fragments under the last eight MFMAs of this chunk
                frag_load(nxt, nxt);
#pragma unroll
                for (int g = 24; g < 32; g++) mfma_g(cur, g);
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                            // every wave is done with V / U: the exchange tile reuses them

        // ---- output transform.  A^T = [1 1 1 0; 0 1 -1 -1].  T_b = sum_j M_ij A_jb :  T_0 = M0 + M1 + M2,  T_1 = M1 - M2 - M3
        float *ex = smem;                                           // [4 waves][64 tiles][W_EL]
        const int HW = p.H * p.W;
        const int up = p.up;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            // residual rows for this pass, requested before the exchange so that their latency hides behind it
            f32x4 rres[4][2];
            if (p.res_mode == PTOCR_RES_ADD_PRE_RELU) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int item = tid + 256 * it;
                    const int tile = item >> 4, cq = item & 15;
                    const int ox = c_ox0 + 2 * (tile % TXN) + b;
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const int oy = c_oy0 + 2 * (tile / TXN) + a;
                        const bool ok = oy < p.H && ox < p.W;
                        const long m = ok ? (long)c_n * HW + (long)oy * p.W + ox : 0;
                        rres[it][a] = *reinterpret_cast<const f32x4 *>(p.res + m * p.res_ldc + c_n0 + cq * 4);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float tv = b == 0 ? acc[0][mt][nt][r] + acc[1][mt][nt][r] + acc[2][mt][nt][r]
                                                : acc[1][mt][nt][r] - acc[2][mt][nt][r] - acc[3][mt][nt][r];
                        const int tile = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        ex[(wave * 64 + tile) * W_EL + nt * 32 + frow] = tv;
                    }
                    __builtin_amdgcn_sched_barrier(0);           // keeps the accumulator reads from piling up in VGPRs
                }
            __syncthreads();
            // Y_ab = sum_i A^T[a][i] T_ib :  Y_0b = T_0 + T_1 + T_2,  Y_1b = T_1 - T_2 - T_3 ; items = 64 tiles x 16 channel quads
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int item = tid + 256 * it;
                const int tile = item >> 4, cq = item & 15;
                const f32x4 t0 = *reinterpret_cast<const f32x4 *>(ex + (0 * 64 + tile) * W_EL + cq * 4);
                const f32x4 t1 = *reinterpret_cast<const f32x4 *>(ex + (1 * 64 + tile) * W_EL + cq * 4);
                const f32x4 t2 = *reinterpret_cast<const f32x4 *>(ex + (2 * 64 + tile) * W_EL + cq * 4);
                const f32x4 t3 = *reinterpret_cast<const f32x4 *>(ex + (3 * 64 + tile) * W_EL + cq * 4);
                const int col = c_n0 + cq * 4;
                const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(p.bias + col);
                const int ox = c_ox0 + 2 * (tile % TXN) + b;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const int oy = c_oy0 + 2 * (tile / TXN) + a;
                    if (oy >= p.H || ox >= p.W) continue;
                    f32x4 v = (a == 0 ? t0 + t1 + t2 : t1 - t2 - t3) + bias4;
                    if (p.res_mode == PTOCR_RES_ADD_PRE_RELU) v += rres[it][a];
                    if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (up == 1) {
                        const long m = (long)c_n * HW + (long)oy * p.W + ox;
                        *reinterpret_cast<f32x4 *>(p.y + m * p.out_ldc + p.out_coff + col) = v;
                    } else {                                        // nearest upsample: up x up replicas
                        const long Wu = (long)p.W * up;
                        float *yb = p.y + (((long)c_n * p.H * up + (long)oy * up) * Wu + (long)ox * up) * p.out_ldc + p.out_coff + col;
                        for (int dy = 0; dy < up; dy++)
                            for (int dx = 0; dx < up; dx++) *reinterpret_cast<f32x4 *>(yb + (dy * Wu + dx) * p.out_ldc) = v;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int TXN>
static int launch_wino(const WinoArgs &a, hipStream_t stream) {
    const size_t lds = sizeof(float) * (4 * W_V + 2 * wino_raw_floats(TXN));
    static bool attr_set = false;
    if (!attr_set) {
        PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<TXN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_wino_kernel<TXN>, dim3((unsigned)a.total), dim3(256), lds, stream, a);
    return launch_ok("conv_wino_kernel");
}

}  // namespace ptocr

using namespace ptocr;

// d_u: weights transformed on the host, packed f32[Cout/64][Cin/4][16][64][4] (U = G g G^T per (cout, cin), BN folded).
// up > 1 writes every output pixel up x up times (nearest upsample) into y[N][H*up][W*up][out_ldc].
extern "C" int ptocr_conv3x3_wino_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                      int N, int H, int W, int Cin, int Cout, int relu, int res_mode, int res_ldc, int out_ldc,
                                      int out_coff, int up, void *stream) {
    PT_CHECK(d_x && d_u && d_bias && d_y, "ptocr_conv3x3_wino_f32: null argument");
    PT_CHECK(N > 0 && H > 0 && W > 0, "ptocr_conv3x3_wino_f32: empty tensor");
    PT_CHECK(Cin % 16 == 0 && Cout % 64 == 0, "ptocr_conv3x3_wino_f32: need Cin %% 16 == 0 and Cout %% 64 == 0");
    PT_CHECK(relu == 0 || relu == 1, "ptocr_conv3x3_wino_f32: activation must be none or ReLU");
    PT_CHECK(res_mode == PTOCR_RES_NONE || (res_mode == PTOCR_RES_ADD_PRE_RELU && d_res), "ptocr_conv3x3_wino_f32: only the pre-ReLU residual add is fused");
    PT_CHECK(up >= 1 && up <= 8 && (up == 1 || res_mode == PTOCR_RES_NONE), "ptocr_conv3x3_wino_f32: up must be 1..8 and excludes the residual");
    PT_CHECK(out_ldc % 4 == 0 && out_coff % 4 == 0 && out_ldc >= out_coff + Cout && (res_mode == 0 || res_ldc % 4 == 0), "ptocr_conv3x3_wino_f32: channel strides must be multiples of 4");
    WinoArgs a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_res; a.y = d_y;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.relu = relu; a.res_mode = res_mode; a.out_ldc = out_ldc; a.out_coff = out_coff; a.res_ldc = res_ldc > 0 ? res_ldc : Cout;
    a.up = up;
    a.x_bytes = (long)N * H * W * Cin * 4;
    a.u_bytes = (long)Cout * Cin * 16 * 4;
    PT_CHECK(a.x_bytes < (1L << 31) && a.u_bytes < (1L << 31), "ptocr_conv3x3_wino_f32: tensor larger than 2 GiB");
    // patch geometry: 16 x 16 outputs, or 32 rows x 8 columns when that covers the image with fewer wasted outputs
    const long sq = (long)cdiv(W, 16) * cdiv(H, 16), tall = (long)cdiv(W, 8) * cdiv(H, 32);
    const bool use_tall = tall < sq;
    a.tiles_x = use_tall ? cdiv(W, 8) : cdiv(W, 16);
    a.tiles_y = use_tall ? cdiv(H, 32) : cdiv(H, 16);
    const long total = (long)N * a.tiles_x * a.tiles_y * (Cout / 64);
    PT_CHECK(total < (1L << 31), "ptocr_conv3x3_wino_f32: too many patches");
    a.total = (int)total;
    return use_tall ? launch_wino<4>(a, (hipStream_t)stream) : launch_wino<8>(a, (hipStream_t)stream);
}
