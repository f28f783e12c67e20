// DB probability-map post-process on MI355X: threshold -> bit-packed bitmap -> run-based union-find CC labelling
// (foreground 8-connected, background 4-connected) -> ORDER-FREE enumeration of the border states of the 1000 bottom-most
// borders -> per-border min-area box, polygon-mask score, Clipper round-offset unclip, final integer box.
//
// Replaces (bit-exact boxes on identical maps; see DESIGN.md for the two documented sub-pixel exceptions):
//   pytocr/postprocess/db_postprocess.py:45-46          pred > thresh (float32 compare)
//   pytocr/postprocess/db_postprocess_fast/src/db_postprocess.cpp:231-317  BoxesFromBitmap and callees
//   (cv::findContours RETR_LIST/CHAIN_APPROX_SIMPLE, cv::minAreaRect, cv::boxPoints, cv::fillPoly(lineType 1),
//    cv::mean(mask), ClipperOffset(jtRound).Execute), :159-192 GetMiniBoxes, :194-229 BoxScore, :16-64 UnClip.
//
// How the sequential reference maps to the GPU:
//  * Suzuki border following starts every border exactly once; its raster scan finds an OUTER border at the
//    raster-first pixel of each 8-connected foreground component and a HOLE border at the raster-first pixel of
//    each 4-connected background component that is not connected to the image frame.  So the start points are
//    the union-find roots (root = minimum pixel index), found fully in parallel.  RETR_LIST returns borders in
//    reverse discovery order and the reference keeps the first 1000 => the 1000 largest root indices.
//  * The border walk itself is NOT replayed.  A walk is a closed chain of states (pixel p, gap): a gap is a maximal
//    counter-clockwise run of background 8-neighbours of p between two foreground neighbours s_in (where the walk came
//    from) and s_out (where it goes), and the walk visits exactly the gaps that contain a 4-neighbour.  Every such
//    (pixel, gap) pair of the image lies on exactly one border: the border between p's foreground component F and the
//    background component B of the gap -- the outer border of F when B is the component surrounding F (the one left
//    of F's raster-first pixel), else the hole border of B.  Everything the reference takes from a contour is a
//    function of the SET of its states: the number of CHAIN_APPROX_SIMPLE points (states with s_out != s_in + 4),
//    the bounding box, the convex hull (column extremes of the state pixels), and the fillPoly mask (each unit step
//    contributes its crossing toggle and its 4-connected edge pixels; a polygon edge is a run of equal unit steps).
//    So one thread per 32-pixel bitmap word lists the states of its pixels, looks up F and B in the CC labels and adds
//    the state to its border: no sequential walk, no dependence on border length.
//  * Per border: column extremes + exact strict-hull filter (a workgroup), Sklansky + float32 rotating calipers
//    (one LANE per border, 64 borders per wave: the arithmetic is the reference's sequential float32 code), fillPoly bit
//    planes + masked mean in double (a workgroup), unclip + second rectangle + final box (one lane per border).
// Memory: everything is integer/bit work bound by HBM/L2 latency, not by MFMA; the bitmap is 1 bit/pixel.
#include "common.h"
#include <cstdlib>

namespace ptocr {

constexpr int MAX_CAND = 1000;          // reference db_postprocess.cpp:239 (hard-coded max_candidates)
constexpr int CHUNK = 1024;             // pixels per root-count chunk
#ifndef PT_SEL_CHUNKS
#define PT_SEL_CHUNKS 2
#endif
constexpr int SEL_CHUNKS = PT_SEL_CHUNKS;   // chunks per block of select_starts_kernel
constexpr int FRAME = -1;               // label of background connected to the image frame

struct DbpostDims {
    int N, H, W, WW;                    // WW = 32-bit words per bitmap row
    long HW;
    int nchunks;                        // chunks per image
    long pool_cap;                      // border states per image the pool holds
    int strip_y;                        // first row of the bottom strip labelled first (0: whole image in one pass)
};

// ------------------------------------------------------------------------------------------ binarize
// 4 pixels per lane (one 16-B load), nibbles of 8 lanes OR-combined into a 32-bit word.
__global__ __launch_bounds__(256) void binarize_kernel(const float *__restrict__ maps, unsigned *__restrict__ bits,
                                                       DbpostDims d, float thresh, int *__restrict__ strip_runs) {
    const int img = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const float *row = maps + (long)img * d.HW + (long)y * d.W;
    unsigned nib = 0;
    if (x0 + 3 < d.W && ((d.W & 3) == 0)) {
        const float4 v = *reinterpret_cast<const float4 *>(row + x0);
        nib = (v.x > thresh) | ((v.y > thresh) << 1) | ((v.z > thresh) << 2) | ((v.w > thresh) << 3);
    } else {
        for (int k = 0; k < 4; k++)
            if (x0 + k < d.W) nib |= (unsigned)(row[x0 + k] > thresh) << k;
    }
    unsigned v = nib << (4 * (threadIdx.x & 7));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    const int wi = x0 >> 5;
    if ((threadIdx.x & 7) == 0 && wi < d.WW) bits[((long)img * d.H + y) * d.WW + wi] = v;
    // run starts of both polarities in the bottom strip (a pixel that differs from its left neighbour; column 0 against the background
    // frame): every component of the strip, foreground or hole, contains at least one
    if (strip_runs && d.strip_y && y >= d.strip_y) {             // uniform over the block
        const int lane = threadIdx.x & 63;
        unsigned prev = __shfl_up(v, 8);                         // the word to the left (same value on the 8 lanes of a word)
        if (lane < 8) prev = (x0 >= 32 && wi < d.WW && (row[(wi << 5) - 1] > thresh)) ? 0x80000000u : 0u;     // first word of the wave: the pixel itself
        int c = 0;
        if ((threadIdx.x & 7) == 0 && wi < d.WW) {
            const int valid = d.W - (wi << 5);
            const unsigned mask = valid >= 32 ? 0xffffffffu : ((1u << valid) - 1u);
            c = __popc((v ^ ((v << 1) | (prev >> 31))) & mask);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
        __shared__ int wsum[4];                                  // one atomic per block: 16 k atomics on 32 words doubled the kernel's time
        if (lane == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (t) atomicAdd(&strip_runs[img], t);
        }
    }
}

// The same for W % 32 == 0 (every map the models emit): threads run over the image's pixel quads without regard to rows, so no
// thread idles past the end of a row (1280 = 1024 + 256 left 3/8 of a row's second block idle), two quads per thread in flight.
__global__ __launch_bounds__(256) void binarize_flat_kernel(const float *__restrict__ maps, unsigned *__restrict__ bits,
                                                            DbpostDims d, float thresh, int *__restrict__ strip_runs) {
    const int img = blockIdx.y;
    const int qpr = d.W >> 2;                                    // quads per row (a multiple of 8: the 8 lanes of a word share a row)
    const long nq = (long)d.H * qpr;
    const float *base = maps + (long)img * d.HW;
    int cnt = 0;
    const bool any_strip = strip_runs && d.strip_y && (long)(blockIdx.x * 512 + 511) / qpr >= d.strip_y;      // uniform over the block
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        const long q = (long)blockIdx.x * 512 + rep * 256 + threadIdx.x;
        const bool on = q < nq;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) v4 = *reinterpret_cast<const float4 *>(base + q * 4);
        const unsigned nib = (v4.x > thresh) | ((v4.y > thresh) << 1) | ((v4.z > thresh) << 2) | ((v4.w > thresh) << 3);
        unsigned v = nib << (4 * (threadIdx.x & 7));
        v |= __shfl_xor(v, 1);
        v |= __shfl_xor(v, 2);
        v |= __shfl_xor(v, 4);
        const int y = on ? (int)(q / qpr) : 0, x0 = on ? (int)(q - (long)y * qpr) * 4 : 0;
        const int wi = x0 >> 5;
        if (on && (threadIdx.x & 7) == 0) bits[((long)img * d.H + y) * d.WW + wi] = v;
        if (any_strip) {
            unsigned prev = __shfl_up(v, 8);                     // the word to the left, unless this is the wave's or the row's first word
            if ((threadIdx.x & 63) < 8) prev = (on && x0 >= 32 && (base[q * 4 - 1 - (x0 & 31)] > thresh)) ? 0x80000000u : 0u;
            if (x0 < 32) prev = 0u;
            if (on && (threadIdx.x & 7) == 0 && y >= d.strip_y) cnt += __popc(v ^ ((v << 1) | (prev >> 31)));
        }
    }
    if (any_strip) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
        __shared__ int wsum[4];
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (t) atomicAdd(&strip_runs[img], t);
        }
    }
}

__global__ __launch_bounds__(256) void pack_u8_kernel(const uint8_t *__restrict__ bm, unsigned *__restrict__ bits, DbpostDims d) {
    const int img = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const uint8_t *row = bm + (long)img * d.HW + (long)y * d.W;
    unsigned nib = 0;
    for (int k = 0; k < 4; k++)
        if (x0 + k < d.W) nib |= (unsigned)(row[x0 + k] != 0) << k;
    unsigned v = nib << (4 * (threadIdx.x & 7));
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    const int wi = x0 >> 5;
    if ((threadIdx.x & 7) == 0 && wi < d.WW) bits[((long)img * d.H + y) * d.WW + wi] = v;
}

// cv2.dilate(mask, [[1,1],[1,1]]) (db_postprocess.py:52-55): anchor (1,1) => pixel |= left | up | up-left, on packed words
__global__ __launch_bounds__(256) void dilate2x2_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out, DbpostDims d) {
    const int img = blockIdx.z, y = blockIdx.y;
    const int wi = blockIdx.x * 256 + threadIdx.x;
    if (wi >= d.WW) return;
    const unsigned *row = in + ((long)img * d.H + y) * d.WW;
    unsigned v = row[wi], c = wi ? row[wi - 1] >> 31 : 0u;
    v |= (v << 1) | c;
    if (y > 0) {
        const unsigned u = row[wi - d.WW], cu = wi ? row[wi - d.WW - 1] >> 31 : 0u;
        v |= u | (u << 1) | cu;
    }
    if (wi == d.WW - 1 && (d.W & 31)) v &= (1u << (d.W & 31)) - 1;      // keep the tail bits beyond W clear
    out[((long)img * d.H + y) * d.WW + wi] = v;
}

// LDS hand-off between the lanes of ONE wave (a wave's LDS operations execute in order; the fences keep the compiler from moving accesses)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- work items of a wave, one per lane.  Most kernels below give a lane one 32-pixel bitmap word and let it walk the word's items (run
// starts, run boundaries with a link, border pixels) one after the other; on ragged maps a few words hold twenty items, each a chain of
// dependent loads, while the other lanes of the wave wait (round 4: that, not bytes, was the stage's time on the scene checkpoint's maps).
// With these helpers a lane publishes the bit mask of its items; the wave then walks ALL items of its 64 words, item `it` going to lane
// it % 64 whatever word it sits in: (word lane, bit) from a prefix sum over the words' item counts.
struct WaveItems { int pre[65]; unsigned mask[64]; };
__device__ __forceinline__ int wave_items_publish(WaveItems &wi, unsigned mymask) {
    const int lane = threadIdx.x & 63;
    int incl = __popc(mymask);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    wi.pre[lane + 1] = incl;
    if (lane == 0) wi.pre[0] = 0;
    wi.mask[lane] = mymask;
    wave_lds_sync();
    return wi.pre[64];
}
// item `it` (< the published total): the lane L whose word holds it and its bit i in that word
__device__ __forceinline__ void wave_item(const WaveItems &wi, int it, int &L, int &i) {
    L = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) if (L + step < 64 && wi.pre[L + step] <= it) L += step;
    unsigned mm = wi.mask[L];
    int nth = it - wi.pre[L];
    i = 0;
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {                       // the nth set bit, by halving
        const int c = __popc(mm & ((1u << sh) - 1u));
        if (nth >= c) { nth -= c; i += sh; mm >>= sh; } else mm &= (1u << sh) - 1u;
    }
}

// ------------------------------------------------------------------------------------------ CC labelling
__device__ __forceinline__ int pix(const unsigned *rowbits, int x) { return (rowbits[x >> 5] >> (x & 31)) & 1; }

// index (x) of the first pixel of the horizontal run of equal class that contains x
__device__ __forceinline__ int run_start(const unsigned *rowbits, int x) {
    int wi = x >> 5;
    const int b = x & 31;
    unsigned w = rowbits[wi];
    const bool cls = (w >> b) & 1;
    unsigned m = (cls ? ~w : w) & (b ? ((1u << b) - 1) : 0u);      // pixels left of x (same word) of the other class
    while (m == 0) {
        if (wi == 0) return 0;
        wi--;
        w = rowbits[wi];
        m = cls ? ~w : w;
    }
    return wi * 32 + (31 - __clz(m)) + 1;
}

// root of v (FRAME = -1 when the tree hangs off the frame root), with path halving: a node is re-pointed at its grandparent,
// which is still an ancestor, so concurrent finds and unions stay correct; it keeps the chains short on speckle maps where
// one giant component collects tens of thousands of runs
__device__ __forceinline__ int uf_find(int *lab, int v) {
    while (v >= 0) {
        const int p = lab[v];
        if (p == v) break;
        if (p >= 0) {
            const int gp = lab[p];
            if (gp != p) lab[v] = gp;
        }
        v = p;
    }
    return v;
}

// read-only root (the flatten pass: every thread stores the FINAL root of its own run starts, and a path-halving write of a
// concurrent find -- parent read before that store, written after it -- would put a stale ancestor back)
__device__ __forceinline__ int uf_root(const int *lab, int v) {
    while (v >= 0) {
        // (round 6: a CACHED load -- workgroup scope.  The agent-scope form went past this XCD's L2 for every step of every chase; nothing here
        // needs another XCD's latest store: the links were made by earlier launches, and a label a concurrent lane has already flattened or
        // not yet flattened leads to the same root)
        const int p = __hip_atomic_load(&lab[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == v) break;
        v = p;
    }
    return v;
}

__device__ __forceinline__ void uf_union(int *lab, int a, int b) {
    for (;;) {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // a > b, so a >= 0
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

// The three CC kernels work on 32-pixel bitmap words (one thread per word) and touch `labels` only at RUN STARTS: a
// run (maximal horizontal stretch of one class) is a union-find node named by the linear index of its first pixel; the
// label entries of all other pixels are never written or read.  Every union below links two run starts, larger root under
// smaller, so a component's root is its raster-first pixel -- exactly where Suzuki's scan starts that border.
struct WordCtx {
    unsigned w, valid, starts, ends;        // bits, in-image mask, run-start mask, run-end mask
};

__device__ __forceinline__ WordCtx word_ctx(const unsigned *row, int wi, const DbpostDims &d) {
    WordCtx c;
    c.w = row[wi];
    const int rem = d.W - wi * 32;
    c.valid = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1);
    const unsigned carry = wi ? row[wi - 1] >> 31 : 0u;
    c.starts = (c.w ^ ((c.w << 1) | carry)) & c.valid;
    if (wi == 0) c.starts |= 1u;                                   // x = 0 always starts a run
    const unsigned next = (wi + 1 < d.WW) ? (row[wi + 1] & 1u) : 0u;
    c.ends = (c.w ^ ((c.w >> 1) | (next << 31))) & c.valid;
    if (rem <= 32) c.ends |= 1u << (rem - 1);                      // x = W-1 always ends a run
    return c;
}

// Two passes share these kernels.  Only the MAX_CAND borders with the largest start index are ever used (the reference keeps
// the first 1000 of findContours' bottom-up list), so pass A labels just the bottom STRIP_ROWS rows -- a run of the strip's
// first row that is linked to the row above joins the FRAME root (its component starts further up: not a start of the
// strip) -- and when an image has >= MAX_CAND starts there (speckle / noise maps: tens of thousands of components) pass B,
// the whole image, is skipped for it.  Clean maps have few components and go through both passes (pass A costs 9 %).
constexpr int STRIP_ROWS = 64;
struct CclPass { int y_first; const int *skip_if_full; const int *skip_if_few; int dbg; int row_step; };
// row_step: the merge kernel visits only the rows y_first + k row_step, k >= 1 (the slab boundaries of ccl_slab_kernel)   // skip_if_full: per-image start counts of pass A (pass B), skip_if_few: run starts of the strip (pass A), or null

__device__ __forceinline__ bool ccl_skip(const CclPass &ps, int img) {
    return (ps.skip_if_full && ps.skip_if_full[img] >= MAX_CAND) || (ps.skip_if_few && ps.skip_if_few[img] < MAX_CAND);
}

// label[s] = s for every run start s; clears the image's chunk counters (a pass that is skipped for an image leaves them alone)
__global__ __launch_bounds__(256) void ccl_init_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                       int *__restrict__ chunk_cnt, DbpostDims d, CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < d.nchunks) chunk_cnt[(long)img * d.nchunks + idx] = 0;       // both passes count their roots from zero
    if (idx >= (d.H - ps.y_first) * d.WW) return;
    const int y = ps.y_first + idx / d.WW, wi = idx % d.WW;
    const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
    int *lab = labels + (long)img * d.HW;
    unsigned m = word_ctx(row, wi, d).starts;
    const int base = y * d.W + wi * 32;
    while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        lab[base + i] = base + i;
    }
}

// merges runs of adjacent rows; only run boundaries issue unions (the rules are those of a per-pixel scan, evaluated
// on bit masks): vertical links where one of the two runs starts; for foreground (8-connected) the NW link at a run
// start and the NE link at a run end, when the pixel straight above is background; background is 4-connected and every
// background run touching the image border is united with the virtual FRAME root.
// one bitmap word's links of row y to the row above (and to FRAME); `part`: the bits of the word this thread takes
__device__ __forceinline__ void merge_word(const unsigned *__restrict__ bits, int *__restrict__ labels, const DbpostDims &d, int img, int y, int wi,
                                           bool cut, unsigned part, int dbg, bool up_links = true) {
    int dummy = 0;
#define MU(l, x, y) do { if (dbg & 1) dummy += (x) ^ (y); else if (dbg & 2) dummy += uf_find(l, x) ^ uf_find(l, y); else uf_union(l, x, y); } while (0)
    const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
    int *lab = labels + (long)img * d.HW;
    const WordCtx c = word_ctx(row, wi, d);
    const int x0 = wi * 32, base = y * d.W + x0;
    auto rs_cur = [&](int i) { return (c.starts >> i) & 1u ? base + i : y * d.W + run_start(row, x0 + i); };
    // background touching the image border belongs to the frame component
    {
        unsigned m = ~c.w & c.valid;
        unsigned f = (y == 0 || y == d.H - 1) ? (m & c.starts) : 0u;
        if (wi == 0) f |= m & 1u;
        if (x0 + 32 >= d.W) f |= m & (1u << (d.W - 1 - x0));
        f &= part;
        while (f) {
            const int i = __ffs(f) - 1;
            f &= f - 1;
            MU(lab, rs_cur(i), FRAME);
        }
    }
    if (y == 0 || !up_links) return;
    const unsigned *up = row - d.WW;
    const WordCtx u = word_ctx(up, wi, d);
    const int ubase = base - d.W;
    auto rs_up = [&](int x) { return (y - 1) * d.W + run_start(up, x); };
    // vertical links: same class above, and the current or the upper run starts here
    unsigned v = ~(c.w ^ u.w) & c.valid & (c.starts | u.starts) & part;
    while (v) {
        const int i = __ffs(v) - 1;
        v &= v - 1;
        MU(lab, rs_cur(i), cut ? FRAME : ((u.starts >> i) & 1u ? ubase + i : rs_up(x0 + i)));
    }
    // foreground with background straight above: diagonal links
    const unsigned fgbg = c.w & ~u.w & c.valid;
    const unsigned ucarry = wi ? up[wi - 1] >> 31 : 0u;
    unsigned nw = fgbg & c.starts & ((u.w << 1) | ucarry) & part;         // pix(up, x-1) set; x = 0 has no NW neighbour (carry 0)
    while (nw) {
        const int i = __ffs(nw) - 1;
        nw &= nw - 1;
        MU(lab, base + i, cut ? FRAME : rs_up(x0 + i - 1));
    }
    const unsigned unext = (wi + 1 < d.WW) ? (up[wi + 1] & 1u) : 0u;
    unsigned ne = fgbg & c.ends & ((u.w >> 1) | (unext << 31)) & part;     // pix(up, x+1) set (bits beyond W are clear)
    while (ne) {
        const int i = __ffs(ne) - 1;
        ne &= ne - 1;
        MU(lab, rs_cur(i), cut ? FRAME : ubase + i + 1);      // up(x) = 0, up(x+1) = 1: a run start
    }
    if ((dbg & 3) && dummy == 0x7fffffff) lab[0] = dummy;
#undef MU
}

template <int TPW>                                              // threads per word: 4 on noise maps, 1 otherwise (see below)
__global__ __launch_bounds__(256) void ccl_merge_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels, DbpostDims d,
                                                        CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    // speckle maps put ~10 unions into a word, each a chain of atomics: four threads per word there, one byte of its boundary masks
    // each.  A text-like map has a union in one word of ten, and walking its 30 000 words with 120 000 threads was half of the
    // kernel's time (38 of 79 us with every union compiled out): one thread per word there.
    const int idx = TPW == 4 ? (blockIdx.x * 256 + threadIdx.x) >> 2 : blockIdx.x * 256 + threadIdx.x;
    const unsigned part = TPW == 4 ? 0xffu << (8 * (threadIdx.x & 3)) : 0xffffffffu;
    int y, wi;
    if (ps.row_step) {                                            // slab boundaries only
        y = ps.y_first + (idx / d.WW + 1) * ps.row_step; wi = idx % d.WW;
        if (y >= d.H) return;
    } else {
        if (idx >= (d.H - ps.y_first) * d.WW) return;
        y = ps.y_first + idx / d.WW; wi = idx % d.WW;
    }
    merge_word(bits, labels, d, img, y, wi, y == ps.y_first && y > 0, part, ps.dbg);
}

// The slab-boundary pass of the text route, one LINK per lane (round 4): a wave takes 64 words of the boundary rows, publishes per kind
// (frame, vertical, north-west, north-east) the bits that carry a link, and walks all links of its 64 words together -- a union is two
// root chases and an atomic in global memory, and the ragged edges of the scene checkpoint's maps put a dozen of them into a word
// (one thread per word: 69 us).  Same rules as merge_word.
__global__ __launch_bounds__(256) void ccl_merge_rows_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels, DbpostDims d, CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nrows = (d.H - 1 - ps.y_first) / ps.row_step;      // boundary rows y_first + k row_step, k = 1 .. nrows
    const int idx0 = blockIdx.x * 256 + wv * 64, nwords = nrows * d.WW;
    if (idx0 >= nwords) return;                                  // whole waves
    const int idx = idx0 + lane;
    const bool on = idx < nwords;
    const int y = ps.y_first + ((on ? idx : idx0) / d.WW + 1) * ps.row_step, wi = (on ? idx : idx0) % d.WW;
    const unsigned *bimg = bits + (long)img * d.H * d.WW;
    int *lab = labels + (long)img * d.HW;
    __shared__ WaveItems items[4][4];                           // [wave][kind of link]
    __shared__ unsigned s_cs[4][64], s_us[4][64];
    const unsigned *row = bimg + (long)y * d.WW, *up = row - d.WW;   // y >= row_step > 0
    WordCtx c = {0u, 0u, 0u, 0u}, u = {0u, 0u, 0u, 0u};
    if (on) { c = word_ctx(row, wi, d); u = word_ctx(up, wi, d); }
    s_cs[wv][lane] = c.starts; s_us[wv][lane] = u.starts;
    const int x0 = wi * 32;
    // the four kinds of links of this word (merge_word's masks)
    unsigned m_f = 0, m_v = 0, m_nw = 0, m_ne = 0;
    if (on) {
        const unsigned bg = ~c.w & c.valid;
        m_f = (y == d.H - 1) ? (bg & c.starts) : 0u;
        if (wi == 0) m_f |= bg & 1u;
        if (x0 + 32 >= d.W) m_f |= bg & (1u << (d.W - 1 - x0));
        m_v = ~(c.w ^ u.w) & c.valid & (c.starts | u.starts);
        const unsigned fgbg = c.w & ~u.w & c.valid;
        const unsigned ucarry = wi ? up[wi - 1] >> 31 : 0u;
        m_nw = fgbg & c.starts & ((u.w << 1) | ucarry);
        const unsigned unext = (wi + 1 < d.WW) ? (up[wi + 1] & 1u) : 0u;
        m_ne = fgbg & c.ends & ((u.w >> 1) | (unext << 31));
    }
    // (round 6) ALL links of the wave's 64 words in ONE walk, one link per lane whatever its kind: the four kinds used to be four walks one
    // after the other, each a union's chain of dependent loads and an atomic (~4 round trips) -- the unions commute, their order is free
    const int G0 = wave_items_publish(items[wv][0], m_f), G1 = G0 + wave_items_publish(items[wv][1], m_v);
    const int G2 = G1 + wave_items_publish(items[wv][2], m_nw), G = G2 + wave_items_publish(items[wv][3], m_ne);
    for (int base = 0; base < G; base += 64) {
        const int it = base + lane;
        if (it < G) {
            const int kind = it < G0 ? 0 : it < G1 ? 1 : it < G2 ? 2 : 3;
            int L, i;
            wave_item(items[wv][kind], it - (kind == 0 ? 0 : kind == 1 ? G0 : kind == 2 ? G1 : G2), L, i);
            const int idxL = idx0 + L;
            const int yL = ps.y_first + (idxL / d.WW + 1) * ps.row_step, wL = idxL % d.WW, xL = wL * 32;
            const unsigned *rowL = bimg + (long)yL * d.WW, *upL = rowL - d.WW;
            const unsigned cs = s_cs[wv][L], us = s_us[wv][L];
            int a, b;
            if (kind == 2) { a = yL * d.W + xL + i; b = (yL - 1) * d.W + run_start(upL, xL + i - 1); }
            else {
                a = (cs >> i) & 1u ? yL * d.W + xL + i : yL * d.W + run_start(rowL, xL + i);      // run start of the current row at bit i
                if (kind == 0) b = FRAME;
                else if (kind == 1) b = (us >> i) & 1u ? (yL - 1) * d.W + xL + i : (yL - 1) * d.W + run_start(upL, xL + i);
                else b = (yL - 1) * d.W + xL + i + 1;              // up(x) = 0, up(x + 1) = 1: a run start
            }
            uf_union(lab, a, b);
        }
    }
}

// Slab labelling (text route, late round 3): one workgroup labels SLAB_ROWS rows in LDS -- its runs numbered in raster order by a
// prefix sum over the words' run starts, the same union rules as ccl_merge_kernel on LDS union-find nodes (node 0 = FRAME) -- and
// writes label[run start] = the slab-local root's pixel (or FRAME).  ccl_merge_kernel then links only the slab boundary rows in global
// memory (row_step), and the flatten pass resolves the short chains that leaves.  Replaces ccl_init_kernel + ccl_merge_kernel<1>
// (6 + 62 us per 32 text-like maps: every union two pointer chases and an atomic in global memory).  A slab with more than
// SLAB_RUNS runs (speckle) is labelled by its workgroup through the global union-find instead.
#ifndef PT_SLAB_ROWS
#define PT_SLAB_ROWS 8
#endif
constexpr int SLAB_ROWS = PT_SLAB_ROWS;         // 8 or 16 (the slab kernel runs one thread per word: 16 x 64 words = 1024 threads at most)
constexpr int SLAB_RUNS = 4096;
__device__ __forceinline__ int lds_find(int *par, int v) {
    for (;;) {
        const int p = par[v];
        if (p == v) return v;
        const int gp = par[p];
        if (gp != p) par[v] = gp;
        v = p;
    }
}
__device__ __forceinline__ void lds_union(int *par, int a, int b) {
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}
// maps != null (round 4; W % 32 == 0): the slab kernel thresholds its own rows -- `pred > thresh` straight from the probability map,
// coalesced 16-byte loads, nibbles of eight lanes combined into a word -- writes the bitmap words for the later kernels and counts the
// run starts of the bottom strip: binarize_flat_kernel's launch (21 us at 5.4 TB/s, then 4 MB read back here) is gone from the text route.
__global__ __launch_bounds__(SLAB_ROWS * 64) void ccl_slab_kernel(unsigned *__restrict__ bits, int *__restrict__ labels, int *__restrict__ chunk_cnt,
                                                       DbpostDims d, CclPass ps, const float *__restrict__ maps, float thresh,
                                                       int *__restrict__ strip_runs, int count_from_row) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    __shared__ unsigned sb[SLAB_ROWS * 64];                        // the slab's bitmap words (WW <= 64)
    __shared__ int sbase[SLAB_ROWS * 64 + 1];                      // run starts in the words before this one
    __shared__ int wsum[SLAB_ROWS];
    __shared__ int par[SLAB_RUNS + 1];
    __shared__ unsigned short rpix[SLAB_RUNS];                     // run -> its first pixel, relative to the slab (< 16 * 2048)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y0 = ps.y_first + blockIdx.x * SLAB_ROWS;
    const int rows = min(SLAB_ROWS, d.H - y0);
    const int nw = rows * d.WW;
    if (blockIdx.x == 0)
        for (int i = tid; i < d.nchunks; i += blockDim.x) chunk_cnt[(long)img * d.nchunks + i] = 0;
    unsigned *bimg = bits + ((long)img * d.H + y0) * d.WW;
    if (maps) {
        // the slab's rows are one contiguous stretch of the map: quad q (four pixels) belongs to word q / 8; blockDim.x >= nw is a
        // multiple of 8, so a lane keeps its place in its word from round to round and eight rounds cover the slab
        const float4 *src = reinterpret_cast<const float4 *>(maps + (long)img * d.HW + (long)y0 * d.W);
        const int nq = nw * 8;
        float4 v4[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int q = tid + k * (int)blockDim.x; v4[k] = q < nq ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int q = tid + k * (int)blockDim.x;
            const unsigned nib = (v4[k].x > thresh) | ((v4[k].y > thresh) << 1) | ((v4[k].z > thresh) << 2) | ((v4[k].w > thresh) << 3);
            unsigned v = nib << (4 * (tid & 7));
            v |= __shfl_xor(v, 1);
            v |= __shfl_xor(v, 2);
            v |= __shfl_xor(v, 4);
            if (q < nq && (tid & 7) == 0) { sb[q >> 3] = v; bimg[q >> 3] = v; }
        }
    } else if (tid < nw) sb[tid] = bimg[tid];
    __syncthreads();
    const int yl = tid / d.WW, wi = tid - yl * d.WW;              // tid >= nw: idle along (barriers)
    const bool on = tid < nw;
    WordCtx c = {0u, 0u, 0u, 0u};
    if (on) c = word_ctx(sb + yl * d.WW, wi, d);
    if (maps && strip_runs && count_from_row > 0 && y0 + rows > count_from_row) {      // (uniform) run starts of both polarities in the bottom strip,
        int cs = 0;                                             // column 0 against the background frame: binarize_flat_kernel's count
        if (on && y0 + yl >= count_from_row) cs = __popc(c.starts) - ((wi == 0 && !(c.w & 1u)) ? 1 : 0);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cs += __shfl_xor(cs, o);
        if (lane == 0 && cs) atomicAdd(&strip_runs[img], cs);
    }
    // exclusive prefix of the run-start counts over the slab's words
    int cnt = __popc(c.starts), incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { if (w < wave) wbase += wsum[w]; total += wsum[w]; }
    const int base = wbase + incl - cnt;
    if (total > SLAB_RUNS) {                                       // uniform
        // a slab too dense for the LDS tables (speckle): the same result through the global union-find, by this workgroup alone --
        // label[s] = s for its run starts, then the links of its rows (the first row's upward links are the boundary pass's)
        int *lab = labels + (long)img * d.HW;
        if (on) {
            unsigned m = c.starts;
            const int b0 = (y0 + yl) * d.W + wi * 32;
            while (m) {
                const int i = __ffs(m) - 1;
                m &= m - 1;
                lab[b0 + i] = b0 + i;
            }
        }
        __threadfence();
        __syncthreads();
        if (on) {
            const int y = y0 + yl;
            const bool cut = yl == 0 && y == ps.y_first && y > 0;
            merge_word(bits, labels, d, img, y, wi, cut, 0xffffffffu, 0, yl > 0 || cut);
        }
        return;
    }
    if (on) sbase[tid] = base;
    for (int i = tid; i <= total; i += blockDim.x) par[i] = i;
    {
        unsigned m = c.starts; int k = 0;
        while (m) {
            const int i = __ffs(m) - 1;
            m &= m - 1;
            rpix[base + k] = (unsigned short)(yl * d.W + wi * 32 + i);
            k++;
        }
    }
    __syncthreads();
    // node of the run that holds pixel x of slab row r: the number of run starts at or before it in raster order (x = 0 always starts one)
    auto node = [&](int r, int x) {
        const int wq = r * d.WW + (x >> 5);
        const unsigned st = word_ctx(sb + r * d.WW, x >> 5, d).starts;
        return sbase[wq] + __popc(st & (0xffffffffu >> (31 - (x & 31))));
    };
    if (on) {
        const int y = y0 + yl, x0 = wi * 32;
        auto ncur = [&](int i) { return base + __popc(c.starts & (0xffffffffu >> (31 - i))); };
        {   // background touching the image border belongs to the frame component
            unsigned m = ~c.w & c.valid;
            unsigned f = (y == 0 || y == d.H - 1) ? (m & c.starts) : 0u;
            if (wi == 0) f |= m & 1u;
            if (x0 + 32 >= d.W) f |= m & (1u << (d.W - 1 - x0));
            while (f) {
                const int i = __ffs(f) - 1;
                f &= f - 1;
                lds_union(par, ncur(i), 0);
            }
        }
        const bool cut = yl == 0 && y == ps.y_first && y > 0;       // first row of the strip: links upwards end in FRAME
        if (yl > 0 || cut) {
            // the row above: inside the slab, or (cut) the global row above the strip, read for its bits only
            const unsigned *uprow = cut ? bits + ((long)img * d.H + y - 1) * d.WW : sb + (yl - 1) * d.WW;
            const WordCtx u = word_ctx(uprow, wi, d);
            unsigned v = ~(c.w ^ u.w) & c.valid & (c.starts | u.starts);
            while (v) {
                const int i = __ffs(v) - 1;
                v &= v - 1;
                lds_union(par, ncur(i), cut ? 0 : node(yl - 1, x0 + i));
            }
            const unsigned fgbg = c.w & ~u.w & c.valid;
            const unsigned ucarry = wi ? uprow[wi - 1] >> 31 : 0u;
            unsigned nwm = fgbg & c.starts & ((u.w << 1) | ucarry);
            while (nwm) {
                const int i = __ffs(nwm) - 1;
                nwm &= nwm - 1;
                lds_union(par, ncur(i), cut ? 0 : node(yl - 1, x0 + i - 1));
            }
            const unsigned unext = (wi + 1 < d.WW) ? (uprow[wi + 1] & 1u) : 0u;
            unsigned ne = fgbg & c.ends & ((u.w >> 1) | (unext << 31));
            while (ne) {
                const int i = __ffs(ne) - 1;
                ne &= ne - 1;
                lds_union(par, ncur(i), cut ? 0 : node(yl - 1, x0 + i + 1));
            }
        }
    }
    __syncthreads();
    int *lab = labels + (long)img * d.HW + (long)y0 * d.W;
    for (int r = tid; r < total; r += blockDim.x) {
        const int root = lds_find(par, r + 1);
        lab[rpix[r]] = root == 0 ? FRAME : y0 * d.W + (int)rpix[root - 1];
    }
}

// label[s] = root for every run start s; word_lab[word] = root of the run that covers bit 0 of the word (so that the root of
// ANY pixel is two loads away: the run start inside its word, or the word's entry); counts border starts (roots) per
// 1024-pixel chunk (chunk_cnt zeroed by the host)
constexpr int ROOT_K = 16;               // roots of a 1024-pixel chunk the flatten pass lists for rank_starts_kernel (more: the chunk is re-scanned)
__global__ __launch_bounds__(256) void ccl_flatten_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                          int *__restrict__ word_lab, int *__restrict__ chunk_cnt, int *__restrict__ chunk_roots,
                                                          DbpostDims d, CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx0 = blockIdx.x * 256 + wv * 64, nwords = (d.H - ps.y_first) * d.WW;
    if (idx0 >= nwords) return;                                  // whole waves
    const int idx = idx0 + lane;
    const bool on = idx < nwords;
    const int y = ps.y_first + (on ? idx : idx0) / d.WW, wi = (on ? idx : idx0) % d.WW;
    const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
    int *lab = labels + (long)img * d.HW;
    const unsigned starts = on ? word_ctx(row, wi, d).starts : 0u;
    __shared__ WaveItems items[4];
    __shared__ int s_r0[4][64], s_rlast[4][64];                 // root of the run that starts at bit 0 of the word / at the word's LAST run start
    const int G = wave_items_publish(items[wv], starts);
    // one run start per lane (a word of a ragged edge holds a dozen, each a chase to its root)
    for (int base = 0; base < G; base += 64) {
        const int it = base + lane;
        if (it < G) {
            int L, i;
            wave_item(items[wv], it, L, i);
            const int idxL = idx0 + L, yL = ps.y_first + idxL / d.WW, wL = idxL - (idxL / d.WW) * d.WW;
            const int s = yL * d.W + wL * 32 + i;
            const int r = uf_root(lab, s);
            lab[s] = r;
            if (i == 0) s_r0[wv][L] = r;
            if ((items[wv].mask[L] >> i) <= 1u) s_rlast[wv][L] = r;     // no run start above bit i in this word
            if (r == s) {                                        // a border start: counted and listed per chunk
                const long ch = (long)img * d.nchunks + s / CHUNK;
                const int pos = atomicAdd(&chunk_cnt[ch], 1);
                if (pos < ROOT_K) chunk_roots[ch * ROOT_K + pos] = s;
            }
        }
    }
    wave_lds_sync();
    if (on) {
        // The run that covers bit 0 of a word starts at the LAST run start of the nearest word to the left that has one (a row's first word
        // always has: x = 0 starts a run, so the search never leaves the row) -- whose root the loop above has just found.  Round 6: taken
        // from there (one ballot, one LDS read).  It used to be a walk to the left through the row's words, a dependent load per word, and
        // then the label chase again -- for every background word of the map, i.e. most of them, all chasing the same few roots.  Only a
        // word whose run reaches in from before the wave's first word still walks.
        const unsigned long long has = __ballot(starts != 0u) & ((1ull << lane) - 1ull);
        int r0;
        if (starts & 1u) r0 = s_r0[wv][lane];
        else if (has) r0 = s_rlast[wv][63 - __clzll((long long)has)];
        else r0 = uf_root(lab, y * d.W + run_start(row, wi * 32));
        word_lab[((long)img * d.H + y) * d.WW + wi] = r0;
    }
}

// suffix sums over chunks (one block per image): chunk_cnt[c] := number of starts in chunks > c; total per image
__global__ __launch_bounds__(1024) void chunk_suffix_kernel(int *__restrict__ chunk_cnt, int *__restrict__ totals, DbpostDims d,
                                                            CclPass ps, int *__restrict__ strip_totals) {
    const int img = blockIdx.x;
    if (ccl_skip(ps, img)) return;                               // pass A was enough: its suffix sums and total stand
    int *cc = chunk_cnt + (long)img * d.nchunks;
    __shared__ int part[1024];
    // each thread owns a contiguous slice of chunks, highest chunks first
    const int per = cdiv(d.nchunks, 1024);
    const int hi = d.nchunks - threadIdx.x * per;             // exclusive upper bound of my slice
    const int lo = hi - per > 0 ? hi - per : 0;
    int s = 0;
    for (int c = hi - 1; c >= lo && c >= 0; c--) s += cc[c];
    part[threadIdx.x] = s;
    __syncthreads();
    // exclusive scan of part[] (thread 0 holds the highest slice)
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;                           // starts in all higher slices
    for (int c = hi - 1; c >= lo && c >= 0; c--) {
        const int v = cc[c];
        cc[c] = run;
        run += v;
    }
    if (threadIdx.x == 1023) { totals[img] = part[1023]; if (!ps.skip_if_full) strip_totals[img] = part[1023]; }
}

// candidate k (0 = bottom-most start) of an image: trigger pixel and kind
struct Cand { int p; int is_hole; };
// what the border-state pass accumulates per candidate
struct Acc { int nstates, npts, xmin, xmax, ymin, ymax, cursor, off; };

// Ranks the roots (border starts) of an image from the bottom: candidate k = the start with the k-th largest pixel index.
// A selected root r is MARKED in the label image, lab[r] = -2 - k, which is how the border-state pass finds a component's
// candidate (roots that are not selected keep lab[r] == r; FRAME stays -1).
__global__ __launch_bounds__(256) void select_starts_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                            const int *__restrict__ chunk_after, const int *__restrict__ totals,
                                                            Cand *__restrict__ cands, Acc *__restrict__ acc, DbpostDims d) {
    const int img = blockIdx.y;
    __shared__ int wave_cnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *lab = labels + (long)img * d.HW;
    // eight chunks per block (29 000 blocks of which a few hundred had work were most of the kernel's 19 us)
    for (int ci = 0; ci < SEL_CHUNKS; ci++) {
    const int chunk = blockIdx.x * SEL_CHUNKS + ci;
    if (chunk >= d.nchunks) break;
    const int after = chunk_after[(long)img * d.nchunks + chunk];
    if (after >= MAX_CAND) continue;                            // every start here ranks beyond the first 1000
    const int upto = chunk ? chunk_after[(long)img * d.nchunks + chunk - 1] : totals[img];
    if (upto == after) continue;                                // no start in this chunk (most chunks of a clean map)
    // the chunk's 1024 pixels in one sweep, highest pixel first: slice k (k = 3 .. 0) holds pixels [256 k, 256 k + 256), thread 0 its
    // highest.  All four label loads of a thread are issued before the first is used, one barrier in all (the first version walked
    // the slices one after the other: four dependent round trips and twelve barriers, 25 us).
    bool is[4];
    int rx[4], ry[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long p = (long)chunk * CHUNK + k * 256 + (255 - threadIdx.x);
        is[k] = false; rx[k] = ry[k] = 0;
        if (p < d.HW) {                                         // only run starts carry labels
            const int y = (int)(p / d.W), x = (int)(p - (long)y * d.W);
            const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
            ry[k] = y; rx[k] = x;
            is[k] = (x == 0 || pix(row, x) != pix(row, x - 1)) && lab[p] == (int)p;
        }
    }
    unsigned long long m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        m[k] = __ballot(is[k]);
        if (lane == 0) wave_cnt[k][wave] = __popcll(m[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        if (!is[k]) continue;
        int before = 0;
        for (int kk = 3; kk > k; kk--) before += wave_cnt[kk][0] + wave_cnt[kk][1] + wave_cnt[kk][2] + wave_cnt[kk][3];
        for (int w = 0; w < wave; w++) before += wave_cnt[k][w];
        before += __popcll(m[k] & ((1ull << lane) - 1));
        const int rank = after + before;
        if (rank < MAX_CAND) {
            const long p = (long)chunk * CHUNK + k * 256 + (255 - threadIdx.x);
            const unsigned *row = bits + ((long)img * d.H + ry[k]) * d.WW;
            Cand c; c.p = (int)p; c.is_hole = !pix(row, rx[k]);
            cands[(long)img * MAX_CAND + rank] = c;
            Acc a; a.nstates = 0; a.npts = 0; a.xmin = 0x7fffffff; a.xmax = -1; a.ymin = 0x7fffffff; a.ymax = -1; a.cursor = 0; a.off = -1;
            acc[(long)img * MAX_CAND + rank] = a;
            lab[p] = -2 - rank;
        }
    }
    __syncthreads();                                           // wave_cnt is reused by the next chunk
    }
}

// Text route (one labelling pass): chunk_suffix_kernel + select_starts_kernel as ONE launch of one block per image (round 4: the pair
// was 5 + 19 us of dependent launches, the second one 15 000 small workgroups of which a few hundred had work).  The flatten pass has
// listed the roots of every chunk (up to ROOT_K): a thread owns a slice of chunks, takes the suffix sums and ranks the listed roots of its
// chunks itself; a chunk with more roots than the list holds (speckle) is re-scanned by the whole block.  Same results as the pair:
// candidate k = the root with the k-th largest pixel index, lab[root] = -2 - k, acc[k] initialised, totals.
__global__ __launch_bounds__(1024) void rank_starts_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                           const int *__restrict__ chunk_cnt, const int *__restrict__ chunk_roots,
                                                           int *__restrict__ totals, int *__restrict__ strip_totals,
                                                           Cand *__restrict__ cands, Acc *__restrict__ acc, DbpostDims d) {
    const int img = blockIdx.x, tid = threadIdx.x;
    const int *cc = chunk_cnt + (long)img * d.nchunks;
    int *lab = labels + (long)img * d.HW;
    __shared__ int part[1024];
    __shared__ int big[128][2], big_n;                           // chunks to re-scan: (chunk, starts in higher chunks)
    __shared__ int wcnt[16];
    if (tid == 0) big_n = 0;
    const int per = cdiv(d.nchunks, 1024);
    const int hi = d.nchunks - tid * per;                        // exclusive upper bound of my slice (thread 0: the highest chunks)
    const int lo = hi - per > 0 ? hi - per : 0;
    int s = 0;
    for (int c = hi - 1; c >= lo && c >= 0; c--) s += cc[c];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid == 1023) { totals[img] = part[1023]; strip_totals[img] = part[1023]; }
    auto take = [&](int p, int rank) {                           // root p is candidate `rank`
        const int y = p / d.W, x = p - y * d.W;
        Cand c; c.p = p; c.is_hole = !pix(bits + ((long)img * d.H + y) * d.WW, x);
        cands[(long)img * MAX_CAND + rank] = c;
        Acc a; a.nstates = 0; a.npts = 0; a.xmin = 0x7fffffff; a.xmax = -1; a.ymin = 0x7fffffff; a.ymax = -1; a.cursor = 0; a.off = -1;
        acc[(long)img * MAX_CAND + rank] = a;
        lab[p] = -2 - rank;
    };
    int after = part[tid] - s;                                  // starts in all higher slices
    for (int c = hi - 1; c >= lo && c >= 0 && after < MAX_CAND; c--) {
        const int cnt = cc[c];
        if (cnt > ROOT_K) {
            const int e = atomicAdd(&big_n, 1);
            if (e < 128) { big[e][0] = c; big[e][1] = after; }  // (at most MAX_CAND / ROOT_K + 1 = 63 such chunks can matter)
        } else if (cnt > 0) {
            // the chunk's roots into registers in one go (a ragged map lists ten in a chunk: ranking them with a load per comparison was
            // a hundred dependent loads), then rank = roots of the chunk with a larger index
            const int *roots = chunk_roots + ((long)img * d.nchunks + c) * ROOT_K;
            int rt[ROOT_K];
#pragma unroll
            for (int j = 0; j < ROOT_K; j++) rt[j] = j < cnt ? roots[j] : -1;
#pragma unroll
            for (int j = 0; j < ROOT_K; j++) {
                int before = 0;
#pragma unroll
                for (int i = 0; i < ROOT_K; i++) before += rt[i] > rt[j] ? 1 : 0;
                if (j < cnt && after + before < MAX_CAND) take(rt[j], after + before);
            }
        }
        after += cnt;
    }
    __syncthreads();
    const int nbig = big_n < 128 ? big_n : 128;
    for (int e = 0; e < nbig; e++) {                             // the block scans the chunk's 1024 pixels, highest pixel first
        const int chunk = big[e][0], aft = big[e][1];
        const long p = (long)chunk * CHUNK + (1023 - tid);
        bool is = false;
        if (p < d.HW) {
            const int y = (int)(p / d.W), x = (int)(p - (long)y * d.W);
            const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
            is = (x == 0 || pix(row, x) != pix(row, x - 1)) && lab[p] == (int)p;      // only run starts carry labels
        }
        const unsigned long long m = __ballot(is);
        if ((tid & 63) == 0) wcnt[tid >> 6] = __popcll(m);
        __syncthreads();
        if (is) {
            int before = __popcll(m & ((1ull << (tid & 63)) - 1));
            for (int w = 0; w < (tid >> 6); w++) before += wcnt[w];
            if (aft + before < MAX_CAND) take((int)p, aft + before);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ border states
// direction k (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE) -> step, from 2-bit fields (value + 1) so that no table load is needed
__device__ __forceinline__ int dir_dx(int k) { return (int)((0x901Au >> (2 * k)) & 3u) - 1; }    // {1,1,0,-1,-1,-1,0,1}
__device__ __forceinline__ int dir_dy(int k) { return (int)((0xA901u >> (2 * k)) & 3u) - 1; }    // {0,-1,-1,-1,0,1,1,1}

// a label entry read at run start rs -> component root (FRAME = -1); a marked root (-2 - k) names itself
__device__ __forceinline__ int canon_root(int v, int rs) { return v >= 0 ? v : (v == FRAME ? FRAME : rs); }

__device__ __forceinline__ int root_of_pixel(const unsigned *bimg, const int *lab, const int *wl, const DbpostDims &d, int x, int y) {
    const int wi = x >> 5, b = x & 31;
    const unsigned w = bimg[(long)y * d.WW + wi];
    const unsigned m = (((w >> b) & 1u) ? ~w : w) & ((1u << b) - 1u);      // pixels of the other class left of x in this word
    if (m) {
        const int rs = y * d.W + wi * 32 + (32 - __clz(m));
        return canon_root(lab[rs], rs);
    }
    return wl[(long)y * d.WW + wi];
}

__device__ __forceinline__ int cand_of_root(const int *lab, int r) {
    if (r < 0) return -1;
    const int v = lab[r];
    return v <= -2 ? -2 - v : -1;
}

// state word: x (11 bits) | y (15 bits) << 11 | s_out << 26 | s_in << 29
__device__ __forceinline__ int st_x(unsigned s) { return (int)(s & 0x7ffu); }
__device__ __forceinline__ int st_y(unsigned s) { return (int)((s >> 11) & 0x7fffu); }
__device__ __forceinline__ int st_out(unsigned s) { return (int)((s >> 26) & 7u); }
__device__ __forceinline__ int st_in(unsigned s) { return (int)(s >> 29); }

// staging list of the border states: a record is (state word, candidate | table slot << 10 | length << 16) -- length > 0: a stretch of straight
// horizontal states starting at that word, 0: one state; table slot: where the tile's wave booked the candidate (sg.tab); the records of an 8-row x 8-word tile lie behind each other in the tile's
// fixed slice, their number in the tile's header word
struct StageArgs2 { uint2 *rec; int *hdr; long cap; int *tab; };     // tab: per tile, the wave's table of border_states_kernel as it stood at the end -- 64 candidates (-1: free slot), 64 state counts

constexpr int STAGE_TILE = 8192;                // record slots per tile (a pixel has at most four gaps: no reservation, no overflow)

// A wave's table of the borders it met (LDS): candidate -> slot by k & 63 with linear probing; what is booked there goes to the border's
// record in global memory ONCE per wave (every direct booking is atomics on one line that all the waves of that border share).
constexpr int WT_SLOTS = 64;
// (round 6: n = states | contour points << 16 of the tile's share -- one add per state instead of two; a tile holds < 2^14 states)
struct WaveTable { int tag[WT_SLOTS], n[WT_SLOTS], x0[WT_SLOTS], x1[WT_SLOTS], y0[WT_SLOTS], y1[WT_SLOTS]; };
// slot of candidate k (claiming a free one), or -1 when the table is full of other candidates (a tile of speckle: more than 64 borders)
__device__ __forceinline__ int wt_slot(int *tag, int k) {
    int slot = k & (WT_SLOTS - 1);
    for (int probe = 0; probe < WT_SLOTS; probe++) {
        // (round 6: a look first -- a tag only ever goes from free to one candidate, so a slot seen with this candidate is this candidate's
        // for good and a slot seen with another one never will be: the CAS is for free slots only, once per border and wave instead of
        // once per state)
        const int cur = *reinterpret_cast<volatile int *>(&tag[slot]);
        if (cur == k) return slot;
        if (cur == -1) {
            const int t = atomicCAS(&tag[slot], -1, k);
            if (t == -1 || t == k) return slot;
        }
        slot = (slot + 1) & (WT_SLOTS - 1);
    }
    return -1;
}

// One WAVE per tile of 8 rows x 8 bitmap words lists the border states of the tile's foreground pixels (see the file header), finds each
// one's border (F, B -> candidate) and stages a (state, candidate) record; counts, contour points and bounding boxes per border go
// through the wave's table.  Two phases:
//  * the straight horizontal states (most of a text line's border) per word, a contiguous stretch per record;
//  * every other state ("generic" pixels: turns, ends, single pixels), ONE PIXEL PER LANE whatever word it sits in -- until round 3 a
//    lane walked the pixels of its own word one after the other, so a ragged edge (twenty such pixels in a word, each a hundred
//    dependent instructions and a look-up) kept one lane busy and sixty-three waiting: the stage's time on the scene checkpoint's maps.
#ifndef BS_DBG
#define BS_DBG 0            // timing experiments only (tools/dbg/bs_knock.sh): 1 no straight stretches (phase 1), 2 no generic pixels (phase 2)
#endif
__global__ __launch_bounds__(256) void border_states_kernel(const unsigned *__restrict__ bits, const int *__restrict__ labels,
                                                            const int *__restrict__ word_lab, const int *__restrict__ strip_totals,
                                                            Acc *__restrict__ acc, DbpostDims d, StageArgs2 sg) {
    const int img = blockIdx.y;
    const int y_first = (d.strip_y && strip_totals[img] >= MAX_CAND) ? d.strip_y : 0;     // rows above carry no labels (and no candidate)
    const int tiles_x = cdiv(d.WW, 8);
    const int wv = threadIdx.x >> 6, lane_t = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wv;
    const int ty0 = (tile / tiles_x) * 8 + (lane_t >> 3), tx0 = (tile % tiles_x) * 8 + (lane_t & 7);
    // strip mode: the row straight ABOVE the strip is listed too -- a hole in the strip's first row has border states on the foreground
    // pixels above it.  Those pixels carry no labels; only their gaps that reach the pixel below (direction S) can belong to a strip
    // border, and that border is the hole's whatever the pixel's own component is (it reaches above the strip, so it is no candidate).
    const int y_lo = y_first ? y_first - 1 : 0;
    const int tiles_y = cdiv(d.H - y_lo, 8);
    if (tile >= tiles_x * tiles_y) return;                      // whole waves (nothing below spans waves)
    const bool in_range = ty0 < d.H - y_lo && tx0 < d.WW;
    const int y = in_range ? y_lo + ty0 : y_lo, wi = in_range ? tx0 : 0;
    const unsigned *bimg = bits + (long)img * d.H * d.WW;
    const unsigned w = in_range ? bimg[(long)y * d.WW + wi] : 0u;
    auto ld = [&](int yy, int ww) -> unsigned {                  // (an all-background word has no border point: no neighbour loads)
        return (w && (unsigned)yy < (unsigned)d.H && (unsigned)ww < (unsigned)d.WW) ? bimg[(long)yy * d.WW + ww] : 0u;
    };
    const unsigned cl = ld(y, wi - 1), cr = ld(y, wi + 1);
    const unsigned up = ld(y - 1, wi), ul = ld(y - 1, wi - 1), ur = ld(y - 1, wi + 1);
    const unsigned dn = ld(y + 1, wi), dl = ld(y + 1, wi - 1), dr = ld(y + 1, wi + 1);
    // neighbour planes: bit i = the neighbour of pixel (32 wi + i) in that direction (outside the image = background)
    const unsigned pE = (w >> 1) | (cr << 31), pW = (w << 1) | (cl >> 31);
    const unsigned pNE = (up >> 1) | (ur << 31), pNW = (up << 1) | (ul >> 31);
    const unsigned pSE = (dn >> 1) | (dr << 31), pSW = (dn << 1) | (dl >> 31);
    const int *lab = labels + (long)img * d.HW;
    const int *wl = word_lab + (long)img * d.H * d.WW;
    Acc *ac = acc + (long)img * MAX_CAND;
    __shared__ WaveTable tabs[4];
    __shared__ unsigned s_pl[4][8][64];                         // the tile's neighbour planes (E NE N NW W SW S SE), word by word
    __shared__ unsigned s_w[4][64];
    __shared__ int s_pre[4][65];                                // exclusive prefix of the words' generic-pixel counts
    WaveTable &T = tabs[wv];
    T.tag[lane_t] = -1; T.n[lane_t] = 0; T.x0[lane_t] = 0x7fffffff; T.x1[lane_t] = -1; T.y0[lane_t] = 0x7fffffff; T.y1[lane_t] = -1;
    uint2 *rec = sg.rec + (long)img * sg.cap + (long)tile * STAGE_TILE;
    int tile_n = 0;                                             // records of the tile so far (wave-uniform)
    // books a state (or a stretch of n states) of border k and stages its record; every lane of the wave calls it together
    auto emit = [&](bool on, int k, unsigned state, int n, int np, int x0, int x1, int yy) {
        const unsigned long long bal = __ballot(on);
        if (on) {
            const int slot = wt_slot(T.tag, k);
            // record: state word | candidate (10 bits) + its slot in the wave's table (6 bits; no slot: 0, which then holds another candidate)
            // + stretch length (16 bits).  The scatter pass finds the border's cursor through the slot without probing.
            static_assert(MAX_CAND <= 1024 && WT_SLOTS == 64, "record layout");
            rec[tile_n + __popcll(bal & ((1ull << lane_t) - 1))] =
                make_uint2(state, (unsigned)k | ((unsigned)(slot >= 0 ? slot : 0) << 10) | ((unsigned)(n > 1 || np < 0 ? n : 0) << 16));
            const int pts = np < 0 ? 0 : np;
            if (slot >= 0) {
                atomicAdd(&T.n[slot], n | (pts << 16));
                atomicMin(&T.x0[slot], x0); atomicMax(&T.x1[slot], x1); atomicMin(&T.y0[slot], yy); atomicMax(&T.y1[slot], yy);
            } else {                                            // table full (speckle): straight to the border's record
                atomicAdd(reinterpret_cast<unsigned long long *>(&ac[k].nstates), (unsigned long long)(unsigned)n | ((unsigned long long)(unsigned)pts << 32));
                atomicMin(&ac[k].xmin, x0); atomicMax(&ac[k].xmax, x1); atomicMin(&ac[k].ymin, yy); atomicMax(&ac[k].ymax, yy);
            }
        }
        tile_n += __popcll(bal);
    };
    // component of the run of row yy that holds bit i of word (yy, ww) with bits wbits -> root (FRAME = -1); the run start inside the word
    // or the word's own label
    auto run_root = [&](int yy, int ww, unsigned wbits, int i) -> int {
        const unsigned mo = (((wbits >> i) & 1u) ? ~wbits : wbits) & ((1u << i) - 1u);     // pixels of the other class left of i
        if (mo) { const int rs = yy * d.W + ww * 32 + (32 - __clz(mo)); return canon_root(lab[rs], rs); }
        return wl[(long)yy * d.WW + ww];
    };
    // (surround, candidate) of foreground component F
    auto comp_info = [&](int F, int &S, int &kF) {
        if (F < 0) { S = -3; kF = -1; return; }                  // (strip mode) component reaches above the strip: never a candidate
        const int fy = F / d.W, fx = F - fy * d.W;
        S = fx == 0 ? FRAME : root_of_pixel(bimg, lab, wl, d, fx - 1, fy);
        kF = cand_of_root(lab, F);
    };
    const unsigned straight_t = w & pE & pW & ~pNE & ~up & ~pNW;
    const unsigned straight_b = w & pE & pW & ~pSW & ~dn & ~pSE;
    const unsigned generic = w & (((pE & ~pNE) & ~straight_t) | (pNE & ~up) | (up & ~pNW) | (pNW & ~pW) | ((pW & ~pSW) & ~straight_b) |
                                  (pSW & ~dn) | (dn & ~pSE) | (pSE & ~pE) | ~(pE | pNE | up | pNW | pW | pSW | dn | pSE));
    s_w[wv][lane_t] = w;
    s_pl[wv][0][lane_t] = pE; s_pl[wv][1][lane_t] = pNE; s_pl[wv][2][lane_t] = up; s_pl[wv][3][lane_t] = pNW;
    s_pl[wv][4][lane_t] = pW; s_pl[wv][5][lane_t] = pSW; s_pl[wv][6][lane_t] = dn; s_pl[wv][7][lane_t] = pSE;
    {
        const int g = in_range ? __popc(generic) : 0;
        int incl = g;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane_t >= o) incl += v; }
        s_pre[wv][lane_t + 1] = incl;
        if (lane_t == 0) s_pre[wv][0] = 0;
        // (keep the generic mask where the pixel phase finds it: plane slot of the word's own lane)
    }
    __shared__ unsigned s_gen[4][64];
    s_gen[wv][lane_t] = in_range ? generic : 0u;
    wave_lds_sync();
    // ---- phase 1: the straight horizontal states.  Along the top edge of a blob the walk runs west through pixels whose gap is
    // exactly {NE, N, NW} (s_in = E, s_out = W), along the bottom edge east through {SW, S, SE}; such states never turn, and a
    // contiguous stretch of them shares F, B and the candidate: one look-up, one record.
    {
        const bool above = y < y_first;
        int f_key = -2, S = -3, kF = -1;                        // cached per lane: run of the current pixel -> surround, candidate
#pragma unroll
        for (int side = 0; side < 2; side++) {
            unsigned M = (in_range && !(BS_DBG & 1)) ? (side ? straight_b : (above ? 0u : straight_t)) : 0u;
            const int zy = side ? y + 1 : y - 1;
            const unsigned rw = side ? dn : up;
            const unsigned code = side ? ((0u << 26) | (4u << 29)) : ((4u << 26) | (0u << 29));        // s_out << 26 | s_in << 29
            while (__ballot(M != 0u)) {
                bool on = M != 0u;
                int i0 = 0, len = 0, k = -1;
                if (on) {
                    i0 = __ffs(M) - 1;
                    const unsigned rest = ~(M >> i0);
                    len = rest ? __ffs(rest) - 1 : 32 - i0;
                    M &= ~((len == 32 ? 0xffffffffu : ((1u << len) - 1u)) << i0);
                    const unsigned m = ~w & ((1u << i0) - 1u);
                    const int key = m ? 32 - __clz(m) : -1;
                    if (key != f_key) { f_key = key; comp_info(above ? -1 : run_root(y, wi, w, i0), S, kF); }
                    int B = FRAME, kB = -1;
                    if ((unsigned)zy < (unsigned)d.H && zy >= y_first) { B = run_root(zy, wi, rw, i0); kB = cand_of_root(lab, B); }
                    k = (B == S) ? kF : kB;
                    on = k >= 0;
                }
                const int x0 = wi * 32 + i0;
                emit(on, k, (unsigned)x0 | ((unsigned)y << 11) | code, len, -1, x0, x0 + len - 1, y);
            }
        }
    }
    // ---- phase 2: the other states, one pixel per lane: pixels with a gap that starts after some other neighbour, or isolated pixels
    const int G = (BS_DBG & 2) ? 0 : s_pre[wv][64];
    for (int base = 0; __builtin_amdgcn_readfirstlane(base) < G; base += 64) {
        const int it = base + lane_t;
        const bool have = it < G;
        // the word that holds item `it` (last lane L with pre[L] <= it) and the (it - pre[L])-th set bit of its generic mask
        int L = 0;
        if (have) {
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) if (L + step < 64 && s_pre[wv][L + step] <= it) L += step;
        }
        unsigned gm = have ? s_gen[wv][L] : 1u;
        int nth = have ? it - s_pre[wv][L] : 0;
        int i = 0;
        {   // n-th set bit by halving
            unsigned mm = gm;
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) {
                const int c = __popc(mm & ((1u << sh) - 1u));
                if (nth >= c) { nth -= c; i += sh; mm >>= sh; } else mm &= (1u << sh) - 1u;
            }
        }
        const int yL = y_lo + (tile / tiles_x) * 8 + (L >> 3), wiL = (tile % tiles_x) * 8 + (L & 7);
        const bool aboveL = yL < y_first;
        const unsigned wL = s_w[wv][L];
        const unsigned nb = ((s_pl[wv][0][L] >> i) & 1u) | (((s_pl[wv][1][L] >> i) & 1u) << 1) | (((s_pl[wv][2][L] >> i) & 1u) << 2) |
                            (((s_pl[wv][3][L] >> i) & 1u) << 3) | (((s_pl[wv][4][L] >> i) & 1u) << 4) | (((s_pl[wv][5][L] >> i) & 1u) << 5) |
                            (((s_pl[wv][6][L] >> i) & 1u) << 6) | (((s_pl[wv][7][L] >> i) & 1u) << 7);
        const int x = wiL * 32 + i;
        int S = -3, kF = -1;
        if (have) comp_info(aboveL ? -1 : run_root(yL, wiL, wL, i), S, kF);
        // gaps: for every foreground neighbour s_in whose counter-clockwise successor direction is background
        unsigned starts = nb & ~((nb >> 1) | (nb << 7)) & 0xffu;   // bit s set: neighbour s foreground, neighbour s+1 background
        const bool lone = nb == 0;
        if (lone) starts = 1u;                                  // isolated pixel: one state (a one-point contour)
        if (!have) starts = 0u;
        while (__ballot(starts != 0u)) {
            bool on = starts != 0u;
            int k = -1, s_in = 0, s_out = 0;
            if (on) {
                s_in = __ffs(starts) - 1;
                starts &= starts - 1;
                int g4;
                bool has4;
                const unsigned rot = ((nb | (nb << 8)) >> (s_in + 1)) & 0xffu;    // bit j = direction s_in + 1 + j
                if (lone) { s_out = 0; g4 = 4; has4 = true; }
                else {
                    const int Lg = __ffs(rot) - 1;               // background neighbours in the gap
                    s_out = (s_in + 1 + Lg) & 7;
                    g4 = ((s_in + 1) & 1) ? ((s_in + 2) & 7) : ((s_in + 1) & 7);      // first 4-direction at or after s_in + 1
                    has4 = Lg >= 2 || (Lg == 1 && ((s_in + 1) & 1) == 0);
                    if (Lg == 3 && (s_in & 3) == 0) has4 = false;    // a straight horizontal state: booked in phase 1
                }
                on = has4;                                      // (a lone diagonal background pixel: the walk passes by)
                if (on && aboveL) {                             // only a gap that holds the pixel below can be a strip border's
                    if (lone || (unsigned)((6 - (s_in + 1)) & 7) >= (unsigned)(__ffs(rot) - 1)) on = false;
                    g4 = 6;
                }
                if (on) {
                    int B = FRAME, kB = -1;
                    const int zx = x + dir_dx(g4), zy = yL + dir_dy(g4);
                    if ((unsigned)zx < (unsigned)d.W && (unsigned)zy < (unsigned)d.H && zy >= y_first) {
                        B = root_of_pixel(bimg, lab, wl, d, zx, zy);
                        kB = cand_of_root(lab, B);
                    }
                    k = (B == S) ? kF : kB;
                    on = k >= 0;
                }
            }
            const int np = (lone || s_out != (s_in ^ 4)) ? 1 : 0;    // the chain turns here: a CHAIN_APPROX_SIMPLE point
            emit(on, k, (unsigned)x | ((unsigned)yL << 11) | ((unsigned)s_out << 26) | ((unsigned)s_in << 29), 1, np, x, x, yL);
        }
    }
    wave_lds_sync();
    if (lane_t == 0) sg.hdr[(long)img * sg.cap / STAGE_TILE + tile] = tile_n;
    // the wave's table goes to the borders' records: one lane per slot -- and, round 6, to the tile's slice of sg.tab: the scatter pass
    // starts from these counts instead of adding them up again from the tile's records (a probing CAS and an LDS atomic per record)
    const int k = T.tag[lane_t];
    if (tile_n) {                                               // (uniform; a tile without records is not read back)
        int *tb = sg.tab + ((long)img * sg.cap / STAGE_TILE + tile) * (2 * WT_SLOTS);
        tb[lane_t] = k; tb[WT_SLOTS + lane_t] = T.n[lane_t] & 0xffff;
    }
    if (k >= 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&ac[k].nstates), (unsigned long long)(unsigned)(T.n[lane_t] & 0xffff) | ((unsigned long long)(unsigned)(T.n[lane_t] >> 16) << 32));
        atomicMin(&ac[k].xmin, T.x0[lane_t]); atomicMax(&ac[k].xmax, T.x1[lane_t]);
        atomicMin(&ac[k].ymin, T.y0[lane_t]); atomicMax(&ac[k].ymax, T.y1[lane_t]);
    }
}

// per-border stage limits the planner below shares with the stage kernels
constexpr int W_MW = 1024;                // border width a wave takes (wider: the full-size pass)
#ifndef PT_BAND_WORDS
#define PT_BAND_WORDS 128         // (round 6, re-tuned after the quad role's re-cut, tools/dbg/post_ab_both.sh: 256 / 16 loads in flight 0.2198 ms per call on the stress
#endif                            //  maps, 128 / 4: 0.2147, 128 / 8: 0.2157, 64: 0.238, 96: 0.226, 512: 0.224; 32 loads in flight 0.251; the scene maps do not care: 0.309-0.315)
constexpr int BAND_WORDS = PT_BAND_WORDS;           // mask words per score band (two LDS planes of this size per wave)
#ifndef PT_SCORE_UF
#define PT_SCORE_UF 4
#endif
constexpr int SCORE_UF = PT_SCORE_UF;     // 16-byte map loads a lane of the score role keeps in flight
// rows per score band of a border whose bounding box is bw wide
__device__ __forceinline__ int band_rows(int bw) { const int pw = (bw + 31) >> 5; return BAND_WORDS / pw > 0 ? BAND_WORDS / pw : 1; }
__device__ __forceinline__ bool border_is_tiny(int bw, int bh) { return (bw - 1) * (bw - 1) + (bh - 1) * (bh - 1) <= 8; }

// exclusive scan of the state counts over the candidates of one image (borders with <= 2 points are dropped by the
// reference, db_postprocess.cpp:255, and get no pool space); in the same sweep the plan of the score stage: a border that the hull
// role of border_wave_kernel will not filter out gets one item per band of band_rows() rows of its bounding box -- sc_off[k] = its
// first item, sc_n[img] = items of the image, sc_item[] = (border | band << 10) of every item
__global__ __launch_bounds__(1024) void pool_offsets_kernel(Acc *__restrict__ acc, const int *__restrict__ totals,
                                                            int *__restrict__ flags, DbpostDims d, int *__restrict__ sc_off,
                                                            int *__restrict__ sc_n, int *__restrict__ sc_item, long sc_cap) {
    const int img = blockIdx.x, k = threadIdx.x;
    __shared__ long sh[1024];
    __shared__ int shb[1024];
    const int num = min(totals[img], MAX_CAND);
    int n = 0, nb = 0;
    if (k < num) {
        const Acc a = acc[(long)img * MAX_CAND + k];
        n = a.npts > 2 ? a.nstates : 0;
        const int bw = a.xmax - a.xmin + 1, bh = a.ymax - a.ymin + 1;
        if (n && !border_is_tiny(bw, bh) && bw <= W_MW) { const int R = band_rows(bw); nb = (bh + R - 1) / R; }
    }
    sh[k] = n; shb[k] = nb;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long v = k >= off ? sh[k - off] : 0;
        const int vb = k >= off ? shb[k - off] : 0;
        __syncthreads();
        sh[k] += v; shb[k] += vb;
        __syncthreads();
    }
    const bool fits = sh[1023] <= d.pool_cap && shb[1023] <= sc_cap;
    if (k < num && n && fits) acc[(long)img * MAX_CAND + k].off = (int)(sh[k] - n);
    if (k < MAX_CAND) sc_off[(long)img * MAX_CAND + k] = shb[k] - nb;
    if (fits) for (int b = 0; b < nb; b++) sc_item[(long)img * sc_cap + shb[k] - nb + b] = k | (b << 10);      // item -> (border, band): no search in the score role
    if (k == 1023) { sc_n[img] = fits ? shb[1023] : 0; if (!fits) atomicOr(&flags[img], 4); }
}

// Moves the staged records of a tile to their borders' pool slots (acc[].off, set by pool_offsets_kernel; -1: the reference drops the
// border).  One wave per tile, one RECORD per lane.  Pass 1 adds up in the wave's table what each border needs; then ONE returning
// atomic per border of the tile reserves it -- one lane per table slot, so the reservations of a tile are in flight together (a returning
// atomic on a border's cursor takes ~200 ns behind the previous one on the same address; round 3 reserved per record round, ragged tiles
// per lane and record); pass 2 places every record behind its border's base with an LDS cursor.  No neighbour word, no label is read.
__global__ __launch_bounds__(256) void scatter_states_kernel(const int *__restrict__ strip_totals, Acc *__restrict__ acc,
                                                             unsigned *__restrict__ pool, const int *__restrict__ flags, DbpostDims d,
                                                             StageArgs2 sg) {
    const int img = blockIdx.y;
    if (flags[img] & 4) return;
    const int y_first = (d.strip_y && strip_totals[img] >= MAX_CAND) ? d.strip_y : 0;
    const int tiles_x = cdiv(d.WW, 8);
    const int wv = threadIdx.x >> 6, lane_t = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wv;
    const int y_lo = y_first ? y_first - 1 : 0;                   // the row above the strip is listed too (border_states_kernel)
    if (tile >= tiles_x * cdiv(d.H - y_lo, 8)) return;
    const int count = sg.hdr[(long)img * sg.cap / STAGE_TILE + tile];
    if (count == 0) return;                                     // (uniform over the wave)
    const uint2 *rec = sg.rec + (long)img * sg.cap + (long)tile * STAGE_TILE;
    Acc *ac = acc + (long)img * MAX_CAND;
    unsigned *pl = pool + (long)img * d.pool_cap;
    __shared__ int s_tag[4][WT_SLOTS], s_cnt[4][WT_SLOTS], s_base[4][WT_SLOTS];
    int *tag = s_tag[wv], *cnt = s_cnt[wv], *bs = s_base[wv];
    {   // (round 6: the table border_states_kernel left for this tile -- candidate and state count per slot -- instead of pass 1's recount;
        //  a border that found no room in it is not in it here either and takes the direct path below, as before)
        const int *tb = sg.tab + ((long)img * sg.cap / STAGE_TILE + tile) * (2 * WT_SLOTS);
        tag[lane_t] = tb[lane_t]; cnt[lane_t] = tb[WT_SLOTS + lane_t];
    }
    wave_lds_sync();
    {
        const int k = tag[lane_t];
        int base = -1;
        if (k >= 0) { const int off = ac[k].off; if (off >= 0) base = off + atomicAdd(&ac[k].cursor, cnt[lane_t]); }
        bs[lane_t] = base;
        cnt[lane_t] = 0;                                        // now the border's cursor inside this tile's share
    }
    wave_lds_sync();
    for (int i = lane_t; i < count; i += 64) {
        const uint2 e = rec[i];
        const int k = (int)(e.y & 0x3ffu), len = (int)(e.y >> 16), n = len ? len : 1;
        const int slot = (int)((e.y >> 10) & (WT_SLOTS - 1));     // (round 6: the slot travels in the record; a border without one finds another candidate there)
        int pos = -1;
        const bool found = tag[slot] == k;
        if (found) { if (bs[slot] >= 0) pos = bs[slot] + atomicAdd(&cnt[slot], n); }
        else { const int off = ac[k].off; if (off >= 0) pos = off + atomicAdd(&ac[k].cursor, n); }      // table was full (speckle): on its own
        if (pos >= 0) {
            if (len) { for (int j = 0; j < len; j++) pl[pos + j] = e.x + (unsigned)j; }
            else pl[pos] = e.x;
        }
    }
}

// ------------------------------------------------------------------------------------------ geometry (one lane)
struct F2 { float x, y; };
struct RRect { float cx, cy, w, h, angle; };

__device__ __forceinline__ int sgn(double v) { return (v > 0) - (v < 0); }

// One Sklansky chain over the x-sorted array (restates OpenCV convhull.cpp Sklansky_).
__device__ int sklansky(const F2 *a, int start, int end, int *stack, int nsign, int sign2) {
    const int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (a[start].x == a[end].x && a[start].y == a[end].y)) { stack[0] = start; return 1; }
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        const float cury = a[pcur].y, nexty = a[pnext].y;
        const float by = nexty - cury;
        if (sgn(by) != nsign) {
            const float ax = a[pcur].x - a[pprev].x;
            const float bx = a[pnext].x - a[pcur].x;
            const float ay = cury - a[pprev].y;
            const double convexity = (double)ay * bx - (double)ax * by;
            if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pcur = pnext; pnext += incr;
                stack[stacksize] = pnext; stacksize++;
            } else if (pprev == stack[0]) {
                pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
            } else {
                stack[stacksize - 2] = pnext;
                pcur = pprev; pprev = stack[stacksize - 4];
                stacksize--;
            }
        } else {
            pnext += incr;
            stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

// convexHull(clockwise=true) of n points sorted by (x, y); hull written to `hull` (capacity >= n); `stack` 2*(n+2) ints
__device__ int convex_hull_sorted(const F2 *a, int n, F2 *hull, int *stack) {
    int nout = 0, miny_ind = 0, maxy_ind = 0;
    for (int i = 1; i < n; i++) {
        const float y = a[i].y;
        if (a[miny_ind].y > y) miny_ind = i;
        if (a[maxy_ind].y < y) maxy_ind = i;
    }
    if (a[0].x == a[n - 1].x && a[0].y == a[n - 1].y) { hull[nout++] = a[0]; return nout; }
    int *tl_stack = stack;
    const int tl_count = sklansky(a, 0, maxy_ind, tl_stack, -1, 1);
    int *tr_stack = stack + tl_count;
    const int tr_count = sklansky(a, n - 1, maxy_ind, tr_stack, -1, -1);
    for (int i = 0; i < tl_count - 1; i++) hull[nout++] = a[tl_stack[i]];
    for (int i = tr_count - 1; i > 0; i--) hull[nout++] = a[tr_stack[i]];
    const int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    int *bl_stack = stack;
    int bl_count = sklansky(a, 0, miny_ind, bl_stack, 1, -1);
    int *br_stack = stack + bl_count;
    int br_count = sklansky(a, n - 1, miny_ind, br_stack, 1, 1);
    { int *ts = bl_stack; const int tc = bl_count; bl_stack = br_stack; bl_count = br_count; br_stack = ts; br_count = tc; }
    if (stop_idx >= 0) {
        const int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
        if (check_idx == stop_idx || (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
            bl_count = bl_count < 2 ? bl_count : 2;
            br_count = br_count < 2 ? br_count : 2;
        }
    }
    for (int i = 0; i < bl_count - 1; i++) hull[nout++] = a[bl_stack[i]];
    for (int i = br_count - 1; i > 0; i--) hull[nout++] = a[br_stack[i]];
    return nout;
}

// float32 rotating calipers (restates OpenCV rotcalipers.cpp, CALIPERS_MINAREARECT); scratch: 3*n floats + 4 ints (the four
// caliper positions are indexed dynamically: kept beside the tables, not in a private array that would live in scratch memory)
__device__ void rotating_calipers(const F2 *points, int n, float *scratch, float *out) {
    float minarea = 3.402823466e+38f;
    float *inv_len = scratch;
    F2 *vect = reinterpret_cast<F2 *>(scratch + n);
    int left = 0, bottom = 0, right = 0, top = 0;
    int *seq = reinterpret_cast<int *>(scratch + 3 * n);
    float orientation = 0, base_a, base_b = 0;
    F2 pt0 = points[0];
    float left_x = pt0.x, right_x = pt0.x, top_y = pt0.y, bottom_y = pt0.y;
    int buf_left = 0, buf_bottom = 0;
    float buf_a = 0, buf_w = 0, buf_b = 0, buf_h = 0;
    for (int i = 0; i < n; i++) {
        if (pt0.x < left_x) { left_x = pt0.x; left = i; }
        if (pt0.x > right_x) { right_x = pt0.x; right = i; }
        if (pt0.y > top_y) { top_y = pt0.y; top = i; }
        if (pt0.y < bottom_y) { bottom_y = pt0.y; bottom = i; }
        const F2 pt = points[(i + 1) < n ? (i + 1) : 0];
        const double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
        vect[i].x = (float)dx; vect[i].y = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0 = pt;
    }
    {
        double ax = vect[n - 1].x, ay = vect[n - 1].y;
        for (int i = 0; i < n; i++) {
            const double bx = vect[i].x, by = vect[i].y;
            const double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    for (int k = 0; k < n; k++) {
        float dp[4];
        dp[0] = +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y;
        dp[1] = -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y;
        dp[2] = -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y;
        dp[3] = +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y;
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; ++i) {
            const float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            const int pindex = seq[main_element];
            const float lead_x = vect[pindex].x * inv_len[pindex];
            const float lead_y = vect[pindex].y * inv_len[pindex];
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
        {
            float dx = points[seq[1]].x - points[seq[3]].x;
            float dy = points[seq[1]].y - points[seq[3]].y;
            const float width = dx * base_a + dy * base_b;
            dx = points[seq[2]].x - points[seq[0]].x;
            dy = points[seq[2]].y - points[seq[0]].y;
            const float height = -dx * base_b + dy * base_a;
            const float area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_left = seq[3]; buf_a = base_a; buf_w = width; buf_b = base_b; buf_h = height; buf_bottom = seq[0];
            }
        }
    }
    const float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
    const float C1 = A1 * points[buf_left].x + points[buf_left].y * B1;
    const float C2 = A2 * points[buf_bottom].x + points[buf_bottom].y * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * buf_w; out[3] = B1 * buf_w;
    out[4] = A2 * buf_h; out[5] = B2 * buf_h;
}

#define PT_PI 3.1415926535897932384626433832795

// minAreaRect on an already x-sorted point list
__device__ RRect min_area_rect_sorted(const F2 *sorted, int n, F2 *hull, int *stack, float *scratch) {
    RRect box; box.cx = box.cy = box.w = box.h = box.angle = 0.f;
    if (n <= 0) return box;
    const int hn = convex_hull_sorted(sorted, n, hull, stack);
    if (hn > 2) {
        float out[6];
        rotating_calipers(hull, hn, scratch, out);
        box.cx = out[0] + (out[2] + out[4]) * 0.5f;
        box.cy = out[1] + (out[3] + out[5]) * 0.5f;
        box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box.angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (hn == 2) {
        box.cx = (hull[0].x + hull[1].x) * 0.5f;
        box.cy = (hull[0].y + hull[1].y) * 0.5f;
        const double dx = hull[1].x - hull[0].x, dy = hull[1].y - hull[0].y;
        box.w = (float)sqrt(dx * dx + dy * dy);
        box.h = 0;
        box.angle = (float)atan2(dy, dx);
    } else if (hn == 1) {
        box.cx = hull[0].x; box.cy = hull[0].y;
    }
    box.angle = (float)(box.angle * 180 / PT_PI);
    return box;
}

// ---- wave-cooperative forms for the one-wave-per-border stages (all 64 lanes call them; blockDim.x == 64).  A lone lane pays
// 8+ cycles per dependent instruction and ~100 per dependent LDS read, so everything that is independent per hull edge -- the
// edge vectors and their inverse lengths (a double sqrt and a double division each), the four extreme vertices and the
// orientation -- runs one edge per lane, with the reference's arithmetic and the reference's tie rules (the FIRST index that
// attains an extreme; the first non-zero turn).  The caliper loop itself stays the reference's sequential code on lane 0.
// LDS hand-off between the lanes of ONE wave (the wave's LDS operations execute in order; the fences keep the compiler from
// moving accesses across, the barrier is a scheduling no-op for a single wave): usable inside any workgroup by one of its waves
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void wave_first_extreme(float v, int idx, bool want_max, int *out_idx) {
    // lane-private (value, index) candidates -> index of the extreme over the wave, lowest index among equals
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float v2 = __shfl_xor(v, o);
        const int i2 = __shfl_xor(idx, o);
        const bool better = i2 >= 0 && (idx < 0 || (want_max ? v2 > v : v2 < v) || (v2 == v && i2 < idx));
        if (better) { v = v2; idx = i2; }
    }
    *out_idx = idx;
}

__device__ void rotating_calipers_wave(const F2 *points, int n, float *scratch, float *out) {
    const int lane = threadIdx.x & 63;
    float *inv_len = scratch;
    F2 *vect = reinterpret_cast<F2 *>(scratch + n);
    int *seq = reinterpret_cast<int *>(scratch + 3 * n);
    // per-lane scan of its own vertices in increasing order (strict comparisons keep the first index), then a wave reduction
    float lx = 0, rx = 0, ty = 0, by = 0;
    int li = -1, ri = -1, ti = -1, bi = -1;
    for (int i = lane; i < n; i += 64) {
        const F2 pt0 = points[i];
        if (li < 0 || pt0.x < lx) { lx = pt0.x; li = i; }
        if (ri < 0 || pt0.x > rx) { rx = pt0.x; ri = i; }
        if (ti < 0 || pt0.y > ty) { ty = pt0.y; ti = i; }
        if (bi < 0 || pt0.y < by) { by = pt0.y; bi = i; }
        const F2 pt = points[(i + 1) < n ? (i + 1) : 0];
        const double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
        vect[i].x = (float)dx; vect[i].y = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    }
    int left, right, top, bottom;
    wave_first_extreme(lx, li, false, &left);
    wave_first_extreme(rx, ri, true, &right);
    wave_first_extreme(ty, ti, true, &top);
    wave_first_extreme(by, bi, false, &bottom);
    wave_sync();
    // orientation: sign of the first non-zero turn (edge i-1 -> edge i), scanning i = 0 .. n-1
    float orientation = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        double convexity = 0;
        if (i < n) {
            const F2 a = vect[i ? i - 1 : n - 1], b = vect[i];
            convexity = (double)a.x * (double)b.y - (double)a.y * (double)b.x;
        }
        const unsigned long long nz = __ballot(convexity != 0);
        if (nz) {
            const int src = __ffsll((long long)nz) - 1;
            orientation = __shfl(convexity > 0 ? 1.f : -1.f, src);
            break;
        }
    }
    if (n <= 64) {
        // The caliper walk is sequential, but it need not live in LDS: lane i keeps edge i and vertex i in registers, the WHOLE wave
        // runs the walk with wave-uniform state, and an indexed access is a v_readlane (a few cycles) instead of a dependent LDS round
        // trip (the lane-0 loop below spent ~1.5 k cycles per step on them).  Only the caliper that advanced is re-read.  Same float
        // operations in the same order as rotating_calipers().
        const int me = lane < n ? lane : 0;
        const F2 myv = vect[me], myp = points[me];
        const float myl = inv_len[me];
        auto rl = [](float v, int idx) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), idx)); };
        float minarea = 3.402823466e+38f;
        float base_a = orientation, base_b = 0;
        float buf_a = 0, buf_w = 0, buf_b = 0, buf_h = 0, bl_x = 0, bl_y = 0, bb_x = 0, bb_y = 0;
        int s0 = __builtin_amdgcn_readfirstlane(bottom), s1 = __builtin_amdgcn_readfirstlane(right), s2 = __builtin_amdgcn_readfirstlane(top),
            s3 = __builtin_amdgcn_readfirstlane(left);
        float v0x = rl(myv.x, s0), v0y = rl(myv.y, s0), l0 = rl(myl, s0), p0x = rl(myp.x, s0), p0y = rl(myp.y, s0);
        float v1x = rl(myv.x, s1), v1y = rl(myv.y, s1), l1 = rl(myl, s1), p1x = rl(myp.x, s1), p1y = rl(myp.y, s1);
        float v2x = rl(myv.x, s2), v2y = rl(myv.y, s2), l2 = rl(myl, s2), p2x = rl(myp.x, s2), p2y = rl(myp.y, s2);
        float v3x = rl(myv.x, s3), v3y = rl(myv.y, s3), l3 = rl(myl, s3), p3x = rl(myp.x, s3), p3y = rl(myp.y, s3);
        for (int k = 0; k < n; k++) {
            float dp[4];
            dp[0] = +base_a * v0x + base_b * v0y;
            dp[1] = -base_b * v1x + base_a * v1y;
            dp[2] = -base_a * v2x - base_b * v2y;
            dp[3] = +base_b * v3x - base_a * v3y;
            float maxcos = dp[0] * l0;
            int main_element = 0;
            { const float c = dp[1] * l1; if (c > maxcos) { main_element = 1; maxcos = c; } }
            { const float c = dp[2] * l2; if (c > maxcos) { main_element = 2; maxcos = c; } }
            { const float c = dp[3] * l3; if (c > maxcos) { main_element = 3; maxcos = c; } }
            main_element = __builtin_amdgcn_readfirstlane(main_element);
            if (main_element == 0) {
                const float lead_x = v0x * l0, lead_y = v0y * l0;
                base_a = lead_x; base_b = lead_y;
                s0 = s0 + 1 == n ? 0 : s0 + 1;
                v0x = rl(myv.x, s0); v0y = rl(myv.y, s0); l0 = rl(myl, s0); p0x = rl(myp.x, s0); p0y = rl(myp.y, s0);
            } else if (main_element == 1) {
                const float lead_x = v1x * l1, lead_y = v1y * l1;
                base_a = lead_y; base_b = -lead_x;
                s1 = s1 + 1 == n ? 0 : s1 + 1;
                v1x = rl(myv.x, s1); v1y = rl(myv.y, s1); l1 = rl(myl, s1); p1x = rl(myp.x, s1); p1y = rl(myp.y, s1);
            } else if (main_element == 2) {
                const float lead_x = v2x * l2, lead_y = v2y * l2;
                base_a = -lead_x; base_b = -lead_y;
                s2 = s2 + 1 == n ? 0 : s2 + 1;
                v2x = rl(myv.x, s2); v2y = rl(myv.y, s2); l2 = rl(myl, s2); p2x = rl(myp.x, s2); p2y = rl(myp.y, s2);
            } else {
                const float lead_x = v3x * l3, lead_y = v3y * l3;
                base_a = -lead_y; base_b = lead_x;
                s3 = s3 + 1 == n ? 0 : s3 + 1;
                v3x = rl(myv.x, s3); v3y = rl(myv.y, s3); l3 = rl(myl, s3); p3x = rl(myp.x, s3); p3y = rl(myp.y, s3);
            }
            float dx = p1x - p3x;
            float dy = p1y - p3y;
            const float width = dx * base_a + dy * base_b;
            dx = p2x - p0x;
            dy = p2y - p0y;
            const float height = -dx * base_b + dy * base_a;
            const float area = width * height;
            if (area <= minarea) {
                minarea = area;
                bl_x = p3x; bl_y = p3y; buf_a = base_a; buf_w = width; buf_b = base_b; buf_h = height; bb_x = p0x; bb_y = p0y;
            }
        }
        const float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
        const float C1 = A1 * bl_x + bl_y * B1;
        const float C2 = A2 * bb_x + bb_y * B2;
        const float idet = 1.f / (A1 * B2 - A2 * B1);
        out[0] = (C1 * B2 - C2 * B1) * idet;
        out[1] = (A1 * C2 - A2 * C1) * idet;
        out[2] = A1 * buf_w; out[3] = B1 * buf_w;
        out[4] = A2 * buf_h; out[5] = B2 * buf_h;
        return;
    }
    if (lane != 0) return;
    float minarea = 3.402823466e+38f;
    float base_a = orientation, base_b = 0;
    int buf_left = 0, buf_bottom = 0;
    float buf_a = 0, buf_w = 0, buf_b = 0, buf_h = 0;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
    for (int k = 0; k < n; k++) {
        const int s0 = seq[0], s1 = seq[1], s2 = seq[2], s3 = seq[3];
        const F2 v0 = vect[s0], v1 = vect[s1], v2 = vect[s2], v3 = vect[s3];
        const float l0 = inv_len[s0], l1 = inv_len[s1], l2 = inv_len[s2], l3 = inv_len[s3];
        float dp[4];
        dp[0] = +base_a * v0.x + base_b * v0.y;
        dp[1] = -base_b * v1.x + base_a * v1.y;
        dp[2] = -base_a * v2.x - base_b * v2.y;
        dp[3] = +base_b * v3.x - base_a * v3.y;
        float maxcos = dp[0] * l0;
        int main_element = 0;
        { const float c = dp[1] * l1; if (c > maxcos) { main_element = 1; maxcos = c; } }
        { const float c = dp[2] * l2; if (c > maxcos) { main_element = 2; maxcos = c; } }
        { const float c = dp[3] * l3; if (c > maxcos) { main_element = 3; maxcos = c; } }
        {
            const F2 vm = main_element == 0 ? v0 : main_element == 1 ? v1 : main_element == 2 ? v2 : v3;
            const float lm = main_element == 0 ? l0 : main_element == 1 ? l1 : main_element == 2 ? l2 : l3;
            const float lead_x = vm.x * lm;
            const float lead_y = vm.y * lm;
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        {
            int sm = seq[main_element] + 1;
            sm = (sm == n) ? 0 : sm;
            seq[main_element] = sm;
        }
        {
            const F2 p0 = points[seq[0]], p1 = points[seq[1]], p2 = points[seq[2]], p3 = points[seq[3]];
            float dx = p1.x - p3.x;
            float dy = p1.y - p3.y;
            const float width = dx * base_a + dy * base_b;
            dx = p2.x - p0.x;
            dy = p2.y - p0.y;
            const float height = -dx * base_b + dy * base_a;
            const float area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_left = seq[3]; buf_a = base_a; buf_w = width; buf_b = base_b; buf_h = height; buf_bottom = seq[0];
            }
        }
    }
    const float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
    const float C1 = A1 * points[buf_left].x + points[buf_left].y * B1;
    const float C2 = A2 * points[buf_bottom].x + points[buf_bottom].y * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * buf_w; out[3] = B1 * buf_w;
    out[4] = A2 * buf_h; out[5] = B2 * buf_h;
}

// convex_hull_sorted() for n <= 64 run by the whole wave with wave-uniform state: lane i keeps point i in registers, an indexed point
// is a v_readlane, the chain's three running points are carried in registers (one new point per step), the index stack stays in
// LDS (written by lane 0, read back only when a point is popped), and the extreme scans / output copies go one element per lane.
// Same comparisons and float operations as sklansky() / convex_hull_sorted(); returns the hull size on every lane.
__device__ int sklansky_wave(float myx, float myy, int start, int end, int *stack, int nsign, int sign2) {
    const bool l0 = (threadIdx.x & 63) == 0;
    auto X = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myx), i)); };
    auto Y = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myy), i)); };
    start = __builtin_amdgcn_readfirstlane(start); end = __builtin_amdgcn_readfirstlane(end);
    const int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (X(start) == X(end) && Y(start) == Y(end))) { if (l0) stack[0] = start; return 1; }
    const int first = pprev;                                    // stack[0] never changes
    if (l0) { stack[0] = pprev; stack[1] = pcur; stack[2] = pnext; }
    end += incr;
    float ppx = X(pprev), ppy = Y(pprev), pcx = X(pcur), pcy = Y(pcur);
    while (pnext != end) {
        const float pnx = X(pnext), pny = Y(pnext);
        const float by = pny - pcy;
        if (sgn(by) != nsign) {
            const float ax = pcx - ppx;
            const float bx = pnx - pcx;
            const float ay = pcy - ppy;
            const double convexity = (double)ay * bx - (double)ax * by;
            if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; ppx = pcx; ppy = pcy;
                pcur = pnext; pcx = pnx; pcy = pny;
                pnext += incr;
                if (l0) stack[stacksize] = pnext;
                stacksize++;
            } else if (pprev == first) {
                pcur = pnext; pcx = pnx; pcy = pny;
                pnext += incr;
                if (l0) { stack[1] = pcur; stack[2] = pnext; }
            } else {
                if (l0) stack[stacksize - 2] = pnext;
                pcur = pprev; pcx = ppx; pcy = ppy;
                pprev = __builtin_amdgcn_readfirstlane(stack[stacksize - 4]);
                ppx = X(pprev); ppy = Y(pprev);
                stacksize--;
            }
        } else {
            pnext += incr;
            if (l0) stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

__device__ int convex_hull_sorted_wave(const F2 *a, int n, F2 *hull, int *stack) {
    const int lane = threadIdx.x & 63;
    const F2 me = a[lane < n ? lane : 0];
    int miny_ind, maxy_ind;
    wave_first_extreme(me.y, lane < n ? lane : -1, false, &miny_ind);      // first index of the minimum / maximum, as the strict scans give
    wave_first_extreme(me.y, lane < n ? lane : -1, true, &maxy_ind);
    miny_ind = __builtin_amdgcn_readfirstlane(miny_ind); maxy_ind = __builtin_amdgcn_readfirstlane(maxy_ind);
    int nout = 0;
    if (a[0].x == a[n - 1].x && a[0].y == a[n - 1].y) { if (lane == 0) hull[0] = a[0]; wave_sync(); return 1; }
    int *tl_stack = stack;
    const int tl_count = sklansky_wave(me.x, me.y, 0, maxy_ind, tl_stack, -1, 1);
    int *tr_stack = stack + tl_count;
    const int tr_count = sklansky_wave(me.x, me.y, n - 1, maxy_ind, tr_stack, -1, -1);
    wave_sync();
    if (lane < tl_count - 1) hull[nout + lane] = a[tl_stack[lane]];
    nout += tl_count > 1 ? tl_count - 1 : 0;
    if (lane < tr_count - 1) hull[nout + lane] = a[tr_stack[tr_count - 1 - lane]];
    nout += tr_count > 1 ? tr_count - 1 : 0;
    const int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    wave_sync();                                                // the stacks are read: the lower chains reuse their memory
    int *bl_stack = stack;
    int bl_count = sklansky_wave(me.x, me.y, 0, miny_ind, bl_stack, 1, -1);
    int *br_stack = stack + bl_count;
    int br_count = sklansky_wave(me.x, me.y, n - 1, miny_ind, br_stack, 1, 1);
    wave_sync();
    { int *ts = bl_stack; const int tc = bl_count; bl_stack = br_stack; bl_count = br_count; br_stack = ts; br_count = tc; }
    if (stop_idx >= 0) {
        const int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
        if (check_idx == stop_idx || (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
            bl_count = bl_count < 2 ? bl_count : 2;
            br_count = br_count < 2 ? br_count : 2;
        }
    }
    if (lane < bl_count - 1) hull[nout + lane] = a[bl_stack[lane]];
    nout += bl_count > 1 ? bl_count - 1 : 0;
    if (lane < br_count - 1) hull[nout + lane] = a[br_stack[br_count - 1 - lane]];
    nout += br_count > 1 ? br_count - 1 : 0;
    wave_sync();
    return nout;
}

// minAreaRect on an already x-sorted point list, wave-cooperative; the result is valid on lane 0.  sh_hn: one LDS int.
__device__ RRect min_area_rect_wave(const F2 *sorted, int n, F2 *hull, int *stack, float *scratch, int *sh_hn) {
    RRect box; box.cx = box.cy = box.w = box.h = box.angle = 0.f;
    if (n <= 0) return box;                                     // uniform
    int hn;
    if (n <= 64) hn = __builtin_amdgcn_readfirstlane(convex_hull_sorted_wave(sorted, n, hull, stack));
    else {
        if ((threadIdx.x & 63) == 0) *sh_hn = convex_hull_sorted(sorted, n, hull, stack);
        wave_sync();
        hn = *sh_hn;
    }
    if (hn > 2) {
        float out[6] = {0, 0, 0, 0, 0, 0};
        rotating_calipers_wave(hull, hn, scratch, out);
        box.cx = out[0] + (out[2] + out[4]) * 0.5f;
        box.cy = out[1] + (out[3] + out[5]) * 0.5f;
        box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box.angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (hn == 2) {
        box.cx = (hull[0].x + hull[1].x) * 0.5f;
        box.cy = (hull[0].y + hull[1].y) * 0.5f;
        const double dx = hull[1].x - hull[0].x, dy = hull[1].y - hull[0].y;
        box.w = (float)sqrt(dx * dx + dy * dy);
        box.h = 0;
        box.angle = (float)atan2(dy, dx);
    } else if (hn == 1) {
        box.cx = hull[0].x; box.cy = hull[0].y;
    }
    box.angle = (float)(box.angle * 180 / PT_PI);
    return box;
}

__device__ void box_points(const RRect &r, F2 pt[4]) {
    const double ang = r.angle * PT_PI / 180.;
    const float b = (float)cos(ang) * 0.5f;
    const float a = (float)sin(ang) * 0.5f;
    pt[0].x = r.cx - a * r.h - b * r.w;
    pt[0].y = r.cy + b * r.h - a * r.w;
    pt[1].x = r.cx + a * r.h - b * r.w;
    pt[1].y = r.cy - b * r.h - a * r.w;
    pt[2].x = 2 * r.cx - pt[0].x;
    pt[2].y = 2 * r.cy - pt[0].y;
    pt[3].x = 2 * r.cx - pt[1].x;
    pt[3].y = 2 * r.cy - pt[1].y;
}

// db_postprocess.cpp:159-192 (std::sort of 4 elements = insertion sort => stable on equal x).  The sorted order is written as
// ranks (point i goes to the number of points that sort before it) and selections: no dynamically indexed private array.
__device__ void get_mini_boxes(const RRect &box, float out[4][2], float *ssid) {
    F2 p[4];
    *ssid = box.w > box.h ? box.w : box.h;
    box_points(box, p);
    int r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j != i) r[i] += (p[j].x < p[i].x || (p[j].x == p[i].x && j < i)) ? 1 : 0;
    }
    F2 q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        q[k] = p[0];
#pragma unroll
        for (int i = 1; i < 4; i++) if (r[i] == k) q[k] = p[i];
    }
    F2 i1, i2, i3, i4;
    if (q[3].y <= q[2].y) { i2 = q[3]; i3 = q[2]; } else { i2 = q[2]; i3 = q[3]; }
    if (q[1].y <= q[0].y) { i1 = q[1]; i4 = q[0]; } else { i1 = q[0]; i4 = q[1]; }
    out[0][0] = i1.x; out[0][1] = i1.y; out[1][0] = i2.x; out[1][1] = i2.y;
    out[2][0] = i3.x; out[2][1] = i3.y; out[3][0] = i4.x; out[3][1] = i4.y;
}

// Clipper 6.4.2 round offset of one closed path (restates clipper.cpp:3837-3879, 3889-3913, 3987-4020, 4160-4244);
// the union clean-up that follows in ClipperOffset::Execute does not change the hull of the result for
// delta >= 0.75 (checked against the vendored Clipper in tests/test_oracle_clipper.py).
struct CPt { long long X, Y; };
__device__ __forceinline__ long long cl_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

// ws: 16 eight-byte words of workspace (LDS): the de-duplicated path and the edge normals are indexed dynamically
__device__ int clipper_offset_round(const CPt *path4, double delta, F2 *out, int cap, long long *ws) {
    const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc = 0.25, arc_tol = 0.25;
    CPt *src = reinterpret_cast<CPt *>(ws);
    double *nx = reinterpret_cast<double *>(ws + 8), *ny = nx + 4;
    int highI = 3, j = 0, nout = 0;
    while (highI > 0 && path4[0].X == path4[highI].X && path4[0].Y == path4[highI].Y) highI--;
    src[0] = path4[0];
    for (int i = 1; i <= highI; i++)
        if (src[j].X != path4[i].X || src[j].Y != path4[i].Y) { j++; src[j] = path4[i]; }
    if (j < 2) return 0;
    const int len = j + 1;
    double a = 0;
    for (int i = 0, k = len - 1; i < len; ++i) { a += ((double)src[k].X + src[i].X) * ((double)src[k].Y - src[i].Y); k = i; }
    if (!(-a * 0.5 >= 0))
        for (int i = 0; i < len / 2; i++) { const CPt t = src[i]; src[i] = src[len - 1 - i]; src[len - 1 - i] = t; }
#define PT_PUSH(px, py) do { if (nout < cap) { out[nout].x = (float)(px); out[nout].y = (float)(py); } nout++; } while (0)
    if (delta > -1.0e-20 && delta < 1.0e-20) {
        for (int i = 0; i < len; i++) PT_PUSH(src[i].X, src[i].Y);
        return nout;
    }
    double yv;
    if (arc_tol > fabs(delta) * def_arc) yv = fabs(delta) * def_arc; else yv = arc_tol;
    double steps = pi / acos(1 - yv / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    double m_sin = sin(two_pi / steps);
    const double m_cos = cos(two_pi / steps);
    const double steps_per_rad = steps / two_pi;
    if (delta < 0.0) m_sin = -m_sin;
    for (int i = 0; i < len; ++i) {
        const CPt p1 = src[i], p2 = src[(i + 1) % len];
        if (p2.X == p1.X && p2.Y == p1.Y) { nx[i] = 0; ny[i] = 0; continue; }
        double Dx = (double)(p2.X - p1.X), dy = (double)(p2.Y - p1.Y);
        const double f = 1 * 1.0 / sqrt(Dx * Dx + dy * dy);
        Dx *= f; dy *= f;
        nx[i] = dy; ny[i] = -Dx;
    }
    int k = len - 1;
    for (j = 0; j < len; ++j) {
        double sinA = nx[k] * ny[j] - nx[j] * ny[k];
        bool done = false;
        if (fabs(sinA * delta) < 1.0) {
            const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
            if (cosA > 0) {
                PT_PUSH(cl_round(src[j].X + nx[k] * delta), cl_round(src[j].Y + ny[k] * delta));
                done = true;                          // the original returns here, before k = j
            }
        } else if (sinA > 1.0) sinA = 1.0;
        else if (sinA < -1.0) sinA = -1.0;
        if (done) continue;
        if (sinA * delta < 0) {
            PT_PUSH(cl_round(src[j].X + nx[k] * delta), cl_round(src[j].Y + ny[k] * delta));
            PT_PUSH(src[j].X, src[j].Y);
            PT_PUSH(cl_round(src[j].X + nx[j] * delta), cl_round(src[j].Y + ny[j] * delta));
        } else {
            const double ang = atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
            const long long r = cl_round(steps_per_rad * fabs(ang));
            const int nsteps = (int)r > 1 ? (int)r : 1;
            double X = nx[k], Y = ny[k], X2;
            for (int s = 0; s < nsteps; ++s) {
                PT_PUSH(cl_round(src[j].X + X * delta), cl_round(src[j].Y + Y * delta));
                X2 = X;
                X = X * m_cos - m_sin * Y;
                Y = X2 * m_sin + Y * m_cos;
            }
            PT_PUSH(cl_round(src[j].X + nx[j] * delta), cl_round(src[j].Y + ny[j] * delta));
        }
        k = j;
    }
#undef PT_PUSH
    return nout;
}

// ---- the union ClipperOffset::Execute runs over its offset polygon (clipper.cpp:3916-3943), as far as cv::minAreaRect can see it
// (hull of the vertices, emptiness).  For the offset of a convex quadrilateral the Vatti sweep can do three things to that hull:
//   1. AddPath never makes edges of duplicate / collinear vertices (clipper.cpp:1097-1130) -- matters because such a vertex raises no
//      scan-line event;
//   2. the polygon is two y-monotone bounds from the bottom (largest Y) to the top.  When an intermediate vertex a of one bound is
//      promoted (clipper.cpp:3102-3136) while the other bound's edge, strictly spanning that scan line, has its ROUNDED position
//      (TopX, clipper.cpp:623-627) on a, and the rest of that edge seen from there has the slope of the edge leaving a, Clipper
//      records a join; JoinCommonEdges pinches the polygon at a: the part above has no area and is dropped, the part below keeps
//      its vertices (fewer than three distinct ones: no solution);
//   3. a polygon without area has no solution.
// The sub-pixel slivers this is about are the thin 1-pixel components of noisy maps.  Pinned through the oracle against the
// reference's own Clipper (2 000 000 random boxes, tests/test_oracle_clipper.py) and here against the oracle WITH that Clipper.
// One lane.  P[n] in / out (a subset survives), q: n points of scratch, chain: 2 n ints.  Returns the count kept (0: no solution).
// The union can only touch the hull when two bounds come within half a pixel of each other at a vertex scan line; the offset polygon
// of distance d is at least 2 sqrt(d - 1/4) - 1 px wide there (0.73 px at d = 1).  Over 1.8 million random boxes the largest distance
// at which the reference's Clipper changed the hull was 0.556 px; the step is skipped from 1 px on (text boxes: d >= 1.27 px).
constexpr float UNION_MAX_DISTANCE = 1.0f;
__device__ int cu_dedupe(F2 *q, int m) {
    int k = 0;
    for (int i = 0; i < m; i++)
        if (k == 0 || q[k - 1].x != q[i].x || q[k - 1].y != q[i].y) q[k++] = q[i];
    while (k > 1 && q[0].x == q[k - 1].x && q[0].y == q[k - 1].y) k--;
    return k;
}
__device__ int clipper_union_cut(F2 *P, int n, F2 *q, int *chain) {
    for (int i = 0; i < n; i++) q[i] = P[i];
    int m = cu_dedupe(q, n);
    for (;;) {                                                  // collinear vertices, one at a time
        if (m < 3) return 0;
        int hit = -1;
        for (int i = 0; i < m && hit < 0; i++) {
            const F2 a = q[(i + m - 1) % m], b = q[i], c = q[(i + 1) % m];
            if ((int)(b.y - a.y) * (int)(c.x - b.x) == (int)(b.x - a.x) * (int)(c.y - b.y)) hit = i;
        }
        if (hit < 0) break;
        for (int j = hit; j < m - 1; j++) q[j] = q[j + 1];
        m = cu_dedupe(q, m - 1);
    }
    long long area2 = 0;
    for (int i = 0; i < m; i++) { const int j = (i + 1) % m; area2 += (long long)((int)q[i].x * (int)q[j].y) - (long long)((int)q[j].x * (int)q[i].y); }
    if (area2 == 0) return 0;
    float ymax = q[0].y, ymin = q[0].y;
    for (int i = 1; i < m; i++) { ymax = fmaxf(ymax, q[i].y); ymin = fminf(ymin, q[i].y); }
    int nb = 0, nt = 0, b0 = -1, b1 = -1, t0 = -1, t1 = -1;
    for (int i = 0; i < m; i++) {
        if (q[i].y == ymax) { nb++; if (b0 < 0) b0 = i; else b1 = i; }
        if (q[i].y == ymin) { nt++; if (t0 < 0) t0 = i; else t1 = i; }
    }
    // the bottom (top) is one vertex or two neighbours; b0 -> b1 and t0 -> t1 run with increasing index
    if (nb == 1) b1 = b0; else if (nb == 2) { if ((b0 + 1) % m != b1) { if ((b1 + 1) % m != b0) return n; const int t = b0; b0 = b1; b1 = t; } } else return n;
    if (nt == 1) t1 = t0; else if (nt == 2) { if ((t0 + 1) % m != t1) { if ((t1 + 1) % m != t0) return n; const int t = t0; t0 = t1; t1 = t; } } else return n;
    int *cA = chain, *cB = chain + m;
    int la = 0, lb = 0;
    for (int i = b1;; i = (i + 1) % m) { cA[la++] = i; if (i == t0) break; if (la >= m) return n; }
    for (int i = b0;; i = (i + m - 1) % m) { cB[lb++] = i; if (i == t1) break; if (lb >= m) return n; }
    for (int i = 0; i + 1 < la; i++) if (!(q[cA[i + 1]].y < q[cA[i]].y)) return n;
    for (int i = 0; i + 1 < lb; i++) if (!(q[cB[i + 1]].y < q[cB[i]].y)) return n;
    if (la + lb - (b0 == b1) - (t0 == t1) != m) return n;
    int ia = 1, ib = 1;
    while (ia < la - 1 || ib < lb - 1) {                        // intermediate vertices of both bounds, from the bottom up
        bool takeA;
        if (ia >= la - 1) takeA = false; else if (ib >= lb - 1) takeA = true; else takeA = q[cA[ia]].y >= q[cB[ib]].y;
        const int *C = takeA ? cA : cB, *O = takeA ? cB : cA;
        const int lo = takeA ? lb : la, k = takeA ? ia++ : ib++;
        const int ax = (int)q[C[k]].x, y = (int)q[C[k]].y, ex = (int)q[C[k + 1]].x, ey = (int)q[C[k + 1]].y;
        for (int e = 0; e + 1 < lo; e++) {
            const int bx = (int)q[O[e]].x, by = (int)q[O[e]].y, tx = (int)q[O[e + 1]].x, ty = (int)q[O[e + 1]].y;
            if (!(by > y && y > ty)) continue;
            const double dx = (double)(tx - bx) / (double)(ty - by);
            const int x = bx + (int)cl_round(dx * (double)(y - by));
            if (x == ax && (ey - y) * (tx - x) == (ex - x) * (ty - y)) {
                int kept = 0, distinct = 0;
                for (int u = 0; u < m; u++)
                    if ((int)q[u].y >= y) P[kept++] = q[u];
                for (int u = 0; u < kept; u++) {
                    int v = 0;
                    for (; v < u; v++) if (P[v].x == P[u].x && P[v].y == P[u].y) break;
                    distinct += v == u;
                }
                return distinct >= 3 ? kept : 0;
            }
            break;
        }
    }
    return n;
}

// ------------------------------------------------------------------------------------------ group-lane geometry
// The sequential float32 / double geometry of a border (Sklansky's chains, the caliper walk, Clipper's round offset) run by a WHOLE
// wave for one border keeps 63 lanes idle: at ~4 600 borders per batch of 32 maps that is what the stage costs (unclip: 43 us for
// one border alone on its SIMD, 89 us with five waves per SIMD).  Here a wave owns FOUR borders, SIXTEEN LANES each (round 6; rounds
// 4-5 gave a border four lanes and left lanes 16..63 of the wave idle):
//   * everything that is independent per point or per edge -- reading the candidates, the edge table of the caliper walk (a double
//     sqrt and a double division per edge), the extreme scans, the rank sort of the offset polygon, the hull's output copy -- runs one
//     element per lane over the sixteen lanes;
//   * the four Sklansky chains of a hull run one per lane on lanes 0..3 of the group, the four corners of the Clipper offset one per
//     lane (the k-chain of OffsetPoint's early return is resolved first, from the normals alone);
//   * the caliper walk -- whose steps depend on each other -- runs ONE CALIPER PER LANE (round 6; before, every lane carried all four):
//     a lane computes its own caliper's cosine, the quad agrees on the winner through DPP quad permutes (a cross-lane operand of the
//     instruction itself: no LDS, no extra latency), only the winner advances.  Lanes 4..15 of the group repeat lanes 0..3 (same
//     instructions, same values): nothing in the walk diverges.
// Same float operations in the same order as the sequential forms above, which stay as the reference for the full-size pass.
constexpr int Q_PTS = 64;                 // hull candidates / offset points a group handles; more -> the full-size pass
constexpr int GL = 16;                    // lanes per border
struct __attribute__((aligned(16))) QuadArena {                        // LDS of one border
    F2 pts[Q_PTS];                        // the point list sorted by (x, y)
    F2 hull[Q_PTS];                       // the unsorted offset polygon, later the convex hull
    union {                               // never alive together: the chains end before the caliper table (or the sort keys) is built
        int stack[4][Q_PTS + 2];          // the four Sklansky chains
        float4 ev[Q_PTS];                 // caliper table: edge vector, inverse length; the rank sort's 64-bit keys
    };
    long long cl_ws[24];                  // Clipper: de-duplicated path (4 x 2), normals (2 x 4), input path (4 x 2)
};

// quad permutes as DPP operands (dpp_ctrl = p0 | p1 << 2 | p2 << 4 | p3 << 6): lane q of every aligned quad reads lane p_q of it
template <int CTRL> __device__ __forceinline__ int qperm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float qperm(float v) { return __builtin_bit_cast(float, qperm_i<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int QP_SWAP1 = 0xB1, QP_SWAP2 = 0x4E;           // [1,0,3,2], [2,3,0,1]: the two butterfly steps of a reduction over a quad
constexpr int QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA, QP_B3 = 0xFF;      // broadcasts of lane 0 / 1 / 2 / 3

// (value, index) candidates of the sixteen lanes -> index of the extreme, lowest index among equals, on all sixteen
__device__ __forceinline__ void group_first_extreme(float v, int idx, bool want_max, int *out_idx) {
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) {
        const float v2 = __shfl_xor(v, o, GL);
        const int i2 = __shfl_xor(idx, o, GL);
        const bool better = i2 >= 0 && (idx < 0 || (want_max ? v2 > v : v2 < v) || (v2 == v && i2 < idx));
        if (better) { v = v2; idx = i2; }
    }
    *out_idx = idx;
}

// sklansky() for one lane of a group: the chain's three running points live in registers (one LDS read per step for the new point,
// two when a point is popped) instead of five dependent reads per step; same comparisons and float operations
__device__ __forceinline__ int sklansky_reg(const F2 *a, int start, int end, int *stack, int nsign, int sign2) {
    const int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (a[start].x == a[end].x && a[start].y == a[end].y)) { stack[0] = start; return 1; }
    const int first = pprev;                                    // stack[0] never changes
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    F2 pp = a[pprev], pc = a[pcur];
    F2 pn = a[pnext != end ? pnext : pcur];
    while (pnext != end) {
        const int nn = pnext + incr;
        const F2 ahead = a[nn != end ? nn : pnext];              // the point after pnext: the one the next step needs in three of four cases
        const float by = pn.y - pc.y;
        if (sgn(by) != nsign) {
            const float ax = pc.x - pp.x;
            const float bx = pn.x - pc.x;
            const float ay = pc.y - pp.y;
            const double convexity = (double)ay * bx - (double)ax * by;
            if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pp = pc;
                pcur = pnext; pc = pn;
                pnext = nn; pn = ahead;
                stack[stacksize] = pnext;
                stacksize++;
            } else if (pprev == first) {
                pcur = pnext; pc = pn;
                pnext = nn; pn = ahead;
                stack[1] = pcur; stack[2] = pnext;
            } else {
                stack[stacksize - 2] = pnext;
                pcur = pprev; pc = pp;
                pprev = stack[stacksize - 4];
                pp = a[pprev];
                stacksize--;
            }
        } else {
            pnext = nn; pn = ahead;
            stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

// convex_hull_sorted() with one Sklansky chain per lane (lanes 0..3 of the group); returns the hull size on the sixteen lanes
__device__ __forceinline__ int convex_hull_sorted_q4(const F2 *a, int n, F2 *hull, int (*stack)[Q_PTS + 2]) {
    const int g = threadIdx.x & (GL - 1);
    float ylo = 0, yhi = 0;
    int ilo = -1, ihi = -1;
    for (int i = g; i < n; i += GL) {                           // strict comparisons keep the first index of my subsequence
        const float y = a[i].y;
        if (ilo < 0 || y < ylo) { ylo = y; ilo = i; }
        if (ihi < 0 || y > yhi) { yhi = y; ihi = i; }
    }
    int miny_ind, maxy_ind;
    group_first_extreme(ylo, ilo, false, &miny_ind);
    group_first_extreme(yhi, ihi, true, &maxy_ind);
    if (a[0].x == a[n - 1].x && a[0].y == a[n - 1].y) { if (g == 0) hull[0] = a[0]; wave_sync(); return 1; }
    // lane 0: top-left chain, 1: top-right, 2: bottom-left, 3: bottom-right (convex_hull_sorted's four calls)
    int cnt = 0;
    if (g < 4) cnt = sklansky_reg(a, (g & 1) ? n - 1 : 0, g < 2 ? maxy_ind : miny_ind, stack[g], g < 2 ? -1 : 1, (g == 0 || g == 3) ? 1 : -1);
    wave_sync();
    const int tl_count = __shfl(cnt, 0, GL), tr_count = __shfl(cnt, 1, GL);
    int bl_count = __shfl(cnt, 3, GL), br_count = __shfl(cnt, 2, GL);     // the reference swaps the two bottom chains before it uses them
    const int *tl_stack = stack[0], *tr_stack = stack[1], *bl_stack = stack[3], *br_stack = stack[2];
    const int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    if (stop_idx >= 0) {
        const int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
        if (check_idx == stop_idx || (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
            bl_count = bl_count < 2 ? bl_count : 2;
            br_count = br_count < 2 ? br_count : 2;
        }
    }
    const int n0 = tl_count > 1 ? tl_count - 1 : 0, n1 = tr_count > 1 ? tr_count - 1 : 0;
    const int n2 = bl_count > 1 ? bl_count - 1 : 0, n3 = br_count > 1 ? br_count - 1 : 0;
    const int total = n0 + n1 + n2 + n3;
    // output element t of the hull, one per lane: the chain it belongs to and its place there (top-left forwards, top-right backwards,
    // bottom-left forwards, bottom-right backwards)
    for (int t = g; t < total; t += GL) {
        int src;
        if (t < n0) src = tl_stack[t];
        else if (t < n0 + n1) src = tr_stack[tr_count - 1 - (t - n0)];
        else if (t < n0 + n1 + n2) src = bl_stack[t - n0 - n1];
        else src = br_stack[br_count - 1 - (t - n0 - n1 - n2)];
        hull[t] = a[src];
    }
    wave_sync();
    return total;
}

// rotating_calipers(): tables, extremes and orientation split over the sixteen lanes; the walk ONE CALIPER PER LANE.  Caliper c (0 bottom,
// 1 right, 2 top, 3 left) sees the base vector turned by c quarter turns: dp[c] = ca * vx + cb * vy with (ca, cb) = (a, b), (-b, a),
// (-a, -b), (b, -a) -- the reference's four expressions term by term (a negation is exact and x - y == x + (-y)).  The winner is the FIRST
// caliper that attains the largest cosine (the reference's strict `>` scan from 0).  The winner alone computes the new base from its
// edge -- (lx, ly), (ly, -lx), (-lx, -ly), (-ly, lx) by caliper -- and the quad takes it as a bit pattern OR-ed across the lanes (the
// others contribute zero bits: exact, signed zeros included).  A caliper's SUCCESSOR entry sits in registers, so the one read a step
// needs (the new successor of the caliper that moved) is issued a whole step before its first possible use.
struct CalEntry { float vx, vy, il, px, py; };
__device__ __forceinline__ void rotating_calipers_q4(const F2 *points, int n, float4 *ev, float *out) {
    const int g = threadIdx.x & (GL - 1), c = g & 3;
    float lx = 0, rx = 0, ty = 0, by = 0;
    int li = -1, ri = -1, ti = -1, bi = -1;
    for (int i = g; i < n; i += GL) {
        const F2 pt0 = points[i];
        if (li < 0 || pt0.x < lx) { lx = pt0.x; li = i; }
        if (ri < 0 || pt0.x > rx) { rx = pt0.x; ri = i; }
        if (ti < 0 || pt0.y > ty) { ty = pt0.y; ti = i; }
        if (bi < 0 || pt0.y < by) { by = pt0.y; bi = i; }
        const F2 pt = points[(i + 1) < n ? (i + 1) : 0];
        const double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
        ev[i] = make_float4((float)dx, (float)dy, (float)(1. / sqrt(dx * dx + dy * dy)), 0.f);
    }
    int left, right, top, bottom;
    group_first_extreme(lx, li, false, &left);
    group_first_extreme(rx, ri, true, &right);
    group_first_extreme(ty, ti, true, &top);
    group_first_extreme(by, bi, false, &bottom);
    wave_sync();
    // orientation: sign of the first non-zero turn (edge i-1 -> edge i), i = 0 .. n-1
    int first = 0x7fffffff;
    float orientation = 0;
    for (int i = g; i < n && first == 0x7fffffff; i += GL) {
        const float4 a = ev[i ? i - 1 : n - 1], b = ev[i];
        const double convexity = (double)a.x * (double)b.y - (double)a.y * (double)b.x;
        if (convexity != 0) { first = i; orientation = convexity > 0 ? 1.f : -1.f; }
    }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) {
        const int f2 = __shfl_xor(first, o, GL);
        const float s2 = __shfl_xor(orientation, o, GL);
        if (f2 < first) { first = f2; orientation = s2; }
    }
    auto entry = [&](int i) { const float4 e = ev[i]; const F2 p = points[i]; CalEntry r; r.vx = e.x; r.vy = e.y; r.il = e.z; r.px = p.x; r.py = p.y; return r; };
    auto succ = [&](int i) { return i + 1 == n ? 0 : i + 1; };
    auto bits = [](float v) { return __builtin_bit_cast(int, v); };
    auto flt = [](int v) { return __builtin_bit_cast(float, v); };
    // this lane's quarter turn: which of (a, b) it multiplies with vx / vy and with which sign; which of (lx, ly) its new base takes
    const bool swap = (c & 1) != 0;
    const int sgn_ca = (c == 1 || c == 2) ? (int)0x80000000 : 0, sgn_cb = c >= 2 ? (int)0x80000000 : 0;
    const int sgn_na = c >= 2 ? (int)0x80000000 : 0, sgn_nb = (c == 1 || c == 2) ? (int)0x80000000 : 0;
    const int my_bit = 1 << c, lower = my_bit - 1;
    float minarea = 3.402823466e+38f;
    float base_a = orientation, base_b = 0;
    float buf_a = 0, buf_w = 0, buf_b = 0, buf_h = 0, bpx = 0, bpy = 0;       // bpx / bpy: MY caliper's vertex at the best step
    int s = c == 0 ? bottom : c == 1 ? right : c == 2 ? top : left;
    CalEntry e = entry(s), nx = entry(succ(s));
    for (int k = 0; k < n; k++) {
        const float ca = flt(bits(swap ? base_b : base_a) ^ sgn_ca), cb = flt(bits(swap ? base_a : base_b) ^ sgn_cb);
        const float dp = ca * e.vx + cb * e.vy;
        const float cosv = dp * e.il;
        float m = cosv;
        { const float o = qperm<QP_SWAP1>(m); m = o > m ? o : m; }
        { const float o = qperm<QP_SWAP2>(m); m = o > m ? o : m; }
        int eq = cosv == m ? my_bit : 0;
        int mask = eq | qperm_i<QP_SWAP1>(eq);
        mask |= qperm_i<QP_SWAP2>(mask);
        const bool win = eq != 0 && (mask & lower) == 0;         // the first caliper at the maximum
        const float lead_x = e.vx * e.il, lead_y = e.vy * e.il;
        int na = win ? bits(swap ? lead_y : lead_x) ^ sgn_na : 0;
        int nb = win ? bits(swap ? lead_x : lead_y) ^ sgn_nb : 0;
        na |= qperm_i<QP_SWAP1>(na); na |= qperm_i<QP_SWAP2>(na);
        nb |= qperm_i<QP_SWAP1>(nb); nb |= qperm_i<QP_SWAP2>(nb);
        base_a = flt(na); base_b = flt(nb);
        if (win) {
            s = succ(s);
            e = nx;
            nx = entry(succ(s));                                 // consumed in a later step at the earliest
        }
        float dx = qperm<QP_B1>(e.px) - qperm<QP_B3>(e.px);
        float dy = qperm<QP_B1>(e.py) - qperm<QP_B3>(e.py);
        const float width = dx * base_a + dy * base_b;
        dx = qperm<QP_B2>(e.px) - qperm<QP_B0>(e.px);
        dy = qperm<QP_B2>(e.py) - qperm<QP_B0>(e.py);
        const float height = -dx * base_b + dy * base_a;
        const float area = width * height;
        if (area <= minarea) {
            minarea = area;
            bpx = e.px; bpy = e.py; buf_a = base_a; buf_w = width; buf_b = base_b; buf_h = height;
        }
    }
    const float bl_x = qperm<QP_B3>(bpx), bl_y = qperm<QP_B3>(bpy), bb_x = qperm<QP_B0>(bpx), bb_y = qperm<QP_B0>(bpy);
    const float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
    const float C1 = A1 * bl_x + bl_y * B1;
    const float C2 = A2 * bb_x + bb_y * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * buf_w; out[3] = B1 * buf_w;
    out[4] = A2 * buf_h; out[5] = B2 * buf_h;
}

// minAreaRect of A.pts[0 .. n) (sorted by (x, y)); the result is valid on all sixteen lanes
__device__ __forceinline__ void stamp(long long *st, int i) {
    if (st && (threadIdx.x & 63) == 0) st[i] = (long long)__builtin_amdgcn_s_memtime();
}
// the same on the 100 MHz clock all XCDs share (s_memtime counts per XCD from different origins): for timelines across the chip
__device__ __forceinline__ void stamp_rt(long long *st, int i) {
    if (st && (threadIdx.x & 63) == 0) st[i] = (long long)__builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ RRect min_area_rect_q4(QuadArena &A, int n, long long *st = nullptr) {
    RRect box; box.cx = box.cy = box.w = box.h = box.angle = 0.f;
    if (n <= 0) return box;                                     // uniform over the group
    const int hn = convex_hull_sorted_q4(A.pts, n, A.hull, A.stack);
    stamp(st, 0);
    if (hn > 2) {
        float out[6] = {0, 0, 0, 0, 0, 0};
        rotating_calipers_q4(A.hull, hn, A.ev, out);
        stamp(st, 1);
        box.cx = out[0] + (out[2] + out[4]) * 0.5f;
        box.cy = out[1] + (out[3] + out[5]) * 0.5f;
        box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box.angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (hn == 2) {
        box.cx = (A.hull[0].x + A.hull[1].x) * 0.5f;
        box.cy = (A.hull[0].y + A.hull[1].y) * 0.5f;
        const double dx = A.hull[1].x - A.hull[0].x, dy = A.hull[1].y - A.hull[0].y;
        box.w = (float)sqrt(dx * dx + dy * dy);
        box.h = 0;
        box.angle = (float)atan2(dy, dx);
    } else if (hn == 1) {
        box.cx = A.hull[0].x; box.cy = A.hull[0].y;
    }
    box.angle = (float)(box.angle * 180 / PT_PI);
    return box;
}

// clipper_offset_round() with one corner per lane (lanes 0..3 of the group).  OffsetPoint(j, k) names the previous corner k, and its early
// return (edges almost in line) leaves k where it was: that chain is walked first, on the normals alone, by every lane; then lane j rounds
// corner j -- its own atan2, its own X/Y recurrence, step after step as the reference -- and the lanes store their points behind each other.
// Returns the number of points (> cap: nothing usable stored).
__device__ __forceinline__ int clipper_offset_round_q4(const CPt *path4, double delta, F2 *out, int cap, long long *ws) {
    const int c = threadIdx.x & (GL - 1);                       // lanes 4..15 own no corner (c >= len below)
    const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc = 0.25, arc_tol = 0.25;
    CPt *src = reinterpret_cast<CPt *>(ws);
    double *nx = reinterpret_cast<double *>(ws + 8), *ny = nx + 4;
    int len = 0;
    if (c == 0) {                                               // clean-up of the closed path and its orientation: a few integer steps
        int highI = 3, j = 0;
        while (highI > 0 && path4[0].X == path4[highI].X && path4[0].Y == path4[highI].Y) highI--;
        src[0] = path4[0];
        for (int i = 1; i <= highI; i++)
            if (src[j].X != path4[i].X || src[j].Y != path4[i].Y) { j++; src[j] = path4[i]; }
        len = j < 2 ? 0 : j + 1;
        if (len) {
            double a = 0;
            for (int i = 0, k = len - 1; i < len; ++i) { a += ((double)src[k].X + src[i].X) * ((double)src[k].Y - src[i].Y); k = i; }
            if (!(-a * 0.5 >= 0))
                for (int i = 0; i < len / 2; i++) { const CPt t = src[i]; src[i] = src[len - 1 - i]; src[len - 1 - i] = t; }
        }
    }
    wave_sync();
    len = __shfl(len, 0, GL);
    if (len == 0) return 0;
    if (delta > -1.0e-20 && delta < 1.0e-20) {
        if (c == 0) for (int i = 0; i < len && i < cap; i++) { out[i].x = (float)src[i].X; out[i].y = (float)src[i].Y; }
        wave_sync();
        return len;
    }
    double yv;
    if (arc_tol > fabs(delta) * def_arc) yv = fabs(delta) * def_arc; else yv = arc_tol;
    double steps = pi / acos(1 - yv / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    // (one range reduction for both: the device libm's sincos is bit-identical to its sin and cos -- tools/dbg/sincos_probe.hip, 2^28 arguments)
    double m_sin, m_cos;
    sincos(two_pi / steps, &m_sin, &m_cos);
    const double steps_per_rad = steps / two_pi;
    if (delta < 0.0) m_sin = -m_sin;
    if (c < len) {
        const CPt p1 = src[c], p2 = src[(c + 1) % len];
        if (p2.X == p1.X && p2.Y == p1.Y) { nx[c] = 0; ny[c] = 0; }
        else {
            double Dx = (double)(p2.X - p1.X), dy = (double)(p2.Y - p1.Y);
            const double f = 1 * 1.0 / sqrt(Dx * Dx + dy * dy);
            Dx *= f; dy *= f;
            nx[c] = dy; ny[c] = -Dx;
        }
    }
    wave_sync();
    // the k-chain: my corner's predecessor
    int k = len - 1, myk = len - 1;
    for (int j = 0; j < len; ++j) {
        if (j == c) myk = k;
        const double sinA = nx[k] * ny[j] - nx[j] * ny[k];
        bool done = false;
        if (fabs(sinA * delta) < 1.0) {
            const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
            if (cosA > 0) done = true;
        }
        if (!done) k = j;
    }
    // my corner: kind 0 = one point (early return), 1 = three points (concave), 2 = the arc
    int cnt = 0, kind = 0, nsteps = 0;
    double sinA = 0;
    const int j = c;
    k = myk;
    if (c < len) {
        sinA = nx[k] * ny[j] - nx[j] * ny[k];
        bool done = false;
        if (fabs(sinA * delta) < 1.0) {
            const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
            if (cosA > 0) done = true;
        } else if (sinA > 1.0) sinA = 1.0;
        else if (sinA < -1.0) sinA = -1.0;
        if (done) { kind = 0; cnt = 1; }
        else if (sinA * delta < 0) { kind = 1; cnt = 3; }
        else {
            const double ang = atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
            const long long r = cl_round(steps_per_rad * fabs(ang));
            nsteps = (int)r > 1 ? (int)r : 1;
            kind = 2; cnt = nsteps + 1;
        }
    }
    int off = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int cq = __shfl(cnt, q, GL);
        if (q < c) off += cq;
        total += cq;
    }
    if (total > cap) return total;
    if (c < len) {
        int pos = off;
        // cl_round() and the float store of a coordinate: v -> (long long)(v -+ 0.5) -> float.  The coordinates are bounded by the map
        // (|v| < 2^20), so the truncation to int (one instruction; the 64-bit conversions are emulated: ~30 each) gives the same integer
        // and the same float; the corner as doubles once, not per point
        auto rf = [](double v) { return (float)(int)(v < 0 ? v - 0.5 : v + 0.5); };
        const double sx = (double)src[j].X, sy = (double)src[j].Y;
#define PT_PUSHQ(px, py) do { out[pos].x = (px); out[pos].y = (py); pos++; } while (0)
        if (kind == 0) PT_PUSHQ(rf(sx + nx[k] * delta), rf(sy + ny[k] * delta));
        else if (kind == 1) {
            PT_PUSHQ(rf(sx + nx[k] * delta), rf(sy + ny[k] * delta));
            PT_PUSHQ((float)sx, (float)sy);
            PT_PUSHQ(rf(sx + nx[j] * delta), rf(sy + ny[j] * delta));
        } else {
            double X = nx[k], Y = ny[k], X2;
            for (int s = 0; s < nsteps; ++s) {
                PT_PUSHQ(rf(sx + X * delta), rf(sy + Y * delta));
                X2 = X;
                X = X * m_cos - m_sin * Y;
                Y = X2 * m_sin + Y * m_cos;
            }
            PT_PUSHQ(rf(sx + nx[j] * delta), rf(sy + ny[j] * delta));
        }
#undef PT_PUSHQ
    }
    wave_sync();
    return total;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }


// ------------------------------------------------------------------------------------------ per-border stages
// Public statuses 0..5 (ptocr_hip.h); internal ones mark where a border stands in the stage pipeline.
enum { ST_OK = 0, ST_SKIP_NPTS = 1, ST_SKIP_SSID = 2, ST_SKIP_SCORE = 3, ST_SKIP_UNCLIP = 4, ST_SKIP_SSID2 = 5, ST_NONE = 6,
       ST_DEFER = 7,          // left by a small-footprint stage for the full-size pass (never visible after a call)
       ST_PEND_RECT = 8, ST_PEND_SCORE = 9, ST_PEND_UNCLIP = 10 };

struct Result { int status; int box[8]; float score; float rect[5]; int npix; float distance; };

constexpr int MAXW = 2048;                // widest map the column tables hold
constexpr int LDS_PLANE_WORDS = 4096;     // full-size pass: mask planes of 131072 pixels in LDS (larger masks go through them in bands)
constexpr int MAXHULL = 512;              // full-size pass: strict hull vertices of a lattice polygon inside 2048 x 32767 stay far below
constexpr int S_MH = 96;                  // row pitch of the hull-candidate table (hin)

__device__ __forceinline__ long long cross3(int ax, int ay, int bx, int by, int px, int py) {
    return (long long)(bx - ax) * (py - ay) - (long long)(by - ay) * (px - ax);
}

// block sum in a FIXED tree order (lanes by shuffle, then the waves in order): sh = blockDim.x / 64 entries (<= 16)
template <typename T>
__device__ T block_reduce_sum(T v, T *sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    T r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) r += sh[w];
    __syncthreads();
    return r;
}

// fillPoly(mask, {border polygon}, 1, lineType=1) over the bounding box + cv::mean(pred, mask), from the border's STATES: a
// polygon edge of the CHAIN_APPROX_SIMPLE contour is a run of equal unit steps, and both things fillPoly derives from an edge
// decompose over its unit steps -- the even-odd crossing of a non-horizontal edge on rows ya <= y < yb (FillEdgeCollection's
// half-open rule) is one toggle at the upper end of every unit step, and the 4-connected edge line (cv::Line, connectivity 1
// -> 4, walked from the left end: x step first, then y step) is the step's two end pixels plus, for a diagonal step, the
// pixel beside its left end.  mask row = prefix_xor(crossings) | edge pixels (a span [c1, c2] of the reference = parity bits
// c1 .. c2-1 plus edge pixel c2).
// The two bit planes live in LDS (`plane_words` words each); a mask that does not fit is built and summed in BANDS of rows, every
// band scanning the state list for the steps that touch it (rows are independent: the parity never crosses a row).  The masked
// sum reads the probability map fully coalesced: a half-wave takes the 32 pixels of one mask word.
// raster == false: returns sum (double; fixed order: per-thread over its words, then the block tree) and pixel count.
// raster == true: the sum is taken by ONE lane in raster order, the order of cv::mean (only when the decision is within rounding).
__device__ __forceinline__ void stamp_rt(long long *st, int i);
template <int UF = 8>                                           // 16-byte map loads in flight per lane in the masked sum
__device__ void score_mask(const unsigned *st, int n, int xmin, int ymin, int bw, int bh, int row_lo, int row_hi, unsigned *border, unsigned *toggle,
                           int plane_words, const float *pimg, int W, double *red_d, int *red_i, bool raster, double *sum_out, int *cnt_out, long long *ts = nullptr) {
    const int tid = threadIdx.x, nt = blockDim.x;
    stamp_rt(ts, 0);
    const int pw = (bw + 31) >> 5;
    const int R = plane_words / pw;                             // rows per band (>= 1: callers size the planes for the widest map)
    double s = 0; int cnt = 0;
    for (int r0 = row_lo; r0 < row_hi; r0 += R) {               // rows [row_lo, row_hi) of the bounding box (the whole mask: 0, bh)
        const int rows = min(R, row_hi - r0);
        for (int i = tid; i < rows * pw; i += nt) { border[i] = 0; toggle[i] = 0; }
        __syncthreads();
#if defined(SC_DBG) && (SC_DBG & 2)
        if (n < 0)
#endif
        for (int i0 = 0; i0 < n; i0 += 8 * nt)
#pragma unroll
        for (int u8 = 0; u8 < 8; u8++) {                        // (the compiler hoists the eight loads of a round: independent addresses, no store between them)
            const int i = i0 + u8 * nt + tid;
            if (i >= n) continue;
            const unsigned sv = st[i];
            const int so = st_out(sv);
            const int x0 = st_x(sv) - xmin, y0 = st_y(sv) - ymin - r0;                 // row relative to the band
            const int dx = dir_dx(so), dy = dir_dy(so);
            if ((unsigned)y0 < (unsigned)rows) atomicOr(&border[y0 * pw + (x0 >> 5)], 1u << (x0 & 31));       // this end (the other end is the next state's)
            if (dy != 0) {
                const int xu = dy > 0 ? x0 : x0 + dx, yu = dy > 0 ? y0 : y0 - 1;       // upper end of the step
                if ((unsigned)yu < (unsigned)rows) atomicXor(&toggle[yu * pw + (xu >> 5)], 1u << (xu & 31));
                if (dx != 0) {                                                       // diagonal: the pixel right of the left end
                    const int xc = dx > 0 ? x0 + 1 : x0, yc = dx > 0 ? y0 : y0 + dy;
                    if ((unsigned)yc < (unsigned)rows) atomicOr(&border[yc * pw + (xc >> 5)], 1u << (xc & 31));
                }
            }
        }
        __syncthreads();
        stamp_rt(ts, 1);
#if defined(SC_DBG) && (SC_DBG & 4)
        if (n < 0)
#endif
        for (int y = tid; y < rows; y += nt) {
            unsigned carry = 0;
            for (int w = 0; w < pw; w++) {
                unsigned t = toggle[y * pw + w];
                t ^= t << 1; t ^= t << 2; t ^= t << 4; t ^= t << 8; t ^= t << 16;
                t ^= carry;
                carry = (t >> 31) ? 0xffffffffu : 0u;
                border[y * pw + w] |= t;
            }
        }
        __syncthreads();
        stamp_rt(ts, 2);
#if defined(SC_DBG) && (SC_DBG & 1)
        if (n >= 0) { cnt = 1; } else
#endif
        if (raster) {
            if (tid == 0)
                for (int y = 0; y < rows; y++)
                    for (int w = 0; w < pw; w++) {
                        unsigned m = border[y * pw + w];
                        const float *prow = pimg + (long)(y + r0 + ymin) * W + xmin + w * 32;
                        while (m) { const int b = __ffs(m) - 1; m &= m - 1; s += (double)prow[b]; }
                    }
        } else {
            // eight lanes per mask word, four consecutive pixels (one 16-byte load) per lane: a wave instruction covers eight words, and
            // UF of them are in flight per lane.  The stage is a chain of memory round trips, not bytes (round 4: a 256-word band took a
            // wave eight dependent trips at UF = 4, 7 us median and 47 us in the tail), so the loads are straight-line code -- every
            // lane loads, from a harmless address when its nibble is empty -- and the word index advances without a division.
            // (round 6, measured and dropped: the first round's loads requested by geometry BEFORE the mask is built, so that the item's two
            // memory round trips -- states, then pixels -- overlap: 8 loads ahead 72.5 -> 76.0 us for the stage launch, 16 ahead 97 us; the
            // role is bound by the scattered 16-byte requests it issues, and the nibbles outside the polygons are requests too)
            struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
            const int grp = tid >> 3, q = tid & 7;
            const int ng = nt >> 3, nw = rows * pw;
            const int qs = ng / pw, rs = ng - qs * pw;           // a step of ng words = qs rows + rs words
            const float *base = pimg + (long)(r0 + ymin) * W + xmin + 4 * q;
            for (int i0 = grp; i0 < nw; i0 += UF * ng) {
                F4 v[UF];
                unsigned mm[UF];
                int y = i0 / pw, w = i0 - y * pw;
                bool any_edge = false;
#pragma unroll
                for (int u = 0; u < UF; u++) {
                    const int i = i0 + u * ng;
                    const unsigned m = i < nw ? (border[i] >> (4 * q)) & 15u : 0u;
                    // the last word of a row may reach past the map's right edge: its bits there are clear, and a 16-byte load that would
                    // cross the end of the row is not issued (those few pixels are added one by one below)
                    const bool edge = xmin + w * 32 + 4 * q + 3 >= W;
                    any_edge |= edge && m;
                    mm[u] = edge ? 0u : m;
                    v[u] = *reinterpret_cast<const F4 *>(mm[u] ? base + (y * W + w * 32) : pimg);
                    y += qs; w += rs;
                    if (w >= pw) { w -= pw; y++; }
                }
#pragma unroll
                for (int u = 0; u < UF; u++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) s += (double)(((mm[u] >> e) & 1u) ? v[u].v[e] : 0.f);      // (+ 0.0 leaves a sum of non-negative terms as it is)
                    cnt += __popc(mm[u]);
                }
                if (any_edge) {
                    y = i0 / pw; w = i0 - y * pw;
                    for (int u = 0; u < UF; u++) {
                        const int i = i0 + u * ng;
                        if (i < nw && xmin + w * 32 + 4 * q + 3 >= W) {
                            const unsigned m = (border[i] >> (4 * q)) & 15u;
                            for (int e = 0; e < 4; e++)
                                if ((m >> e) & 1u) { s += (double)base[y * W + w * 32 + e]; cnt++; }
                        }
                        y += qs; w += rs;
                        if (w >= pw) { w -= pw; y++; }
                    }
                }
            }
        }
        __syncthreads();
    }
    stamp_rt(ts, 3);
    if (raster) { *sum_out = s; return; }                        // valid on thread 0
    *sum_out = block_reduce_sum<double>(s, red_d);
    *cnt_out = block_reduce_sum<int>(cnt, red_i);
    stamp_rt(ts, 4);
}

// score of a border (db_postprocess.cpp:194-229 + the float compare of :272): parallel sum; when the score lands within 1e-6 of
// box_thresh the mask is rebuilt and summed in cv::mean's raster order, so the decision is the reference's.  sh_tie: one LDS double.
__device__ float border_score(const unsigned *st, int n, int xmin, int ymin, int bw, int bh, unsigned *border, unsigned *toggle, int plane_words,
                              const float *pimg, int W, double *red_d, int *red_i, double *sh_tie, float box_thresh, int *flag_word, int *npix_out) {
    double total; int npix;
    score_mask(st, n, xmin, ymin, bw, bh, 0, bh, border, toggle, plane_words, pimg, W, red_d, red_i, false, &total, &npix);
    float score = (float)(npix ? total / npix : 0.0);
    if (fabs((double)score - (double)box_thresh) <= 1e-6) {      // uniform: every thread holds the same total
        double t2; int dummy;
        score_mask(st, n, xmin, ymin, bw, bh, 0, bh, border, toggle, plane_words, pimg, W, red_d, red_i, true, &t2, &dummy);
        if (threadIdx.x == 0) { *sh_tie = t2; atomicOr(flag_word, 2); }
        __syncthreads();
        score = (float)(npix ? *sh_tie / npix : 0.0);
        __syncthreads();
    }
    *npix_out = npix;
    return score;
}

// ---- stage A (a workgroup): hull candidates of a border.  Per-column extremes of the contour points, i.e. of the states at
// which the chain turns (every strict hull vertex is one, and a text blob has few: 8 of 169 columns is typical) -> compaction of the non-empty columns (a text blob's border touches few columns) -> a column extreme
// survives only if it is a strict vertex of its chain: for the min-y chain, point i survives iff it lies strictly on the
// outer side of every chord (j, k), j < i < k; it is enough to test the chord through the steepest predecessor and the
// steepest successor.  Exact integer arithmetic, O(m^2 / 256).  Survivors come out in (x, y) order, the order
// cv::convexHull sorts to.  arena: 3 * MW ints of LDS.  Returns the number of survivors (may exceed cap: nothing beyond
// cap is stored).
// a hull candidate goes out as one 8-byte store; THROUGH = true: as a device-scope atomic store (written through the XCD's L2, so that
// a wave on another XCD that polls the border's ready flag in the same launch reads it: border_stage_kernel)
// (round 5) ... and as a TAGGED granule {x | y << 16, tag = the call's epoch}: the reader checks the tag of every point it takes, so a point
// of another call (or memory nobody wrote) can never pass for this call's, whatever order the stores become visible in.
template <bool THROUGH>
__device__ __forceinline__ void put_point(F2 *out, int pos, int x, int y, unsigned tag) {
    if (THROUGH) {
        const unsigned long long v = (unsigned long long)((unsigned)x | ((unsigned)y << 16)) | ((unsigned long long)tag << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(out + pos), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { out[pos].x = (float)x; out[pos].y = (float)y; }
}

template <int MW, int NT, bool THROUGH = false>
__device__ int hull_candidates(const unsigned *st, int n, int xmin, int bw, unsigned *arena, F2 *out, int cap, int *wave_cnt, int *sh_n, unsigned tag = 0u) {
    const int tid = threadIdx.x;
    int *col_lo = reinterpret_cast<int *>(arena);               // [MW] min y of the border pixels per column
    int *col_hi = col_lo + MW;                                  // [MW] max y
    int *col_x = col_hi + MW;                                   // [MW] x (relative to xmin) of compacted column a
    for (int i = tid; i < bw; i += NT) { col_lo[i] = 0x7fffffff; col_hi[i] = -0x7fffffff; }
    if (tid == 0) *sh_n = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 8 * NT) {                    // eight loads in flight per lane: a lone wave pays the memory latency once per round
        unsigned sv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * NT + tid; sv[u] = i < n ? st[i] : 0u; }     // 0: s_out == s_in ^ 4 is false for it, but ...
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned s = sv[u];
            if (i0 + u * NT + tid >= n || st_out(s) == (st_in(s) ^ 4)) continue;   // the chain runs straight through: not a contour point, not a hull vertex
            const int x = st_x(s) - xmin, y = st_y(s);
            atomicMin(&col_lo[x], y);
            atomicMax(&col_hi[x], y);
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    int m_cols = 0;
    if (NT == 64) {
        // one wave: compaction in place, 64 columns at a time (a compacted entry never lands right of its source, and the wave reads a
        // chunk before it writes it), only over the columns the border has
        for (int r = 0; r * 64 < bw; r++) {
            const int i = r * 64 + tid;
            const bool has = i < bw && col_lo[i] != 0x7fffffff;
            const int vlo = has ? col_lo[i] : 0, vhi = has ? col_hi[i] : 0;
            const unsigned long long bal = __ballot(has);
            const int pos = m_cols + __popcll(bal & ((1ull << lane) - 1));
            __syncthreads();
            if (has) { col_lo[pos] = i | (vlo << 11); col_hi[pos] = i | (vhi << 11); }     // packed (x, y): 11 + 15 bits; no x table in this form
            m_cols += __popcll(bal);
        }
        __syncthreads();
    } else {
        constexpr int ROUNDS = MW / NT;                 // bw <= MW
        int v_lo[ROUNDS], v_hi[ROUNDS], v_pos[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int i = r * NT + tid;
            const bool has = i < bw && col_lo[i] != 0x7fffffff;
            v_lo[r] = has ? col_lo[i] : 0; v_hi[r] = has ? col_hi[i] : 0;
            const unsigned long long bal = __ballot(has);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int pos = m_cols + __popcll(bal & ((1ull << lane) - 1));
            for (int w = 0; w < wave; w++) pos += wave_cnt[w];
            v_pos[r] = has ? pos : -1;
            for (int w = 0; w < NT / 64; w++) m_cols += wave_cnt[w];
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
            if (v_pos[r] >= 0) { col_lo[v_pos[r]] = v_lo[r]; col_hi[v_pos[r]] = v_hi[r]; col_x[v_pos[r]] = r * NT + tid; }
        __syncthreads();
    }
    if (NT == 64) {
        // One wave: the strict hull vertices of the two x-monotone chains (column minima / column maxima) by PRUNING instead of the chord
        // tests below (O(m^2) per chain: a 900-pixel-wide merged blob with a ragged edge kept its wave busy for 70 k cycles while the other
        // borders of the kernel had long finished).  A point that is not strictly outside the segment between its two current
        // neighbours lies inside the hull of the set and goes; all such points go at once (their neighbours, removed or not, belong
        // to the set); when a round removes nothing every consecutive triple is strictly convex, i.e. the chain is the hull chain.
        // Same survivors as the chord tests, in a handful of rounds of m / 64 steps.
        __shared__ unsigned long long keep_bits[MW / 64];
        int len[2];
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            int *P = ch ? col_hi : col_lo;
            int L = m_cols;
            for (;;) {
                for (int r = 0; r * 64 < L; r++) {               // who stays (reads only)
                    const int i = r * 64 + tid;
                    bool keep = false;
                    if (i < L) {
                        keep = true;
                        if (i > 0 && i < L - 1) {
                            const int a = P[i - 1], b = P[i + 1], q = P[i];
                            const long long cr = cross3(a & 0x7ff, a >> 11, b & 0x7ff, b >> 11, q & 0x7ff, q >> 11);
                            keep = ch ? cr > 0 : cr < 0;         // strictly below (max-y chain) / above (min-y chain) the chord of its neighbours
                        }
                    }
                    const unsigned long long bal = __ballot(keep);
                    if (tid == 0) keep_bits[r] = bal;
                }
                __syncthreads();
                int newL = 0;
                for (int r = 0; r * 64 < L; r++) {               // compaction in place, a chunk at a time
                    const int i = r * 64 + tid;
                    const unsigned long long bal = keep_bits[r];
                    const bool keep = (bal >> tid) & 1ull;
                    const int v = keep ? P[i] : 0;
                    const int pos = newL + __popcll(bal & ((1ull << tid) - 1));
                    __syncthreads();
                    if (keep) P[pos] = v;
                    newL += __popcll(bal);
                }
                __syncthreads();
                if (newL == L) break;
                L = newL;
            }
            len[ch] = L;
        }
        // merge in (x, y) order: at equal x the column minimum first; a column whose minimum and maximum are one point is emitted once
        const int L0 = len[0], L1 = len[1];
        auto lower = [&](const int *P, int L, int x) { int lo = 0, hi = L; while (lo < hi) { const int m = (lo + hi) >> 1; if ((P[m] & 0x7ff) < x) lo = m + 1; else hi = m; } return lo; };
        int dup_before = 0, total_dup = 0;                      // duplicates among the max-chain entries (running count per chunk)
        for (int r = 0; r * 64 < L0; r++) {
            const int j = r * 64 + tid;
            if (j < L0) {
                const int v = col_lo[j], x = v & 0x7ff;
                const int k = lower(col_hi, L1, x);
                // entries of the other chain that come before: those with smaller x, minus the duplicates among them
                int d = 0;
                for (int t = 0; t < k; t++) { const int u = col_hi[t]; const int kk = lower(col_lo, L0, u & 0x7ff); d += (kk < L0 && col_lo[kk] == u) ? 1 : 0; }
                const int pos = j + k - d;
                if (pos < cap) put_point<THROUGH>(out, pos, x + xmin, v >> 11, tag);
            }
        }
        for (int r = 0; r * 64 < L1; r++) {
            const int j = r * 64 + tid;
            bool dup = false;
            int v = 0, x = 0, kle = 0;
            if (j < L1) {
                v = col_hi[j]; x = v & 0x7ff;
                const int k = lower(col_lo, L0, x);
                dup = k < L0 && col_lo[k] == v;
                kle = k + ((k < L0 && (col_lo[k] & 0x7ff) == x) ? 1 : 0);      // min-chain entries with x' <= x
            }
            const unsigned long long bal = __ballot(dup);
            const int d = dup_before + __popcll(bal & ((1ull << tid) - 1));
            if (j < L1 && !dup) {
                const int pos = kle + j - d;
                if (pos < cap) put_point<THROUGH>(out, pos, x + xmin, v >> 11, tag);
            }
            dup_before += __popcll(bal);
        }
        total_dup = dup_before;
        __syncthreads();
        return L0 + L1 - total_dup;
    }
    for (int base = 0; base < m_cols; base += NT) {
        const int i = base + tid;
        int keep_lo = 0, keep_hi = 0, ylo = 0, yhi = 0, xi = 0;
        if (i < m_cols) {
            ylo = col_lo[i]; yhi = col_hi[i]; xi = col_x[i];
            if (i == 0 || i == m_cols - 1) { keep_lo = 1; keep_hi = yhi != ylo; }
            else {
                // min-y chain: the predecessor / successor for which i is "most hidden", found with cross products
                int bj = 0, bk = i + 1;
                for (int j = 1; j < i; j++)
                    if (cross3(col_x[j], col_lo[j], xi, ylo, col_x[bj], col_lo[bj]) > 0) bj = j;
                for (int kk = i + 2; kk < m_cols; kk++)
                    if (cross3(xi, ylo, col_x[kk], col_lo[kk], col_x[bk], col_lo[bk]) > 0) bk = kk;
                // strict vertex of the min-y chain <=> i strictly above (smaller y) the chord bj -> bk: cross(bj, bk, i) < 0
                keep_lo = cross3(col_x[bj], col_lo[bj], col_x[bk], col_lo[bk], xi, ylo) < 0;
                bj = 0; bk = i + 1;
                for (int j = 1; j < i; j++)
                    if (cross3(col_x[j], col_hi[j], xi, yhi, col_x[bj], col_hi[bj]) < 0) bj = j;
                for (int kk = i + 2; kk < m_cols; kk++)
                    if (cross3(xi, yhi, col_x[kk], col_hi[kk], col_x[bk], col_hi[bk]) < 0) bk = kk;
                keep_hi = cross3(col_x[bj], col_hi[bj], col_x[bk], col_hi[bk], xi, yhi) > 0;
                if (yhi == ylo && keep_lo) keep_hi = 0;          // one point, emit once
            }
        }
        const int mine = keep_lo + keep_hi;
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        int pos = *sh_n + incl - mine;
        for (int w = 0; w < wave; w++) pos += wave_cnt[w];
        if (keep_lo) { if (pos < cap) put_point<THROUGH>(out, pos, xi + xmin, ylo, tag); pos++; }
        if (keep_hi) { if (pos < cap) put_point<THROUGH>(out, pos, xi + xmin, yhi, tag); pos++; }
        __syncthreads();
        if (tid == NT - 1) *sh_n = pos;
        __syncthreads();
    }
    return *sh_n;
}

// ---- stage B (one lane): min-area rectangle of the border, mini-box, first size filter (db_postprocess.cpp:259-265)
__device__ int rect_finish(const RRect &box, Result *res, float (*mini)[2]) {
    float ssid;
    get_mini_boxes(box, mini, &ssid);
    res->rect[0] = box.cx; res->rect[1] = box.cy; res->rect[2] = box.w; res->rect[3] = box.h; res->rect[4] = box.angle;
    return ssid < 3 ? ST_SKIP_SSID : ST_PEND_SCORE;             // min_size, db_postprocess.cpp:265
}
__device__ int rect_stage(const F2 *cand, int n, F2 *hull, int *stack, float *scratch, Result *res, float (*mini)[2]) {
    return rect_finish(min_area_rect_sorted(cand, n, hull, stack, scratch), res, mini);
}

// ---- stage D, first half (one lane): UnClip's distance and Clipper's round offset of the truncated mini-box
// (db_postprocess.cpp:16-49).  Returns the number of offset points (<= 0: empty; > cap: does not fit, nothing usable stored).
__device__ int unclip_offset(const float (*mini)[2], float unclip_ratio, F2 *pts, int cap, Result *res, int *flag_word, long long *ws) {
    float area = 0.0f, dist = 0.0f;
    for (int i = 0; i < 4; i++) {
        const int nn = (i + 1) % 4;
        area += mini[i][0] * mini[nn][1] - mini[i][1] * mini[nn][0];
        dist += sqrtf((mini[i][0] - mini[nn][0]) * (mini[i][0] - mini[nn][0]) +
                      (mini[i][1] - mini[nn][1]) * (mini[i][1] - mini[nn][1]));
    }
    area = (float)fabs((double)(float)(area / 2.0));
    const float distance = area * unclip_ratio / dist;
    res->distance = distance;
    CPt *path = reinterpret_cast<CPt *>(ws + 16);
    for (int i = 0; i < 4; i++) { path[i].X = (long long)(int)mini[i][0]; path[i].Y = (long long)(int)mini[i][1]; }
    return clipper_offset_round(path, (double)distance, pts, cap, ws);
}

// ---- stage D, second half (one lane): minAreaRect of the offset polygon (pts sorted by (x, y)), the two filters and the final
// box (db_postprocess.cpp:57-64, 276-311)
__device__ int unclip_box(const RRect &ub, Result *res, int src_w, int src_h, int use_padding_resize, const DbpostDims &d) {
    if (ub.h < 1.001 && ub.w < 1.001) return ST_SKIP_UNCLIP;
    float clip[4][2], ssid;
    get_mini_boxes(ub, clip, &ssid);
    if (ssid < 5) return ST_SKIP_SSID2;                        // min_size + 2
    for (int j = 0; j < 4; j++) {
        if (use_padding_resize) {
            // get_affine_transform(center, max(src_w, src_h), H, inv=1) + transform_preds (db_postprocess.cpp:111-145, 289-301):
            // a uniform scale about the centres, evaluated in double on the float32 triangle coordinates
            const float cx = (float)(src_w / 2.0), cy = (float)(src_h / 2.0);
            const int img_maxsize = src_w > src_h ? src_w : src_h;
            const float s1y = cy + (float)((float)img_maxsize / 2.0);
            const float d0 = (float)((float)d.H / 2.0), d1y = d0 + (float)((float)d.H / 2.0);
            const double scale = ((double)s1y - (double)cy) / ((double)d1y - (double)d0);
            const float tx = (float)(scale * ((double)clip[j][0] - (double)d0) + (double)cx);
            const float ty = (float)(scale * ((double)clip[j][1] - (double)d0) + (double)cy);
            res->box[2 * j]     = (int)clampf(roundf(tx), 0, (float)src_w);
            res->box[2 * j + 1] = (int)clampf(roundf(ty), 0, (float)src_h);
        } else {
            res->box[2 * j]     = (int)clampf(roundf(clip[j][0] / (float)d.W * (float)src_w), 0, (float)src_w);
            res->box[2 * j + 1] = (int)clampf(roundf(clip[j][1] / (float)d.H * (float)src_h), 0, (float)src_h);
        }
    }
    return ST_OK;
}

__device__ int unclip_finish(const F2 *pts, int np, F2 *hull, int *stack, float *scratch, Result *res, int src_w, int src_h,
                             int use_padding_resize, const DbpostDims &d) {
    RRect ub;
    if (np <= 0) { ub.cx = 0; ub.cy = 0; ub.w = 1; ub.h = 1; ub.angle = 0; }
    else ub = min_area_rect_sorted(pts, np, hull, stack, scratch);
    return unclip_box(ub, res, src_w, src_h, use_padding_resize, d);
}

struct ScorePart { double sum; int cnt; int pad; };
struct StageArgs {
    const float *maps; const Cand *cands; const int *totals; const Acc *acc; const unsigned *pool;
    Result *results; int *flags; const int *src_wh; F2 *hin; float *mini;
    long long *stamps;                      // timing experiments: s_memtime stamps of the stage kernels' phases, 16 per record (null: none)
    int *ready; int epoch; int *tie;        // per border: the hull role's ready word (epoch << 9 | candidates << 2 | state) for the quad role; the score's tie marker
    const int *sc_off; const int *sc_n; const int *sc_item;   // score bands: first item of every border, items per image, item -> border | band << 10
    ScorePart *sc_part; long sc_cap;        // partial sums per item (a fixed slice of sc_cap items per image)
    int *sc_done;                           // per border: bands finished (returns to zero by itself)
    float box_thresh, unclip_ratio; int use_padding_resize;
};


// ---- the per-border stages (round 4).  Rounds 2-3 gave a border ONE wave for its hull candidates and then its BoxScore mask
// (db_postprocess.cpp:194-229): 78 us per call = the 160 KB masked sum of the LARGEST border (a merged blob 900 x 45 pixels) by one wave,
// and then a second launch of 54 us for the geometry (four lanes per border).  What a border needs splits into independent chains:
//   * hull candidates (one wave per border) -> geometry: rectangle, size filter, Clipper offset, second rectangle, final box
//     (four lanes per border, ~100 k cycles of dependent float / double code whatever the border's size);
//   * score: needs only the border's states and its bounding box, and its result enters the reference's control flow in ONE place, the
//     `score < box_thresh` filter (db_postprocess.cpp:272); a mask's rows are independent (the crossing parity never leaves a row).
// border_stage_kernel runs all three as ROLES of one launch, in grid order: hull blocks, quad blocks, score blocks.
//   * hull role: one wave per border; the candidates go out as device-scope stores, then the border's ready word.
//   * quad role: sixteen lanes per border, four borders per wave; a quad polls its border's ready word (written by a block that was
//     dispatched BEFORE it -- workgroups are dispatched in grid order, so what a quad waits for is running or done; the wait is
//     bounded all the same: a quad that gives up defers its border to the full-size pass) and starts while other hulls and the
//     scores are still being computed: the launch lasts about as long as hull + geometry of one border, not hull of all, then score
//     of all, then geometry.  The geometry assumes that the score passes; compact_image applies the filter where the reference does
//     (after the size filter, before the unclip filters) when it writes the boxes out.
//   * score role: a mask as BANDS of rows of at most BAND_WORDS words, one band per wave, planned by pool_offsets_kernel from the
//     bounding boxes: the merged blob is six waves' work.  The last band of a border to finish (a ticket per border) adds the partial
//     sums in band order -- a fixed order, whoever finishes last.
// Ready words carry the call's epoch, so nothing is cleared between calls and a word left over by a call that was cut short is ignored.
constexpr int WAVE_NT = 64;               // threads per border in the hull kernel
#ifndef PT_STAGE_GRID
#define PT_STAGE_GRID 256
#endif
#ifndef PT_SCORE_GRID
#define PT_SCORE_GRID 256
#endif
constexpr int STAGE_GRID = PT_STAGE_GRID;           // hull-role blocks per image (a text-like map has ~150-300 borders of the 1000 slots)
constexpr int SCORE_GRID = PT_SCORE_GRID;
constexpr int QUADS = 4;                   // borders per wave of the quad role (LDS: one QuadArena each)
constexpr int QUAD_BLOCKS = (MAX_CAND + QUADS - 1) / QUADS;      // quad-role blocks per image
constexpr int POINT_SPINS = 1 << 10;       // re-polls of one hull candidate whose tag is not this call's before the quad defers the border
constexpr int READY_SPINS = 1 << 16;       // polls (with s_sleep) before a quad gives up: ~50 ms, a thousand times the launch           // score-role blocks per image (each walks the image's band items with this stride)

// hull role.  Terminal statuses are written here; a border that goes on to the geometry gets no status from this role (the quad
// role, or the full-size pass if the quad gave up, writes it).  The border's ready word = epoch << 9 | candidates << 2 | 1 (go on) or 2 (done here).
__device__ __forceinline__ void border_hull_body(const StageArgs &a, const DbpostDims &d, int img, int k, unsigned *arena) {
    const int tid = threadIdx.x;
    const long bi = (long)img * MAX_CAND + k;
    Result *res = &a.results[bi];
    const Acc ac = a.acc[bi];
    int state = 2, n_pub = 0;
    if (a.flags[img] & 4) { if (tid == 0) res->status = ST_NONE; }
    else if (ac.npts <= 2) { if (tid == 0) res->status = ST_SKIP_NPTS; }     // db_postprocess.cpp:255
    else {
        const int bw = ac.xmax - ac.xmin + 1, bh = ac.ymax - ac.ymin + 1;
        // Speckle: both sides of a min-area rectangle are projections of the point set, so neither exceeds its diameter, which is at most
        // the diagonal of the bounding box; a diagonal <= sqrt(8) means ssid < 3 (db_postprocess.cpp:265) without computing the rectangle.
        if (border_is_tiny(bw, bh)) { if (tid == 0) res->status = ST_SKIP_SSID; }
        else if (bw > W_MW) { if (tid == 0) { res->status = ST_DEFER; atomicOr(&a.flags[img], 8); } }
        else {
            __shared__ int wave_cnt[WAVE_NT / 64];
            __shared__ int sh_n;
            const unsigned *st = a.pool + (long)img * d.pool_cap + ac.off;
            long long *ts = a.stamps ? a.stamps + ((long)img * MAX_CAND + k) * 16 + 12 : nullptr;   // slots 12..13 of the border's record
            stamp_rt(ts, 0);
            const int n = hull_candidates<W_MW, WAVE_NT, true>(st, ac.nstates, ac.xmin, bw, arena, a.hin + bi * S_MH, Q_PTS, wave_cnt, &sh_n, (unsigned)a.epoch);
            stamp_rt(ts, 1);
            if (n > Q_PTS) { if (tid == 0) { res->status = ST_DEFER; atomicOr(&a.flags[img], 8); } }
            else { n_pub = n; state = 1; }
        }
    }
    // The candidates are out as write-through (sc1) stores; every lane waits for its own (the asm form: the compiler may drop a
    // builtin s_waitcnt it believes redundant, MI355X_MICROARCH.md "Compiler hazard"), the block's barrier, then ONE word carries
    // everything the quad needs to know: epoch << 9 | count << 2 | state.  The count travels INSIDE the epoch-tagged word, so it cannot be
    // older than the flag, and every candidate carries the epoch as well (put_point): the hand-off validates itself.  (The guide has
    // measured the drained-sc1 form for one workgroup per CU; here a dozen 64-thread blocks share a CU -- hence tags instead of trust.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.ready[bi], (a.epoch << 9) | (n_pub << 2) | state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- score role: one band of one border (one wave)
template <int UF = SCORE_UF>
__device__ __forceinline__ void score_band_item(const StageArgs &a, const DbpostDims &d, int img, int item, unsigned *planes) {
    const int tid = threadIdx.x;
    const int *off = a.sc_off + (long)img * MAX_CAND;
    const int code = a.sc_item[(long)img * a.sc_cap + item];
    const int k = code & 1023, band = code >> 10;
    const long bi = (long)img * MAX_CAND + k;
    Result *res = &a.results[bi];
    const Acc ac = a.acc[bi];                                   // (pool_offsets_kernel plans items only for borders the hull role does not filter out)
    const int bw = ac.xmax - ac.xmin + 1, bh = ac.ymax - ac.ymin + 1;
    const int R = band_rows(bw), nb = (bh + R - 1) / R;
    __shared__ double red_d[WAVE_NT / 64];
    __shared__ int red_i[WAVE_NT / 64];
    __shared__ int sh_ticket;
    const unsigned *st = a.pool + (long)img * d.pool_cap + ac.off;
    const float *pimg = a.maps + (long)img * d.HW;
    long long *ts = a.stamps ? a.stamps + ((long)img * MAX_CAND + k) * 16 + 14 : nullptr;   // slots 14..15: band 0 of the border
    if (band == 0) stamp_rt(ts, 0);
    double total; int npix;
    score_mask<UF>(st, ac.nstates, ac.xmin, ac.ymin, bw, bh, band * R, min(bh, band * R + R), planes, planes + BAND_WORDS, BAND_WORDS, pimg, d.W, red_d, red_i,
               false, &total, &npix, (a.stamps && band < 2 && img >= 4) ? a.stamps + ((long)img * MAX_CAND + k) * 16 + band * 6 : nullptr);    // (records the quads do not use)
    if (a.stamps && band < 2 && img >= 4) stamp_rt(a.stamps + ((long)img * MAX_CAND + k) * 16 + band * 6, 5);
    if (nb > 1) {
        // partial sums meet through memory: every band stores its pair, then takes a ticket; the holder of the last ticket reads them all.
        // The XCDs' L2s are not coherent with each other, and a device-scope FENCE here writes back / invalidates a whole L2 (measured:
        // with two __threadfence() per multi-band border the map loads of every wave on the chip grew tails of 40-70 us).  So the pair
        // goes out as device-scope relaxed atomic stores (write-through, no cache operation), the wave waits for them to be
        // acknowledged (a workgroup-scope release = s_waitcnt), and only then takes its ticket; the reader's loads are device-scope
        // atomics as well and depend on the ticket's value.
        ScorePart *part = a.sc_part + (long)img * a.sc_cap + off[k];
        if (tid == 0) {
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(&part[band].sum), (unsigned long long)__double_as_longlong(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part[band].cnt, npix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            sh_ticket = __hip_atomic_fetch_add(&a.sc_done[bi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const bool last = sh_ticket == nb - 1;
        __syncthreads();
        if (!last) return;
        total = 0; npix = 0;
        for (int b = 0; b < nb; b++) {                          // band order, whoever came last
            total += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&part[b].sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            npix += __hip_atomic_load(&part[b].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) __hip_atomic_store(&a.sc_done[bi], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the counters clean themselves for the next call
    }
    float score = (float)(npix ? total / npix : 0.0);
    int tie = 0;
    if (fabs((double)score - (double)a.box_thresh) <= 1e-6) {   // uniform: within rounding of the filter -> cv::mean's raster order decides
        double t2; int dummy;
        score_mask<UF>(st, ac.nstates, ac.xmin, ac.ymin, bw, bh, 0, bh, planes, planes + BAND_WORDS, BAND_WORDS, pimg, d.W, red_d, red_i, true, &t2, &dummy);
        __shared__ double sh_tie;
        if (tid == 0) sh_tie = t2;
        __syncthreads();
        score = (float)(npix ? sh_tie / npix : 0.0);
        tie = 1;
        __syncthreads();
    }
    if (tid == 0) { res->score = score; res->npix = npix; a.tie[bi] = tie; }
    if (band == 0 || nb > 1) stamp_rt(ts, 1);
}

// ---- quad role: SIXTEEN LANES per border, QUADS borders per wave (group-lane primitives above)
__device__ __forceinline__ void border_quad_body(const StageArgs &a, const DbpostDims &d, int img, int group, QuadArena *arena) {
    const int num = min(a.totals[img], MAX_CAND);
    const int c = threadIdx.x & (GL - 1), qi = threadIdx.x / GL;       // c: lane of the border's group
    const int k = group * QUADS + qi;
    if (qi >= QUADS || k >= num) return;                        // whole groups leave; nothing below spans groups
    const long bi = (long)img * MAX_CAND + k;
    // wait for the hull role's word of this call: epoch << 9 | count << 2 | state; state 1 = candidates are out, 2 = the border ended there
    int word = 0, spins = 0;
    for (;;) {
        if (c == 0) word = __hip_atomic_load(&a.ready[bi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        word = __shfl(word, 0, GL);
        if ((word >> 9) == a.epoch) break;
        if (++spins > READY_SPINS) break;
        __builtin_amdgcn_s_sleep(32);
    }
    Result *res = &a.results[bi];
    if ((word >> 9) != a.epoch) {                               // gave up (never seen): the full-size pass takes the border from its states
        if (c == 0) { res->status = ST_DEFER; atomicOr(&a.flags[img], 8); }
        return;
    }
    if ((word & 3) != 1) return;
    QuadArena &A = arena[qi];
    long long *st = (a.stamps && group < 4 * 63 && (group & 3) == 0) ? a.stamps + ((long)img * 63 + (group >> 2)) * 16 : nullptr;
    stamp(st, 0);
    // The count came with the word.  Every candidate must carry this call's epoch; one that does not (a store not yet visible, a word
    // the hull role never wrote) is polled again a bounded number of times, then the quad gives the border to the full-size pass, which
    // recomputes it from its states: a violated hand-off costs time, never a wrong box and never an index out of range.
    const int n = (word >> 2) & 127;
    bool bad = n > Q_PTS;
    for (int i = c; i < n && !bad; i += GL) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.hin + bi * S_MH + i);
        unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int tries = 0; (unsigned)(v >> 32) != (unsigned)a.epoch && tries < POINT_SPINS; tries++) {
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((unsigned)(v >> 32) != (unsigned)a.epoch) bad = true;
        A.pts[i].x = (float)(int)((unsigned)v & 0xffffu); A.pts[i].y = (float)(int)(((unsigned)v >> 16) & 0xffffu);
    }
    bad = ((__ballot(bad) >> (GL * qi)) & 0xffffull) != 0;      // any lane of this group
    if (bad) {
        if (c == 0) { res->status = ST_DEFER; atomicOr(&a.flags[img], 8); }
        return;
    }
    wave_sync();
    // minAreaRect of the border, mini-box, first size filter (db_postprocess.cpp:259-265); then UnClip and the minAreaRect of the offset
    // polygon.  The two rectangles are the two passes of ONE loop, so that the whole geometry is inlined once and every table access is
    // an LDS instruction (round 6: as a called function the arena was a generic pointer -- flat loads, each waited for at once).
    stamp(st, 1);
    int status = -1;                                            // -1: the second rectangle decides (unclip_box)
    int npts = n;
    RRect box;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && npts <= 0) { box.cx = 0; box.cy = 0; box.w = 1; box.h = 1; box.angle = 0; }
        else box = min_area_rect_q4(A, npts, st ? st + (pass ? 8 : 2) : nullptr);      // stamps 2 / 8: hull, 3 / 9: calipers
        if (pass == 1) break;
        stamp(st, 4);
        float mini[4][2], ssid;
        get_mini_boxes(box, mini, &ssid);
        stamp(st, 5);
        if (c == 0) { res->rect[0] = box.cx; res->rect[1] = box.cy; res->rect[2] = box.w; res->rect[3] = box.h; res->rect[4] = box.angle; }
        if (ssid < 3) { status = ST_SKIP_SSID; break; }         // min_size
        // (the score filter of db_postprocess.cpp:272 sits here in the reference: compact_image applies it)
        // UnClip (db_postprocess.cpp:16-49): distance from the float mini-box, Clipper's round offset of its truncated vertices
        float area = 0.0f, dist = 0.0f;
        for (int i = 0; i < 4; i++) {
            const int nn = (i + 1) % 4;
            area += mini[i][0] * mini[nn][1] - mini[i][1] * mini[nn][0];
            dist += sqrtf((mini[i][0] - mini[nn][0]) * (mini[i][0] - mini[nn][0]) +
                          (mini[i][1] - mini[nn][1]) * (mini[i][1] - mini[nn][1]));
        }
        area = (float)fabs((double)(float)(area / 2.0));
        const float distance = area * a.unclip_ratio / dist;
        if (c == 0) res->distance = distance;
        CPt *path = reinterpret_cast<CPt *>(A.cl_ws + 16);
        if (c == 0) for (int i = 0; i < 4; i++) { path[i].X = (long long)(int)mini[i][0]; path[i].Y = (long long)(int)mini[i][1]; }
        wave_sync();
        F2 *raw = A.hull;
        int np = clipper_offset_round_q4(path, (double)distance, raw, Q_PTS, A.cl_ws);
        stamp(st, 6);
        if (np > 0 && np <= Q_PTS && distance < UNION_MAX_DISTANCE) {              // Execute's union: sub-pixel slivers only
            if (c == 0) np = clipper_union_cut(raw, np, A.pts, &A.stack[0][0]);
            wave_sync();
            np = __shfl(np, 0, GL);
        }
        if (np > Q_PTS) { status = ST_DEFER; if (c == 0) atomicOr(&a.flags[img], 8); break; }
        // sort by (x, y) like cv::convexHull: every lane ranks up to four of the points.  The offset polygon has integer vertices, so a
        // point packs into one 64-bit key (x + 2^20) << 40 | (y + 2^20) << 8 | index -- the index makes equal points keep their order
        // (they are identical anyway) -- and its rank is the number of smaller keys: one compare and one add per pair
        {
            unsigned long long *keys = reinterpret_cast<unsigned long long *>(A.ev);      // the caliper table is not in use yet
            for (int i = c; i < np; i += GL) {
                const F2 t = raw[i];
                keys[i] = ((unsigned long long)((int)t.x + (1 << 20)) << 40) | ((unsigned long long)((int)t.y + (1 << 20)) << 8) | (unsigned)i;
            }
            wave_sync();
            static_assert(4 * GL >= Q_PTS, "a lane ranks at most four keys");
            const int i0 = c < np ? c : 0;                       // (a lane beyond the list ranks key 0 again and stores nothing)
            unsigned long long t[4]; int rank[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = keys[i0 + GL * u < np ? i0 + GL * u : i0];
            for (int j0 = 0; j0 < np; j0 += 8) {
                unsigned long long o[8];
#pragma unroll
                for (int v = 0; v < 8; v++) o[v] = j0 + v < np ? keys[j0 + v] : ~0ull;
#pragma unroll
                for (int v = 0; v < 8; v++)
#pragma unroll
                    for (int u = 0; u < 4; u++) rank[u] += o[v] < t[u] ? 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (c + GL * u < np) A.pts[rank[u]] = raw[c + GL * u];
        }
        wave_sync();
        stamp(st, 7);
        npts = np;
    }
    if (status < 0) {
        stamp(st, 10);
        if (c == 0) status = unclip_box(box, res, a.src_wh[2 * img], a.src_wh[2 * img + 1], a.use_padding_resize, d);
        stamp(st, 11);
    }
    if (c == 0) res->status = status;
}

// One launch, three roles, in grid order: N * STAGE_GRID hull blocks, N * QUAD_BLOCKS quad blocks, N * SCORE_GRID score blocks.
// (the hull role needs 32 VGPRs and the 8 KB column tables, the score role 2 KB and registers for loads in flight, the quad role
// 9 KB: one LDS block for all)
#ifndef PT_STAGE_ORDER
#define PT_STAGE_ORDER 0
#endif
#ifndef PT_STAGE_WAVES
#define PT_STAGE_WAVES 3                    // waves per SIMD the fused stage kernel is compiled for (4 = 128 VGPRs: the quad role's double-precision offset code spills, measured 74 against 73 us)
#endif
__global__ __launch_bounds__(WAVE_NT, PT_STAGE_WAVES) void border_stage_kernel(StageArgs a, DbpostDims d) {
    constexpr int ARENA_BYTES = (int)(QUADS * sizeof(QuadArena)) > 8 * W_MW ? (int)(QUADS * sizeof(QuadArena)) : 8 * W_MW;
    static_assert(8 * W_MW >= 8 * BAND_WORDS, "the column tables hold the two mask planes of a band");
    __shared__ __attribute__((aligned(16))) unsigned char arena_raw[ARENA_BYTES];
    unsigned *arena = reinterpret_cast<unsigned *>(arena_raw);     // hull role: column tables; score role: mask planes; quad role: the quads' arenas
    const int nh = d.N * STAGE_GRID, nq = d.N * QUAD_BLOCKS;
    int b = blockIdx.x;
#if PT_STAGE_ORDER == 1                                         // experiment: hull and score blocks alternate in front of the quads
    static_assert(STAGE_GRID == SCORE_GRID, "alternating order");
    int role;
    if (b < 2 * nh) { role = (b & 1) ? 2 : 0; b >>= 1; }
    else { role = 1; b -= 2 * nh; }
#else
    int role = 0;
    if (b >= nh) { b -= nh; role = 1; if (b >= nq) { b -= nq; role = 2; } }
#endif
    if (role == 0) {
        const int img = b / STAGE_GRID, num = min(a.totals[img], MAX_CAND);
        for (int k = b % STAGE_GRID; k < num; k += STAGE_GRID) {
            border_hull_body(a, d, img, k, arena);
            __syncthreads();
        }
        return;
    }
    if (role == 1) {
        border_quad_body(a, d, b / QUAD_BLOCKS, b % QUAD_BLOCKS, reinterpret_cast<QuadArena *>(arena_raw));
        return;
    }
    const int img = b / SCORE_GRID, items = a.sc_n[img];
    for (int item = b % SCORE_GRID; item < items; item += SCORE_GRID) {
        score_band_item(a, d, img, item, arena);
        __syncthreads();
    }
}

// boxes of one image in candidate order -> dense int16 list + count
// Last step of a call (round 6: the tail of contour_big_kernel -- it was a launch of its own, 4 us of kernel behind a launch boundary; ONE
// workgroup of 1024 threads per image runs it: workgroup 0 of the image when nothing was deferred, else the last of the image's full-size
// workgroups to finish).  It applies the score filter the quad role left out -- a border whose geometry went through the size filter
// (status OK, or one of the two unclip filters) is dropped when its score is below box_thresh, as db_postprocess.cpp:272 does before the
// unclip -- and raises the tie flag for borders whose score the reference would have computed; writes the boxes out densely; hands the
// image's flag word and strip count to the host copies and CLEARS the per-call block (flags, strip totals, strip run counts) for the
// next call -- the memset that used to open every call is gone.
struct CompactArgs { short *boxes; int *counts; int max_boxes; int *strip_totals; int *strip_runs; int *flags_out; int *strip_out; int *big_done; };
__device__ void compact_image(const StageArgs &a, const CompactArgs &c, int img, int *sh /* 1024 ints */, int *sh_tie) {
    const int k = threadIdx.x;
    Result *results = a.results;
    if (k == 0) *sh_tie = 0;
    __syncthreads();
    const int num = min(a.totals[img], MAX_CAND);
    bool ok = false;
    if (k < num) {
        Result *r = &results[(long)img * MAX_CAND + k];
        int st = r->status;
        if (st == ST_OK || st == ST_SKIP_UNCLIP || st == ST_SKIP_SSID2) {
            if (a.tie[(long)img * MAX_CAND + k]) *sh_tie = 2;     // within rounding of box_thresh: re-summed in raster order (flag bit 1)
            if (r->score < a.box_thresh) { st = ST_SKIP_SCORE; r->status = st; }
        }
        ok = st == ST_OK;
    }
    sh[k] = ok;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = k >= off ? sh[k - off] : 0;
        __syncthreads();
        sh[k] += v;
        __syncthreads();
    }
    if (ok) {
        const int pos = sh[k] - 1;
        if (pos < c.max_boxes)
            for (int j = 0; j < 8; j++) c.boxes[((long)img * c.max_boxes + pos) * 8 + j] = (short)results[(long)img * MAX_CAND + k].box[j];
    }
    if (k == 1023) c.counts[img] = sh[1023];
    if (k == 0) {
        c.flags_out[img] = a.flags[img] | *sh_tie; c.strip_out[img] = c.strip_runs[img];
        a.flags[img] = 0; c.strip_totals[img] = 0; c.strip_runs[img] = 0;
    }
}

// ---- full-size pass: a few workgroups of 1024 threads per image walk the borders the small-footprint stages deferred (borders wider
// than 1024 px, more than 96 hull / offset points: usually none; on noise maps the one giant component) through all four stages:
// the parallel parts on all 16 waves, the rectangles on wave 0 (cooperative forms above)
constexpr int BIG_THREADS = 1024;
constexpr int BIG_GRID = 8;                 // workgroups per image (they share the image's list of deferred borders; launching 64 that leave at once cost 5 us)
static_assert(BIG_THREADS == 1024, "compact_image is written for 1024 threads");
__global__ __launch_bounds__(BIG_THREADS, 1) void contour_big_kernel(StageArgs a, DbpostDims d, CompactArgs cp) {
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    __shared__ int sh_scan[1024];
    __shared__ int sh_tieflag, sh_ticket;
    if (!(a.flags[img] & 8)) {                            // internal bit 3: a small stage deferred at least one border of this image -- not set:
        if (blockIdx.x == 0) compact_image(a, cp, img, sh_scan, &sh_tieflag);      // the image is finished (the other workgroups read the bit clear whether before or after this one's reset)
        return;
    }
    constexpr int ARENA = 2 * LDS_PLANE_WORDS > 3 * MAXW ? 2 * LDS_PLANE_WORDS : 3 * MAXW;
    __shared__ __attribute__((aligned(16))) unsigned arena[ARENA];     // column tables (hull stage) / mask planes (score stage)
    __shared__ F2 cand_pts[MAXHULL];
    __shared__ F2 hull_pts[MAXHULL];
    __shared__ F2 raw_pts[MAXHULL];
    __shared__ int stack[2 * (MAXHULL + 2)];
    __shared__ float cal_scratch[3 * MAXHULL + 4];
    __shared__ long long cl_ws[24];
    __shared__ double red_d[BIG_THREADS / 64];
    __shared__ int red_i[BIG_THREADS / 64];
    __shared__ int wave_cnt[BIG_THREADS / 64];
    __shared__ int sh_n, sh_status, sh_hn, sh_np;
    __shared__ double sh_tie;
    __shared__ float sh_mini[4][2];
    const int num = min(a.totals[img], MAX_CAND);
    // the deferred borders of the image in candidate order (every workgroup of the image builds the same list: thread k looks at border k)
    __shared__ int sh_list[MAX_CAND];
    __shared__ int sh_wcnt[BIG_THREADS / 64];
    static_assert(BIG_THREADS >= MAX_CAND, "one thread per candidate");
    {
        const bool def = tid < num && a.results[(long)img * MAX_CAND + tid].status == ST_DEFER;
        const unsigned long long bal = __ballot(def);
        if ((tid & 63) == 0) sh_wcnt[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << (tid & 63)) - 1));
        for (int w = 0; w < (tid >> 6); w++) before += sh_wcnt[w];
        if (def) sh_list[before] = tid;
        __syncthreads();
    }
    int ndef = 0;
    for (int w = 0; w < BIG_THREADS / 64; w++) ndef += sh_wcnt[w];
    for (int e = blockIdx.x; e < ndef; e += gridDim.x) {
        const int k = sh_list[e];
        const long bi = (long)img * MAX_CAND + k;
        Result *res = &a.results[bi];
        if (tid == 0) a.tie[bi] = 0;                      // (this pass raises the tie flag itself; the marker may be a band score's, or stale)
        __syncthreads();                                  // the LDS below is reused from the previous border
        const Acc ac = a.acc[bi];
        const unsigned *st = a.pool + (long)img * d.pool_cap + ac.off;
        const int bw = ac.xmax - ac.xmin + 1, bh = ac.ymax - ac.ymin + 1;
        const int n = hull_candidates<MAXW, BIG_THREADS>(st, ac.nstates, ac.xmin, bw, arena, cand_pts, MAXHULL, wave_cnt, &sh_n);
        if (n > MAXHULL) { if (tid == 0) { atomicOr(&a.flags[img], 4); res->status = ST_NONE; } continue; }
        if (tid < 64) {
            const RRect box = min_area_rect_wave(cand_pts, n, hull_pts, stack, cal_scratch, &sh_hn);
            if (tid == 0) sh_status = rect_finish(box, res, sh_mini);
        }
        __syncthreads();
        if (sh_status != ST_PEND_SCORE) { if (tid == 0) res->status = sh_status; continue; }
        int npix;
        const float score = border_score(st, ac.nstates, ac.xmin, ac.ymin, bw, bh, arena, arena + LDS_PLANE_WORDS, LDS_PLANE_WORDS,
                                         a.maps + (long)img * d.HW, d.W, red_d, red_i, &sh_tie, a.box_thresh, &a.flags[img], &npix);
        __syncthreads();
        if (tid == 0) {
            res->score = score; res->npix = npix;
            if (score < a.box_thresh) { res->status = ST_SKIP_SCORE; sh_np = -2; }       // db_postprocess.cpp:272
            else {
                int n0 = max(unclip_offset(sh_mini, a.unclip_ratio, raw_pts, MAXHULL, res, &a.flags[img], cl_ws), 0);
                if (n0 > 0 && n0 <= MAXHULL && res->distance < UNION_MAX_DISTANCE) n0 = clipper_union_cut(raw_pts, n0, cand_pts, stack);
                sh_np = n0;
            }
        }
        __syncthreads();
        const int np = sh_np;
        if (np == -2) continue;
        if (np > MAXHULL) { if (tid == 0) { atomicOr(&a.flags[img], 4); res->status = ST_NONE; } continue; }
        for (int i = tid; i < np; i += BIG_THREADS) {           // sort by (x, y) like cv::convexHull: rank every point
            const F2 t = raw_pts[i];
            int rank = 0;
            for (int j = 0; j < np; j++) {
                const F2 o = raw_pts[j];
                rank += (o.x < t.x || (o.x == t.x && (o.y < t.y || (o.y == t.y && j < i)))) ? 1 : 0;
            }
            cand_pts[rank] = t;
        }
        __syncthreads();
        if (tid < 64) {
            RRect ub;
            if (np <= 0) { ub.cx = 0; ub.cy = 0; ub.w = 1; ub.h = 1; ub.angle = 0; }
            else ub = min_area_rect_wave(cand_pts, np, hull_pts, stack, cal_scratch, &sh_hn);
            if (tid == 0) res->status = unclip_box(ub, res, a.src_wh[2 * img], a.src_wh[2 * img + 1], a.use_padding_resize, d);
        }
    }
    // the image's workgroups meet: what each wrote goes out to memory (a device-scope release: this pass is rare and the chip is otherwise
    // idle, so the L2 write-back is cheap here), then a ticket; the holder of the last one reads with the L2 invalidated and compacts
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        sh_ticket = __hip_atomic_fetch_add(&cp.big_done[img], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (sh_ticket != (int)gridDim.x - 1) return;
    if (tid == 0) __hip_atomic_store(&cp.big_done[img], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (clean for the next call)
    __threadfence();
    compact_image(a, cp, img, sh_scan, &sh_tieflag);
}

}  // namespace ptocr

using namespace ptocr;

constexpr int DBPOST_STREAMS = 4;
struct ptocr_dbpost {
    int max_n, max_h, max_w;
    unsigned *bits; unsigned *bits2; int *labels; int *word_lab; int *chunk_cnt; int *chunk_roots; int *totals; int *strip_totals; Cand *cands; Acc *acc;
    unsigned *pool; F2 *hin; float *mini;
    Result *results; int *flags; int *src_wh; short *boxes; int *counts;
    int boxes_cap;
    long pool_cap;
    hipEvent_t ev0, ev1;          // device time of the last call's kernels (ptocr_dbpost_last_device_ms)
    hipStream_t sub[DBPOST_STREAMS];   // a batch runs as up to four parts side by side (run_chain)
    hipEvent_t ev_fork, ev_join[DBPOST_STREAMS];
    int timed;
    int *strip_runs;              // per image: run starts (both polarities) in the bottom strip, counted by binarize_kernel
    int strip_hint;               // run the bottom-strip labelling pass in the next call (see run_chain)
    int route;                    // ptocr_dbpost_set_route: 0 = from the history of this workspace's calls, 1 = text route, 2 = noise route
    int noise_now;                // the route this call takes
    int dirty;                    // a call did not finish: the per-call block is cleared before the next one
    unsigned noise_hist;          // bit k: the call k + 1 calls ago met a noise-like image
    int *h_strip;                 // pinned: the strip's run-start counts of the last call
    int *h_meta;                  // pinned: flags | strip counts | box counts of the last call, as one copy delivers them
    int *tickets;                 // per image: arrival counters of the kernels whose last workgroup does an image's closing step (return to zero by themselves)
    int *zeroed;                  // the per-call block: flags | strip_totals | strip_runs (max_n ints each); zero at creation, cleared again by compact_image
    int *flags_out; int *strip_out;   // what the host copies of a call read (compact_image)
    long long *stamps;            // PTOCR_DBPOST_STAMPS=1: phase time stamps of the stage kernels (max_n * MAX_CAND * 16)
    int *sc_off; int *sc_n; int *sc_item; ScorePart *sc_part; long sc_cap; int *sc_done;     // score bands: plan, partial sums, tickets (border_wave_kernel)
    int *list; int *tie;          // per border: the hull role's ready word; score tie marker
    int epoch;                    // number of the current call (ready words of other calls are ignored); never 0
    int *stage_tab;               // per tile: border_states_kernel's table (2 * WT_SLOTS ints), read by scatter_states_kernel
    uint2 *stage; int *stage_hdr; long stage_cap; // border states: staged records (a fixed slice per tile), record count per tile
};

static int dbpost_alloc(ptocr_dbpost *h, int max_n, int max_h, int max_w);

extern "C" int ptocr_dbpost_create(ptocr_dbpost_t *out, int max_n, int max_h, int max_w) {
    PT_CHECK(out && max_n > 0 && max_h > 0 && max_w > 0, "ptocr_dbpost_create: bad arguments");
    PT_CHECK(max_w <= MAXW && max_h < 32768 && (long)max_h * max_w < (1L << 29), "ptocr_dbpost_create: map larger than %d wide / 32767 high / 2^29 pixels", MAXW);
    ptocr_dbpost *h = new ptocr_dbpost();
    memset(h, 0, sizeof *h);
    if (int e = dbpost_alloc(h, max_n, max_h, max_w)) {        // an allocation failed part-way: give back what the others took
        (void)ptocr_dbpost_destroy(h);
        return e;
    }
    *out = h;
    return 0;
}

static int dbpost_alloc(ptocr_dbpost *h, int max_n, int max_h, int max_w) {
    h->max_n = max_n; h->max_h = max_h; h->max_w = max_w;
    const long hw = (long)max_h * max_w, ww = cdiv(max_w, 32);
    h->pool_cap = 4 * hw + 64;                  // a pixel has at most 4 gaps: no map can overflow this
    h->boxes_cap = MAX_CAND;
    const long nch = (hw + CHUNK - 1) / CHUNK;
    // Two kinds of buffers.  PROTOCOL state must be zero before the first call and is kept by the calls themselves (the per-call block
    // that compact_image clears, the score tickets that return to zero, ready words and candidate tags whose zero is "no call's epoch").
    // Everything else is written by a kernel of the call before any kernel of the call reads it, so its content at allocation is
    // irrelevant -- and PTOCR_DBPOST_POISON=1 (tests) fills it with 0xA5 bytes to prove that: a kernel that did read a table before it
    // was written would chase garbage indices (under tools/guard: fault) instead of the zeros fresh device memory usually holds.
    static const bool poison = getenv("PTOCR_DBPOST_POISON") && atoi(getenv("PTOCR_DBPOST_POISON")) == 1;
    auto alloc = [&](auto **p, size_t bytes, bool protocol_zero) -> int {
        PT_HIP(dev_malloc(p, bytes));
        if (protocol_zero) PT_HIP(hipMemset(*p, 0, bytes));
        else if (poison) PT_HIP(hipMemset(*p, 0xA5, bytes));
        return 0;
    };
#define DB_ALLOC(ptr, bytes, zero) do { if (int e_ = alloc(&(ptr), (bytes), (zero))) return e_; } while (0)
    DB_ALLOC(h->bits, sizeof(unsigned) * max_n * max_h * ww, false);
    DB_ALLOC(h->bits2, sizeof(unsigned) * max_n * max_h * ww, false);
    DB_ALLOC(h->labels, sizeof(int) * max_n * hw, false);
    DB_ALLOC(h->word_lab, sizeof(int) * max_n * max_h * ww, false);
    DB_ALLOC(h->chunk_cnt, sizeof(int) * max_n * nch, false);
    DB_ALLOC(h->chunk_roots, sizeof(int) * max_n * nch * ROOT_K, false);
    DB_ALLOC(h->totals, sizeof(int) * max_n, false);
    DB_ALLOC(h->zeroed, sizeof(int) * 6 * max_n, true);
    DB_ALLOC(h->tickets, sizeof(int) * 4 * max_n, true);
    // score bands: a border's mask is cut into bands of band_rows() = BAND_WORDS / pw >= BAND_WORDS / 32 rows (pw <= 32 words: wider borders
    // go to the full-size pass), so it has at most bh / (BAND_WORDS / 32) + 1 of them
    static_assert(BAND_WORDS >= 32 && BAND_WORDS % 32 == 0, "a band holds at least one row of the widest border a wave takes");
    h->sc_cap = (long)MAX_CAND * (max_h / (BAND_WORDS / 32) + 2);
    DB_ALLOC(h->sc_off, sizeof(int) * max_n * MAX_CAND, false);
    DB_ALLOC(h->sc_n, sizeof(int) * max_n, false);
    DB_ALLOC(h->sc_item, sizeof(int) * max_n * h->sc_cap, false);
    DB_ALLOC(h->sc_part, sizeof(ScorePart) * max_n * h->sc_cap, false);
    DB_ALLOC(h->sc_done, sizeof(int) * max_n * MAX_CAND, true);
    PT_HIP(hipHostMalloc(&h->h_strip, sizeof(int) * max_n));
    h->strip_hint = 1;
    h->noise_hist = 0x80u;                      // the first call takes the noise route; a text-like first batch clears it at once
    h->flags = h->zeroed; h->strip_totals = h->zeroed + max_n; h->strip_runs = h->zeroed + 2 * max_n;
    h->flags_out = h->zeroed + 3 * max_n; h->strip_out = h->zeroed + 4 * max_n; h->counts = h->zeroed + 5 * max_n;      // one block: ONE copy to the host per call
    PT_HIP(hipHostMalloc(&h->h_meta, sizeof(int) * 3 * max_n));
    if (getenv("PTOCR_DBPOST_STAMPS")) DB_ALLOC(h->stamps, sizeof(long long) * 16 * (size_t)max_n * MAX_CAND, true);
    DB_ALLOC(h->list, sizeof(int) * max_n * MAX_CAND, true);                       // ready words: epoch 0 = no call's
    DB_ALLOC(h->tie, sizeof(int) * max_n * MAX_CAND, false);
    DB_ALLOC(h->cands, sizeof(Cand) * max_n * MAX_CAND, false);
    DB_ALLOC(h->acc, sizeof(Acc) * max_n * MAX_CAND, false);
    DB_ALLOC(h->pool, sizeof(unsigned) * max_n * h->pool_cap, false);
    h->stage_cap = ((long)cdiv(max_h, 8) * cdiv(max_w, 256) + cdiv(max_h, 8) + cdiv(max_w, 256) + 1) * STAGE_TILE;      // any H x W within the workspace: cdiv(H,8) cdiv(WW,8) tiles
    DB_ALLOC(h->stage, sizeof(uint2) * max_n * h->stage_cap, false);
    DB_ALLOC(h->stage_hdr, sizeof(int) * max_n * (h->stage_cap / STAGE_TILE), false);
    DB_ALLOC(h->stage_tab, sizeof(int) * max_n * (h->stage_cap / STAGE_TILE) * 2 * WT_SLOTS, false);
    DB_ALLOC(h->hin, sizeof(F2) * (size_t)max_n * MAX_CAND * S_MH, true);          // candidate granules: tag 0 = no call's epoch (put_point)
    DB_ALLOC(h->mini, sizeof(float) * 8 * max_n * MAX_CAND, false);
    DB_ALLOC(h->results, sizeof(Result) * max_n * MAX_CAND, false);
    DB_ALLOC(h->src_wh, sizeof(int) * 2 * max_n, false);
    DB_ALLOC(h->boxes, sizeof(short) * 8 * max_n * MAX_CAND, false);
#undef DB_ALLOC
    PT_HIP(hipEventCreate(&h->ev0));
    PT_HIP(hipEventCreate(&h->ev1));
    PT_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int p = 0; p < DBPOST_STREAMS; p++) {
        PT_HIP(hipStreamCreateWithFlags(&h->sub[p], hipStreamNonBlocking));
        PT_HIP(hipEventCreateWithFlags(&h->ev_join[p], hipEventDisableTiming));
    }
    return 0;
}

extern "C" int ptocr_dbpost_destroy(ptocr_dbpost_t h) {
    if (!h) return 0;
    void *bufs[] = {h->bits, h->bits2, h->labels, h->word_lab, h->chunk_cnt, h->chunk_roots, h->totals, h->zeroed, h->tickets, h->cands, h->acc, h->pool, h->hin,
                    h->mini, h->results, h->src_wh, h->boxes, h->list, h->tie, h->stamps, h->sc_off, h->sc_n, h->sc_item, h->sc_part, h->sc_done,
                    h->stage, h->stage_hdr, h->stage_tab};
    for (void *b : bufs) (void)dev_free(b);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->h_strip) (void)hipHostFree(h->h_strip);
    if (h->h_meta) (void)hipHostFree(h->h_meta);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int p = 0; p < DBPOST_STREAMS; p++) {
        if (h->ev_join[p]) (void)hipEventDestroy(h->ev_join[p]);
        if (h->sub[p]) (void)hipStreamDestroy(h->sub[p]);
    }
    delete h;
    return 0;
}

// debug/inspection hook used by the parity tests: copies the per-candidate records of the last call
struct CandInfoOut { int npts, off; short xmin, xmax, ymin, ymax; };
extern "C" int ptocr_dbpost_debug_results(ptocr_dbpost_t h, int img, int32_t *h_total, void *h_results /* MAX_CAND x Result */,
                                          void *h_cands /* MAX_CAND x Cand */, void *h_info /* MAX_CAND x CandInfo */) {
    PT_CHECK(h && img >= 0 && img < h->max_n, "ptocr_dbpost_debug_results: bad arguments");
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemcpy(h_total, h->totals + img, sizeof(int), hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_results, h->results + (long)img * MAX_CAND, sizeof(Result) * MAX_CAND, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_cands, h->cands + (long)img * MAX_CAND, sizeof(Cand) * MAX_CAND, hipMemcpyDeviceToHost));
    static thread_local Acc accs[MAX_CAND];
    PT_HIP(hipMemcpy(accs, h->acc + (long)img * MAX_CAND, sizeof(Acc) * MAX_CAND, hipMemcpyDeviceToHost));
    CandInfoOut *o = static_cast<CandInfoOut *>(h_info);
    for (int k = 0; k < MAX_CAND; k++) {
        o[k].npts = accs[k].npts; o[k].off = accs[k].off;
        o[k].xmin = (short)accs[k].xmin; o[k].xmax = (short)accs[k].xmax; o[k].ymin = (short)accs[k].ymin; o[k].ymax = (short)accs[k].ymax;
    }
    return 0;
}

// second inspection hook: the border states of candidate k of image img (x | y << 11 | s_out << 26 | s_in << 29); returns the
// number of states through h_n (0 for a border that got no pool space: <= 2 contour points)
extern "C" int ptocr_dbpost_debug_states(ptocr_dbpost_t h, int img, int k, uint32_t *h_states, int cap, int32_t *h_n) {
    PT_CHECK(h && img >= 0 && img < h->max_n && k >= 0 && k < MAX_CAND && h_states && h_n, "ptocr_dbpost_debug_states: bad arguments");
    PT_HIP(hipDeviceSynchronize());
    Acc a;
    PT_HIP(hipMemcpy(&a, h->acc + (long)img * MAX_CAND + k, sizeof a, hipMemcpyDeviceToHost));
    *h_n = a.off >= 0 ? a.nstates : 0;
    const int n = *h_n < cap ? *h_n : cap;
    if (n > 0) PT_HIP(hipMemcpy(h_states, h->pool + (long)img * h->pool_cap + a.off, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

// timing experiments (PTOCR_DBPOST_STAMPS=1 at create time): the s_memtime stamps of the last call, 16 per record, max_n * 1000 records
extern "C" int ptocr_dbpost_debug_stamps(ptocr_dbpost_t h, int64_t *h_stamps, long n_records) {
    PT_CHECK(h && h->stamps && n_records <= (long)h->max_n * MAX_CAND, "ptocr_dbpost_debug_stamps: no stamp buffer (set PTOCR_DBPOST_STAMPS=1 before creating the workspace)");
    PT_HIP(hipDeviceSynchronize());
    if (!h_stamps) {                                            // null destination: clear the buffer (so that the next call's records stand alone)
        PT_HIP(hipMemset(h->stamps, 0, sizeof(long long) * 16 * (size_t)h->max_n * MAX_CAND));
        return 0;
    }
    PT_HIP(hipMemcpy(h_stamps, h->stamps, sizeof(long long) * 16 * n_records, hipMemcpyDeviceToHost));
    return 0;
}

// third inspection hook: the label image (valid at run starts) and the per-word labels of image img of the last H x W call
extern "C" int ptocr_dbpost_debug_labels(ptocr_dbpost_t h, int img, int H, int W, int32_t *h_labels, int32_t *h_word_labels) {
    PT_CHECK(h && img >= 0 && img < h->max_n && h_labels && h_word_labels, "ptocr_dbpost_debug_labels: bad arguments");
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemcpy(h_labels, h->labels + (long)img * H * W, sizeof(int) * (size_t)H * W, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_word_labels, h->word_lab + (long)img * H * cdiv(W, 32), sizeof(int) * (size_t)H * cdiv(W, 32), hipMemcpyDeviceToHost));
    return 0;
}

// test hook for the hull -> quad hand-off (tests/test_gpu_dbpost.py): plants, for EVERY border slot, a ready word that claims the NEXT
// call's epoch with a candidate count beyond the quad's table (100 > 64).  A quad that meets such a word before the hull role has
// overwritten it must hand the border to the full-size pass (never index by the count); boxes stay bit-exact either way.  (Words of
// OTHER epochs -- every call leaves a thousand behind -- are simply not this call's; memory nobody wrote is zero = no call's epoch.)
extern "C" int ptocr_dbpost_debug_plant(ptocr_dbpost_t h) {
    PT_CHECK(h, "ptocr_dbpost_debug_plant: null workspace");
    PT_HIP(hipDeviceSynchronize());
    const int next = (h->epoch % 0x3fffff) + 1;
    const size_t n = (size_t)h->max_n * MAX_CAND;
    int *words = (int *)malloc(sizeof(int) * n);
    PT_CHECK(words, "ptocr_dbpost_debug_plant: out of host memory");
    for (size_t i = 0; i < n; i++) words[i] = (next << 9) | (100 << 2) | 1;
    const hipError_t e = hipMemcpy(h->list, words, sizeof(int) * n, hipMemcpyHostToDevice);
    free(words);
    PT_HIP(e);
    return 0;
}

// Labelling route of the next calls on this workspace: 0 = chosen from the workspace's last eight calls (default), 1 = text route (LDS
// slabs), 2 = noise route (global union-find, four threads per word, bottom strip first).  Results never depend on it; tests pin it so
// that a case exercises the same kernels whatever ran before.
extern "C" int ptocr_dbpost_set_route(ptocr_dbpost_t h, int route) {
    PT_CHECK(h && route >= 0 && route <= 2, "ptocr_dbpost_set_route: route must be 0 (auto), 1 (text) or 2 (noise)");
    h->route = route;
    return 0;
}

// device time (HIP events on the call's stream) from the first to the last kernel of the last call on this workspace
extern "C" int ptocr_dbpost_last_device_ms(ptocr_dbpost_t h, float *ms) {
    PT_CHECK(h && ms && h->timed, "ptocr_dbpost_last_device_ms: no completed call on this workspace");
    PT_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

// the whole kernel chain for images [i0, i0 + N) of the call, on stream s
static void run_chain(ptocr_dbpost *h, const float *d_maps, const uint8_t *d_bitmap, int i0, int N, int H, int W, float thresh, float box_thresh,
                      float unclip_ratio, int use_padding_resize, int use_dilation, int max_boxes, hipStream_t s) {
    DbpostDims d;
    d.N = N; d.H = H; d.W = W; d.WW = cdiv(W, 32); d.HW = (long)H * W;
    d.nchunks = (int)((d.HW + CHUNK - 1) / CHUNK);
    d.pool_cap = h->pool_cap;
    // views of the workspace that start at image i0
    const long hw = d.HW, rw = (long)H * d.WW;
    if (d_bitmap) d_bitmap += i0 * hw;
    d_maps += i0 * hw;
    unsigned *w_bits = h->bits + i0 * rw, *w_bits2 = h->bits2 + i0 * rw;
    int *w_labels = h->labels + i0 * hw, *w_word_lab = h->word_lab + i0 * rw, *w_chunk = h->chunk_cnt + (long)i0 * d.nchunks;
    int *w_totals = h->totals + i0, *w_flags = h->flags + i0, *w_strip_totals = h->strip_totals + i0, *w_strip_runs = h->strip_runs + i0;
    Cand *w_cands = h->cands + (long)i0 * MAX_CAND;
    Acc *w_acc = h->acc + (long)i0 * MAX_CAND;
    unsigned *w_pool = h->pool + i0 * h->pool_cap;
    // The strip pass pays on noise maps only (an image whose bottom 64 rows hold >= 1000 run starts); on text-like maps its four
    // launches do nothing and cost ~20 us.  So it is launched when one of the previous EIGHT calls on this workspace met such an image (or could
    // not count: caller's bitmap, dilation); the labels are the same either way, a first noise batch just takes the slower route once.
    // (Also tried: the 5-us single-block kernels -- chunk suffix sums, pool offsets, box compaction -- as "last block of the image"
    // tails of their predecessors: the ticket atomics and the lone tail block cost more than the launches saved, 0.504 against 0.490 ms.)
    const bool counted = !d_bitmap && !use_dilation;
    const int strip_y = (H > 2 * STRIP_ROWS && (h->noise_now || !counted)) ? H - STRIP_ROWS : 0;      // small maps: one pass over everything
    d.strip_y = strip_y;
    // run starts in the strip, counted while binarizing: an upper bound of the strip's components.  Below MAX_CAND the strip pass cannot
    // be enough and is left out for that image (its four kernels return at once); a caller's own bitmap or the dilation is not counted
    // (no count: the strip pass always runs).
    const dim3 row_grid(cdiv(W, 1024), H, N);
    // text route on maps whose width is a multiple of 32: the slab kernel thresholds its own rows (no binarize launch)
    static const int global_ccl = getenv("PTOCR_DBPOST_GLOBAL_CCL") && atoi(getenv("PTOCR_DBPOST_GLOBAL_CCL")) == 1;      // experiment: text route on the global union-find
    const bool fused_binarize = counted && !h->noise_now && strip_y == 0 && (W & 31) == 0 && !global_ccl;
    const int count_row = H > 2 * STRIP_ROWS ? H - STRIP_ROWS : 0;
    if (d_bitmap) hipLaunchKernelGGL(pack_u8_kernel, row_grid, dim3(256), 0, s, d_bitmap, w_bits, d);
    else if (!fused_binarize) {
        DbpostDims dc = d;                                       // the count is taken whether or not the strip pass runs this time
        dc.strip_y = count_row;
        if ((W & 31) == 0)
            hipLaunchKernelGGL(binarize_flat_kernel, dim3(cdiv(H * (W >> 2), 512), N), dim3(256), 0, s, d_maps, w_bits, dc, thresh, counted ? w_strip_runs : nullptr);
        else
            hipLaunchKernelGGL(binarize_kernel, row_grid, dim3(256), 0, s, d_maps, w_bits, dc, thresh, counted ? w_strip_runs : nullptr);
    }
    unsigned *bits = w_bits;
    if (use_dilation) {
        hipLaunchKernelGGL(dilate2x2_kernel, dim3(cdiv(d.WW, 256), H, N), dim3(256), 0, s, w_bits, w_bits2, d);
        bits = w_bits2;
    }
    for (int pass = strip_y ? 0 : 1; pass < 2; pass++) {
        CclPass ps;
        ps.dbg = 0;
        ps.y_first = pass == 0 ? strip_y : 0;
        ps.skip_if_full = (pass == 1 && strip_y) ? w_strip_totals : nullptr;
        ps.skip_if_few = (pass == 0 && counted) ? w_strip_runs : nullptr;
        const int words = (H - ps.y_first) * d.WW;
        const dim3 word_grid(cdiv(words > d.nchunks ? words : d.nchunks, 256), N);
        ps.row_step = 0;
        if (h->noise_now || !counted) {                             // noise route (or no count): the global union-find, four threads per word
            hipLaunchKernelGGL(ccl_init_kernel, word_grid, dim3(256), 0, s, bits, w_labels, w_chunk, d, ps);
            hipLaunchKernelGGL(ccl_merge_kernel<4>, dim3(cdiv(4 * words, 256), N), dim3(256), 0, s, bits, w_labels, d, ps);
        } else if (global_ccl) {                                    // rounds 2-3: global union-find, one thread per word
            hipLaunchKernelGGL(ccl_init_kernel, word_grid, dim3(256), 0, s, bits, w_labels, w_chunk, d, ps);
            hipLaunchKernelGGL(ccl_merge_kernel<1>, dim3(cdiv(words, 256), N), dim3(256), 0, s, bits, w_labels, d, ps);
        } else {
            // text route: slabs of eight rows labelled in LDS, then the slab boundary rows linked globally
            const int rows = H - ps.y_first, nslab = cdiv(rows, SLAB_ROWS);
            hipLaunchKernelGGL(ccl_slab_kernel, dim3(nslab, N), dim3(cdiv(SLAB_ROWS * d.WW, 64) * 64), 0, s, bits, w_labels, w_chunk, d, ps,
                               fused_binarize ? d_maps : nullptr, thresh, w_strip_runs, count_row);
            if (nslab > 1) {
                CclPass pb = ps;
                pb.row_step = SLAB_ROWS;
                hipLaunchKernelGGL(ccl_merge_rows_kernel, dim3(cdiv((nslab - 1) * d.WW, 256), N), dim3(256), 0, s, bits, w_labels, d, pb);
            }
        }
        hipLaunchKernelGGL(ccl_flatten_kernel, word_grid, dim3(256), 0, s, bits, w_labels, w_word_lab, w_chunk, h->chunk_roots + (long)i0 * d.nchunks * ROOT_K, d, ps);
        if (strip_y) hipLaunchKernelGGL(chunk_suffix_kernel, dim3(N), dim3(1024), 0, s, w_chunk, w_totals, d, ps, w_strip_totals);
    }
    if (strip_y)         // two labelling passes (bottom strip first): suffix sums per pass, then the selection over whichever pass stands
        hipLaunchKernelGGL(select_starts_kernel, dim3(cdiv(d.nchunks, SEL_CHUNKS), N), dim3(256), 0, s, bits, w_labels, w_chunk, w_totals, w_cands, w_acc, d);
    else                 // one pass: suffix sums and selection in one launch
        hipLaunchKernelGGL(rank_starts_kernel, dim3(N), dim3(1024), 0, s, bits, w_labels, w_chunk, h->chunk_roots + (long)i0 * d.nchunks * ROOT_K, w_totals, w_strip_totals,
                           w_cands, w_acc, d);
    const dim3 all_words(cdiv(cdiv(H, 8) * cdiv(d.WW, 8), 4), N);      // 8 x 8-word tiles, four per block
    StageArgs2 sg;
    sg.rec = h->stage + (long)i0 * h->stage_cap;
    sg.hdr = h->stage_hdr + (long)i0 * (h->stage_cap / STAGE_TILE); sg.cap = h->stage_cap;
    sg.tab = h->stage_tab + (long)i0 * (h->stage_cap / STAGE_TILE) * 2 * WT_SLOTS;
    // ONE enumeration of the border states (count + stage), offsets (+ the score plan), scatter
    hipLaunchKernelGGL(border_states_kernel, all_words, dim3(256), 0, s, bits, w_labels, w_word_lab, w_strip_totals, w_acc, d, sg);
    hipLaunchKernelGGL(pool_offsets_kernel, dim3(N), dim3(1024), 0, s, w_acc, w_totals, w_flags, d, h->sc_off + (long)i0 * MAX_CAND, h->sc_n + i0, h->sc_item + (long)i0 * h->sc_cap, h->sc_cap);
    hipLaunchKernelGGL(scatter_states_kernel, all_words, dim3(256), 0, s, w_strip_totals, w_acc, w_pool, w_flags, d, sg);
    StageArgs a;
    a.maps = d_maps; a.cands = w_cands; a.totals = w_totals; a.acc = w_acc; a.pool = w_pool;
    a.results = h->results + (long)i0 * MAX_CAND; a.flags = w_flags; a.src_wh = h->src_wh + 2 * i0; a.hin = h->hin + (long)i0 * MAX_CAND * S_MH; a.mini = h->mini + (long)i0 * MAX_CAND * 8;
    a.box_thresh = box_thresh; a.unclip_ratio = unclip_ratio; a.use_padding_resize = use_padding_resize;
    a.ready = h->list + (long)i0 * MAX_CAND; a.epoch = h->epoch; a.tie = h->tie + (long)i0 * MAX_CAND;
    a.sc_off = h->sc_off + (long)i0 * MAX_CAND; a.sc_n = h->sc_n + i0; a.sc_item = h->sc_item + (long)i0 * h->sc_cap; a.sc_part = h->sc_part + (long)i0 * h->sc_cap; a.sc_cap = h->sc_cap;
    a.sc_done = h->sc_done + (long)i0 * MAX_CAND;
    a.stamps = h->stamps ? h->stamps + (long)i0 * MAX_CAND * 16 : nullptr;
    // (round 6, measured and dropped: the hull + score roles and the quad role as TWO launches on two streams, so that the first could hold five
    // waves per SIMD beside the quads' 168 registers -- the launches did not overlap usefully, 0.281 against 0.224 ms per call)
    // (round 6, measured and dropped: the three roles fed by per-image ticket queues from a fixed number of resident waves instead of the
    // grid order -- hull items first, then quad groups, then score items: 150 us against 72.8 for the launch; with every role's body inside a
    // ticket loop the allocator spills in all of them, and a ticket is a 0.7-us returning atomic per item, tools/dbg/queue_probe.hip)
    // (... and the score role as a launch of its own on a side stream beside a hull + quad launch: 0.279 against 0.225 ms per call -- the
    // two launches slow each other down more than the grid order costs: 81 + 53 us against 72.7 fused)
    hipLaunchKernelGGL(border_stage_kernel, dim3((unsigned)N * (STAGE_GRID + QUAD_BLOCKS + SCORE_GRID)), dim3(WAVE_NT), 0, s, a, d);
    CompactArgs cp;
    cp.boxes = h->boxes + (long)i0 * max_boxes * 8; cp.counts = h->counts + i0; cp.max_boxes = max_boxes; cp.strip_totals = w_strip_totals; cp.strip_runs = w_strip_runs;
    cp.flags_out = h->flags_out + i0; cp.strip_out = h->strip_out + i0; cp.big_done = h->tickets + i0;
    hipLaunchKernelGGL(contour_big_kernel, dim3(BIG_GRID, N), dim3(BIG_THREADS), 0, s, a, d, cp);      // (+ the compaction: its last workgroup per image)
}

extern "C" int ptocr_db_postprocess(ptocr_dbpost_t h, const float *d_maps, const uint8_t *d_bitmap, int N, int H, int W,
                                    float thresh, float box_thresh, float unclip_ratio, const int *h_src_wh,
                                    int use_padding_resize, int16_t *h_boxes, int max_boxes, int32_t *h_counts,
                                    int32_t *h_flags, void *stream) {
    return ptocr_db_postprocess_ex(h, d_maps, d_bitmap, N, H, W, thresh, box_thresh, unclip_ratio, h_src_wh, use_padding_resize, 0,
                                   h_boxes, max_boxes, h_counts, h_flags, stream);
}

extern "C" int ptocr_db_postprocess_ex(ptocr_dbpost_t h, const float *d_maps, const uint8_t *d_bitmap, int N, int H, int W,
                                       float thresh, float box_thresh, float unclip_ratio, const int *h_src_wh,
                                       int use_padding_resize, int use_dilation, int16_t *h_boxes, int max_boxes,
                                       int32_t *h_counts, int32_t *h_flags, void *stream) {
    PT_CHECK(h && d_maps && h_src_wh && h_boxes && h_counts && h_flags, "ptocr_db_postprocess: null argument");
    PT_CHECK(N >= 1 && N <= h->max_n && H >= 1 && W >= 1 && H <= h->max_h && W <= h->max_w && (long)H * W <= (long)h->max_h * h->max_w,
             "ptocr_db_postprocess: batch %dx%dx%d exceeds the workspace (%dx%dx%d)", N, H, W, h->max_n, h->max_h, h->max_w);
    PT_CHECK(!use_padding_resize || H == W, "ptocr_db_postprocess: use_padding_resize expects the square padded map the reference uses");
    PT_CHECK(max_boxes >= 1 && max_boxes <= MAX_CAND, "ptocr_db_postprocess: max_boxes must be in [1, %d]", MAX_CAND);
    hipStream_t s = (hipStream_t)stream;
    PT_HIP(hipMemcpyAsync(h->src_wh, h_src_wh, sizeof(int) * 2 * N, hipMemcpyHostToDevice, s));
    if (h->dirty) {                                         // (the compaction of a finished call leaves both clear)
        PT_HIP(hipMemsetAsync(h->zeroed, 0, sizeof(int) * 3 * h->max_n, s));
        PT_HIP(hipMemsetAsync(h->tickets, 0, sizeof(int) * 4 * h->max_n, s));
    }
    h->dirty = 1;
    h->noise_now = h->route == 1 ? 0 : (h->route == 2 ? 1 : h->strip_hint);
    h->epoch = (h->epoch % 0x3fffff) + 1;                    // 22 bits: the ready word is epoch << 9 | count << 2 | state
    if (h->epoch == 1 && h->timed) {                        // the epoch wrapped: words and tags of 4 M calls ago must not pass for this call's
        PT_HIP(hipMemsetAsync(h->list, 0, sizeof(int) * h->max_n * MAX_CAND, s));
        PT_HIP(hipMemsetAsync(h->hin, 0, sizeof(F2) * (size_t)h->max_n * MAX_CAND * S_MH, s));
    }
    PT_HIP(hipEventRecord(h->ev0, s));
    // Most kernels of the chain are bound by the latency of ONE image's dependent steps (label chases, per-border geometry), not by
    // the chip: 2 maps take 0.25 ms of kernel time, 32 maps 0.51.  So a batch is cut into up to four parts whose chains run on four
    // streams of the workspace at once (fork / join on events around them): one part's latency-bound kernel fills the CUs another
    // part's leaves idle.  Images are independent: every buffer is indexed by image, a part just starts at its first image.
    static const int want_parts = getenv("PTOCR_DBPOST_PARTS") ? atoi(getenv("PTOCR_DBPOST_PARTS")) : 1;      // measured: 1 part 0.61 ms wall per call, 2 parts 0.62, 4 parts 0.89 (the host issues four times the launches)
    int parts = want_parts < 1 ? 1 : (want_parts > DBPOST_STREAMS ? DBPOST_STREAMS : want_parts);
    if (N < 2 * parts) parts = N >= 4 ? 2 : 1;
    if (parts > 1) PT_HIP(hipEventRecord(h->ev_fork, s));
    // (round 6: the parts' launches issued ROUND-ROBIN over their streams, so that the chains start together instead of a chain's worth of host
    // launch time apart: 2 parts 0.254 ms of device time either way, 4 parts 0.39, against 0.219 for one -- the chains do not overlap usefully)
    for (int p = 0, i0 = 0; p < parts; p++) {
        const int n = N / parts + (p < N % parts ? 1 : 0);
        hipStream_t ps = parts > 1 ? h->sub[p] : s;
        if (parts > 1) PT_HIP(hipStreamWaitEvent(ps, h->ev_fork, 0));
        run_chain(h, d_maps, d_bitmap, i0, n, H, W, thresh, box_thresh, unclip_ratio, use_padding_resize, use_dilation, max_boxes, ps);
        if (parts > 1) {
            PT_HIP(hipEventRecord(h->ev_join[p], ps));
            PT_HIP(hipStreamWaitEvent(s, h->ev_join[p], 0));
        }
        i0 += n;
    }
    if (int e = launch_ok("dbpost kernels")) return e;
    PT_HIP(hipEventRecord(h->ev1, s));
    h->timed = 1;
    PT_HIP(hipMemcpyAsync(h->h_meta, h->flags_out, sizeof(int) * 3 * h->max_n, hipMemcpyDeviceToHost, s));      // flags, strip counts, box counts: one copy (three cost 5 us each on the stream)
    PT_HIP(hipMemcpyAsync(h_boxes, h->boxes, sizeof(short) * 8 * (size_t)N * max_boxes, hipMemcpyDeviceToHost, s));
    PT_HIP(hipStreamSynchronize(s));
    h->dirty = 0;
    for (int i = 0; i < N; i++) {
        h_flags[i] = h->h_meta[i] & 7;                      // bit 3 is internal (deferred borders)
        h->h_strip[i] = h->h_meta[h->max_n + i];
        h_counts[i] = h->h_meta[2 * h->max_n + i];
    }
    if (!d_bitmap && !use_dilation) {
        // the noise route stays on for eight calls after the last noise-like image: a workspace fed text-like and noise-like batches in
        // turn (bench.py's two passes per step did exactly that) would otherwise take the wrong route every time -- one thread per word
        // on a speckle map is 0.8-1.6 ms of merge, and the full-size labelling pass instead of the strip another 0.5 ms
        int noisy = 0;
        // "noise-like": four run starts per candidate slot in the bottom strip (a speckle map holds ~20 000 there, text with ragged edges
        // 1 000 - 3 000, clean text 300).  Only the ROUTE depends on it; an image's strip pass is still skipped on the exact bound
        // (fewer than MAX_CAND run starts cannot be MAX_CAND components), and the labels are the same on either route.
        for (int i = 0; i < N; i++) noisy |= h->h_strip[i] >= 4 * MAX_CAND;
        h->noise_hist = ((h->noise_hist << 1) | (unsigned)noisy) & 0xffu;
        h->strip_hint = h->noise_hist != 0;
    }
    for (int i = 0; i < N; i++)
        if (h_flags[i] & 4) return fail("ptocr_db_postprocess: internal capacity exceeded on image %d (state pool %ld entries or hull "
                                        "candidates %d)", i, h->pool_cap, MAXHULL);
    return 0;
}
