// DB probability-map post-process on MI355X: threshold -> bit-packed bitmap -> run-based union-find CC labelling
// (foreground 8-connected, background 4-connected) -> border following of the 1000 bottom-most borders ->
// per-border min-area box, polygon-mask score, Clipper round-offset unclip, final integer box.
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
//  * Each border is then traced by one thread (the trace only reads the binary image), twice: count, then
//    write the direction-change points (CHAIN_APPROX_SIMPLE) into an exactly-sized slot of a point pool.
//  * One workgroup per border does the rest: column-extreme + Akl-Toussaint pre-filter and Sklansky hull,
//    float32 rotating calipers, fillPoly rasterisation into bit planes (edge lines 4-connected, even-odd fill
//    via per-row prefix-xor of crossing toggles), masked mean in double, unclip and final box.
// Memory: everything is integer/bit work bound by HBM/L2 latency, not by MFMA; the bitmap is 1 bit/pixel so the
// trace works out of L2.
#include "common.h"

namespace ptocr {

constexpr int MAX_CAND = 1000;          // reference db_postprocess.cpp:239 (hard-coded max_candidates)
constexpr int CHUNK = 1024;             // pixels per root-count chunk
constexpr int FRAME = -1;               // label of background connected to the image frame
constexpr int QCAP = 512;               // points of a border's quick slot: borders up to QCAP points are traced once
constexpr long QOFF = (long)MAX_CAND * QCAP;   // start of the exact-size region inside an image's pool

struct DbpostDims {
    int N, H, W, WW;                    // WW = 32-bit words per bitmap row
    long HW;
    int nchunks;                        // chunks per image
    long pool_cap;                      // points per image in the exact-size region of the point pool
    long pool_stride;                   // points per image: quick slots (MAX_CAND x QCAP) + exact-size region
};

// ------------------------------------------------------------------------------------------ binarize
// 4 pixels per lane (one 16-B load), nibbles of 8 lanes OR-combined into a 32-bit word.
__global__ __launch_bounds__(256) void binarize_kernel(const float *__restrict__ maps, unsigned *__restrict__ bits,
                                                       DbpostDims d, float thresh) {
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
struct CclPass { int y_first; const int *skip_if_full; };        // skip_if_full: per-image start counts of pass A, or null

__device__ __forceinline__ bool ccl_skip(const CclPass &ps, int img) { return ps.skip_if_full && ps.skip_if_full[img] >= MAX_CAND; }

// label[s] = s for every run start s (pass B also clears the image's chunk counters)
__global__ __launch_bounds__(256) void ccl_init_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                       int *__restrict__ chunk_cnt, DbpostDims d, CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (ps.skip_if_full && idx < d.nchunks) chunk_cnt[(long)img * d.nchunks + idx] = 0;
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
__global__ __launch_bounds__(256) void ccl_merge_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels, DbpostDims d,
                                                        CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int idx = (blockIdx.x * 256 + threadIdx.x) >> 2;       // four threads per word, one byte of its boundary masks each:
    const unsigned part = 0xffu << (8 * (threadIdx.x & 3));     // speckle maps put ~10 unions into a word, each a chain of atomics
    if (idx >= (d.H - ps.y_first) * d.WW) return;
    const int y = ps.y_first + idx / d.WW, wi = idx % d.WW;
    const bool cut = y == ps.y_first && y > 0;                    // first row of the strip: links upwards end in FRAME
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
            uf_union(lab, rs_cur(i), FRAME);
        }
    }
    if (y == 0) return;
    const unsigned *up = row - d.WW;
    const WordCtx u = word_ctx(up, wi, d);
    const int ubase = base - d.W;
    auto rs_up = [&](int x) { return (y - 1) * d.W + run_start(up, x); };
    // vertical links: same class above, and the current or the upper run starts here
    unsigned v = ~(c.w ^ u.w) & c.valid & (c.starts | u.starts) & part;
    while (v) {
        const int i = __ffs(v) - 1;
        v &= v - 1;
        uf_union(lab, rs_cur(i), cut ? FRAME : ((u.starts >> i) & 1u ? ubase + i : rs_up(x0 + i)));
    }
    // foreground with background straight above: diagonal links
    const unsigned fgbg = c.w & ~u.w & c.valid;
    const unsigned ucarry = wi ? up[wi - 1] >> 31 : 0u;
    unsigned nw = fgbg & c.starts & ((u.w << 1) | ucarry) & part;         // pix(up, x-1) set; x = 0 has no NW neighbour (carry 0)
    while (nw) {
        const int i = __ffs(nw) - 1;
        nw &= nw - 1;
        uf_union(lab, base + i, cut ? FRAME : rs_up(x0 + i - 1));
    }
    const unsigned unext = (wi + 1 < d.WW) ? (up[wi + 1] & 1u) : 0u;
    unsigned ne = fgbg & c.ends & ((u.w >> 1) | (unext << 31)) & part;     // pix(up, x+1) set (bits beyond W are clear)
    while (ne) {
        const int i = __ffs(ne) - 1;
        ne &= ne - 1;
        uf_union(lab, rs_cur(i), cut ? FRAME : ubase + i + 1);      // up(x) = 0, up(x+1) = 1: a run start
    }
}

// label[s] = root for every run start s; counts border starts (roots) per 1024-pixel chunk (chunk_cnt zeroed by the host)
__global__ __launch_bounds__(256) void ccl_flatten_kernel(const unsigned *__restrict__ bits, int *__restrict__ labels,
                                                          int *__restrict__ chunk_cnt, DbpostDims d, CclPass ps) {
    const int img = blockIdx.y;
    if (ccl_skip(ps, img)) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (d.H - ps.y_first) * d.WW) return;
    const int y = ps.y_first + idx / d.WW, wi = idx % d.WW;
    const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
    int *lab = labels + (long)img * d.HW;
    unsigned m = word_ctx(row, wi, d).starts;
    const int base = y * d.W + wi * 32;
    while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        const int s = base + i;
        const int r = uf_find(lab, s);
        lab[s] = r;
        if (r == s) atomicAdd(&chunk_cnt[(long)img * d.nchunks + s / CHUNK], 1);
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

__global__ __launch_bounds__(256) void gather_starts_kernel(const unsigned *__restrict__ bits, const int *__restrict__ labels,
                                                            const int *__restrict__ chunk_after, Cand *__restrict__ cands,
                                                            DbpostDims d) {
    const int img = blockIdx.y, chunk = blockIdx.x;
    const int after = chunk_after[(long)img * d.nchunks + chunk];
    if (after >= MAX_CAND) return;                              // every start here ranks beyond the first 1000
    const int *lab = labels + (long)img * d.HW;
    __shared__ int wave_cnt[4];
    __shared__ int base;                                        // starts already ranked in this chunk (from the top index down)
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = CHUNK / 256 - 1; k >= 0; k--) {               // highest pixels first
        const long p = (long)chunk * CHUNK + k * 256 + (255 - threadIdx.x);   // thread 0 takes the highest pixel
        bool is = false;
        if (p < d.HW) {                                         // only run starts carry labels
            const int y = (int)(p / d.W), x = (int)(p - (long)y * d.W);
            const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
            is = (x == 0 || pix(row, x) != pix(row, x - 1)) && lab[p] == (int)p;
        }
        const unsigned long long m = __ballot(is);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; w++) before += wave_cnt[w];
        before += __popcll(m & ((1ull << lane) - 1));
        if (is) {
            const int rank = after + before;
            if (rank < MAX_CAND) {
                const int y = (int)(p / d.W), x = (int)(p - (long)y * d.W);
                const unsigned *row = bits + ((long)img * d.H + y) * d.WW;
                Cand c; c.p = (int)p; c.is_hole = !pix(row, x);
                cands[(long)img * MAX_CAND + rank] = c;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ border following
// direction k (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE) -> step, from 2-bit fields (value + 1) so that no table load sits in the walk
__device__ __forceinline__ int dir_dx(int k) { return (int)((0x901Au >> (2 * k)) & 3u) - 1; }    // {1,1,0,-1,-1,-1,0,1}
__device__ __forceinline__ int dir_dy(int k) { return (int)((0xA901u >> (2 * k)) & 3u) - 1; }    // {0,-1,-1,-1,0,1,1,1}

struct TraceOut { int npts, xmin, xmax, ymin, ymax; };

// One WAVE per border.  The walk itself is sequential, so all 64 lanes execute it redundantly on wave-uniform state while
// the bitmap around the current pixel sits in registers: an 8-row x 8-word (256-pixel) window, lane l = word (l & 7) of
// row (l >> 3), fetched with one coalesced load and re-centred when the walk leaves it.  A step then costs a few readlanes
// instead of six dependent global loads per lane, and the kernel holds no LDS (it has to co-reside with the convolution
// workgroups of the next batch, which take almost a whole CU's LDS).  What remains is the dependent instruction chain of
// the walk itself (~1200 cycles per step for a lone wave), i.e. latency, not bandwidth.
struct BitWindow {
    const unsigned *bits; int H, W, WW;
    int y0, wx0;                 // window origin: row, word column (may lie outside the image: those words read 0)
    unsigned word;               // this lane's word
    __device__ __forceinline__ void load(int x, int y, int lane) {
        y0 = y - 3;
        wx0 = (x >> 5) - 4 + ((x & 31) >= 16);
        const int r = y0 + (lane >> 3), wi = wx0 + (lane & 7);
        word = ((unsigned)r < (unsigned)H && (unsigned)wi < (unsigned)WW) ? bits[(long)r * WW + wi] : 0u;
    }
    __device__ __forceinline__ bool inside(int x, int y) const {
        return y >= y0 + 1 && y <= y0 + 6 && x >= wx0 * 32 + 1 && x <= wx0 * 32 + 254;
    }
    // 3 pixels (x-1, x, x+1) of window row rr as bits 0..2
    __device__ __forceinline__ unsigned row3(int rr, int x) const {
        const int xs = x - 1 - wx0 * 32;                        // 0 .. 253
        const int l = __builtin_amdgcn_readfirstlane(rr * 8 + (xs >> 5));
        const unsigned lo = __builtin_amdgcn_readlane(word, l);
        const unsigned hi = __builtin_amdgcn_readlane(word, (l + 1) & 63);   // next row's word only when xs & 31 <= 29: unused bits
        const unsigned long long c = lo | ((unsigned long long)hi << 32);
        return (unsigned)(c >> (xs & 31)) & 7u;
    }
    // 8-neighbour mask, bit k = neighbour in direction k (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE); pixels outside the image = 0
    __device__ __forceinline__ unsigned nbr8(int x, int y, int lane) {
        if (!inside(x, y)) load(x, y, lane);
        const int rr = y - y0;
        const unsigned a = row3(rr - 1, x), b = row3(rr, x), c = row3(rr + 1, x);
        return ((b >> 2) & 1) | (((a >> 2) & 1) << 1) | (((a >> 1) & 1) << 2) | ((a & 1) << 3) |
               ((b & 1) << 4) | ((c & 1) << 5) | (((c >> 1) & 1) << 6) | (((c >> 2) & 1) << 7);
    }
};

// Suzuki-Abe border following from (sx, sy) with the direction-change points of CHAIN_APPROX_SIMPLE, executed by a whole
// wave on uniform state.  WRITE: points are stored as (x | y << 16), 64 at a time (lane i keeps point i of the batch).
template <bool WRITE>
__device__ TraceOut trace_border(BitWindow &im, int sx, int sy, int is_hole, unsigned *out, int cap, int lane) {
    TraceOut t; t.npts = 0; t.xmin = t.xmax = sx; t.ymin = t.ymax = sy;
    im.load(sx, sy, lane);
    unsigned nb = im.nbr8(sx, sy, lane);
    // first neighbour clockwise from W (outer) or from E (hole)
    int s = is_hole ? 0 : 4, s_end = s;
    do { s = (s - 1) & 7; } while (!((nb >> s) & 1) && s != s_end);
    if (s == s_end) {                            // single pixel domain
        if (WRITE && lane == 0) out[0] = (unsigned)sx | ((unsigned)sy << 16);
        t.npts = 1;
        return t;
    }
    const int i1x = sx + dir_dx(s), i1y = sy + dir_dy(s);
    int prev_s = s ^ 4;
    int x = sx, y = sy;
    unsigned mine = 0;                           // the batch point held by this lane
    // every wave must terminate: a border has fewer steps than 4 per pixel; a longer walk means a broken start
    for (long guard = 4L * im.H * im.W + 16; guard > 0; guard--) {
        // counter-clockwise search for the next border pixel starting after direction s
        const unsigned rot = ((nb | (nb << 8)) >> (s + 1)) & 0xffu;     // bit j = direction s+1+j
        if (rot == 0) break;                                              // isolated pixel: cannot happen after a valid start
        const int j = __ffs(rot) - 1;
        s = __builtin_amdgcn_readfirstlane((s + 1 + j) & 7);   // wave-uniform: keep the walk on the scalar unit
        if (s != prev_s) {
            if (WRITE && t.npts < cap) {                        // cap is a multiple of 64: whole batches only
                if (lane == (t.npts & 63)) mine = (unsigned)x | ((unsigned)y << 16);
                if ((t.npts & 63) == 63) out[(t.npts & ~63) + lane] = mine;
            }
            t.npts++;
            prev_s = s;
            t.xmin = min(t.xmin, x); t.xmax = max(t.xmax, x); t.ymin = min(t.ymin, y); t.ymax = max(t.ymax, y);
        }
        const int nx = x + dir_dx(s), ny = y + dir_dy(s);
        if (nx == sx && ny == sy && x == i1x && y == i1y) break;
        x = __builtin_amdgcn_readfirstlane(nx); y = __builtin_amdgcn_readfirstlane(ny);
        s = (s + 4) & 7;
        nb = im.nbr8(x, y, lane);
    }
    if (WRITE && t.npts < cap && lane < (t.npts & 63)) out[(t.npts & ~63) + lane] = mine;      // the last partial batch
    return t;
}

struct CandInfo {          // per candidate, filled by the count pass
    int npts, off;         // number of approx points, offset into the image's point pool
    short xmin, xmax, ymin, ymax;
};

// first pass: counts the points of every border and stores them in the border's quick slot as long as they fit (<= QCAP:
// every text-like border); only longer borders are traced a second time into an exact-size slot
__global__ __launch_bounds__(64) void trace_count_kernel(const unsigned *__restrict__ bits, const Cand *__restrict__ cands,
                                                         const int *__restrict__ totals, CandInfo *__restrict__ info,
                                                         unsigned *__restrict__ pool, DbpostDims d) {
    const int img = blockIdx.y, k = blockIdx.x;
    const int lane = threadIdx.x;
    const int num = min(totals[img], MAX_CAND);
    if (k >= num) return;
    const Cand c = cands[(long)img * MAX_CAND + k];
    BitWindow im; im.bits = bits + (long)img * d.H * d.WW; im.H = d.H; im.W = d.W; im.WW = d.WW;
    const int p = __builtin_amdgcn_readfirstlane(c.p), hole = __builtin_amdgcn_readfirstlane(c.is_hole);
    const int y = p / d.W, x = p - y * d.W;
    const TraceOut t = trace_border<true>(im, x - hole, y, hole, pool + (long)img * d.pool_stride + (long)k * QCAP, QCAP, lane);
    if (lane == 0) {
        CandInfo ci; ci.npts = t.npts; ci.off = k * QCAP;
        ci.xmin = (short)t.xmin; ci.xmax = (short)t.xmax; ci.ymin = (short)t.ymin; ci.ymax = (short)t.ymax;
        info[(long)img * MAX_CAND + k] = ci;
    }
}

// exclusive scan of npts over the candidates of one image (borders with <= 2 points are dropped by the
// reference, db_postprocess.cpp:255, and get no pool space)
__global__ __launch_bounds__(1024) void pool_offsets_kernel(CandInfo *__restrict__ info, const int *__restrict__ totals,
                                                            int *__restrict__ flags, DbpostDims d) {
    const int img = blockIdx.x, k = threadIdx.x;
    __shared__ int sh[1024];
    const int num = min(totals[img], MAX_CAND);
    int n = 0;
    if (k < num) { n = info[(long)img * MAX_CAND + k].npts; if (n <= QCAP) n = 0; }   // short borders sit in their quick slots
    sh[k] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = k >= off ? sh[k - off] : 0;
        __syncthreads();
        sh[k] += v;
        __syncthreads();
    }
    if (k < num && n) info[(long)img * MAX_CAND + k].off = (int)QOFF + sh[k] - n;
    if (k == 1023 && (long)sh[1023] > d.pool_cap) atomicOr(&flags[img], 4);
}

__global__ __launch_bounds__(64) void trace_write_kernel(const unsigned *__restrict__ bits, const Cand *__restrict__ cands,
                                                         const int *__restrict__ totals, const CandInfo *__restrict__ info,
                                                         unsigned *__restrict__ pool, const int *__restrict__ flags, DbpostDims d) {
    const int img = blockIdx.y, k = blockIdx.x;
    const int lane = threadIdx.x;
    const int num = min(totals[img], MAX_CAND);
    if (k >= num || (flags[img] & 4)) return;
    const CandInfo ci = info[(long)img * MAX_CAND + k];
    if (ci.npts <= QCAP) return;
    const Cand c = cands[(long)img * MAX_CAND + k];
    BitWindow im; im.bits = bits + (long)img * d.H * d.WW; im.H = d.H; im.W = d.W; im.WW = d.WW;
    const int p = __builtin_amdgcn_readfirstlane(c.p), hole = __builtin_amdgcn_readfirstlane(c.is_hole);
    const int off = __builtin_amdgcn_readfirstlane(ci.off);
    const int y = p / d.W, x = p - y * d.W;
    trace_border<true>(im, x - hole, y, hole, pool + (long)img * d.pool_stride + off, 0x7fffffc0, lane);
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

// float32 rotating calipers (restates OpenCV rotcalipers.cpp, CALIPERS_MINAREARECT); scratch: 3*n floats
__device__ void rotating_calipers(const F2 *points, int n, float *scratch, float *out) {
    float minarea = 3.402823466e+38f;
    float *inv_len = scratch;
    F2 *vect = reinterpret_cast<F2 *>(scratch + n);
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4];
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

// db_postprocess.cpp:159-192 (std::sort of 4 elements = insertion sort => stable on equal x)
__device__ void get_mini_boxes(const RRect &box, float out[4][2], float *ssid) {
    F2 p[4];
    *ssid = box.w > box.h ? box.w : box.h;
    box_points(box, p);
    for (int i = 1; i < 4; i++) {
        const F2 t = p[i];
        int j = i;
        for (; j > 0 && t.x < p[j - 1].x; j--) p[j] = p[j - 1];
        p[j] = t;
    }
    F2 i1, i2, i3, i4;
    if (p[3].y <= p[2].y) { i2 = p[3]; i3 = p[2]; } else { i2 = p[2]; i3 = p[3]; }
    if (p[1].y <= p[0].y) { i1 = p[1]; i4 = p[0]; } else { i1 = p[0]; i4 = p[1]; }
    out[0][0] = i1.x; out[0][1] = i1.y; out[1][0] = i2.x; out[1][1] = i2.y;
    out[2][0] = i3.x; out[2][1] = i3.y; out[3][0] = i4.x; out[3][1] = i4.y;
}

// Clipper 6.4.2 round offset of one closed path (restates clipper.cpp:3837-3879, 3889-3913, 3987-4020, 4160-4244);
// the union clean-up that follows in ClipperOffset::Execute does not change the hull of the result for
// delta >= 0.75 (checked against the vendored Clipper in tests/test_oracle_clipper.py).
struct CPt { long long X, Y; };
__device__ __forceinline__ long long cl_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

__device__ int clipper_offset_round(const CPt *path4, double delta, F2 *out, int cap) {
    const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc = 0.25, arc_tol = 0.25;
    CPt src[4]; double nx[4], ny[4];
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

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }

// ------------------------------------------------------------------------------------------ per-border workgroup
enum { ST_OK = 0, ST_SKIP_NPTS = 1, ST_SKIP_SSID = 2, ST_SKIP_SCORE = 3, ST_SKIP_UNCLIP = 4, ST_SKIP_SSID2 = 5, ST_NONE = 6,
       ST_DEFER = 7 };   // ST_DEFER: left by the small-footprint pass for the full-size one (never visible after a call)

struct Result { int status; int box[8]; float score; float rect[5]; int npix; float distance; };

constexpr int CT_THREADS = 256;
constexpr int MAXW = 2048;                // widest map the column tables hold
constexpr int LDS_PLANE_WORDS = 4096;     // mask planes up to 131072 pixels live in LDS; larger ones in a global slot
constexpr int MAXHULL = 512;              // strict hull vertices of a lattice polygon inside 2048 x 32767 stay far below
constexpr int NSLOTS = 256;               // global mask slots (two full-image bit planes each)

__device__ __forceinline__ long long cross3(int ax, int ay, int bx, int by, int px, int py) {
    return (long long)(bx - ax) * (py - ay) - (long long)(by - ay) * (px - ax);
}

template <typename T>
__device__ T block_reduce_sum(T v, T *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = CT_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = sh[tid] + sh[tid + s];
        __syncthreads();
    }
    const T r = sh[0];
    __syncthreads();
    return r;
}

// Plane words may live in LDS or in a global slot.  Global words are written with atomics (performed at L2) by
// every wave of the workgroup, so they are read back with agent-scope loads that bypass this CU's L1.
template <bool GLOBAL>
__device__ __forceinline__ unsigned plane_ld(const unsigned *p) {
    if (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool GLOBAL>
__device__ __forceinline__ void plane_st(unsigned *p, unsigned v) {
    if (GLOBAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// fillPoly(mask, {border polygon}, 1, lineType=1) over the bounding box + cv::mean(pred, mask):
// returns sum (double, block tree order) and pixel count; leaves the final mask in `border`.
template <bool GLOBAL>
__device__ void score_mask(const unsigned *pts, int n, int xmin, int ymin, int bw, int bh, unsigned *border, unsigned *toggle,
                           const float *pimg, int W, double *red_d, int *red_i, double *sum_out, int *cnt_out) {
    const int tid = threadIdx.x;
    const int pw = (bw + 31) >> 5;
    const long plane_words = (long)pw * bh;
    for (long i = tid; i < plane_words; i += CT_THREADS) { plane_st<GLOBAL>(&border[i], 0); plane_st<GLOBAL>(&toggle[i], 0); }
    __syncthreads();
    for (int i = tid; i < n; i += CT_THREADS) {
        const unsigned p0 = pts[i == 0 ? n - 1 : i - 1], p1 = pts[i];
        int x0 = (int)(p0 & 0xffff) - xmin, y0 = (int)(p0 >> 16) - ymin;
        int x1 = (int)(p1 & 0xffff) - xmin, y1 = (int)(p1 >> 16) - ymin;
        // even-odd crossings of the non-horizontal edge on rows ya <= y < yb (FillEdgeCollection's half-open rule)
        if (y0 != y1) {
            int ya, yb, xa, xb;
            if (y0 < y1) { ya = y0; yb = y1; xa = x0; xb = x1; } else { ya = y1; yb = y0; xa = x1; xb = x0; }
            const int dxs = (xb - xa) / (yb - ya);              // -1, 0, +1: border edges run in the 8 chain directions
            for (int y = ya, x = xa; y < yb; y++, x += dxs) atomicXor(&toggle[(long)y * pw + (x >> 5)], 1u << (x & 31));
        }
        // the edge itself, 4-connected (cv::Line: connectivity 1 -> 4), walked from its left end point
        if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
        const int dx = x1 - x0, dy = y1 - y0;
        if (dy == 0) {
            for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
                const int lo = max(x0, w * 32) & 31, hi = min(x1, w * 32 + 31) & 31;
                const unsigned m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1)) & ~((1u << lo) - 1);
                atomicOr(&border[(long)y0 * pw + w], m);
            }
        } else if (dx == 0) {
            const int ya = min(y0, y1), yb = max(y0, y1);
            for (int y = ya; y <= yb; y++) atomicOr(&border[(long)y * pw + (x0 >> 5)], 1u << (x0 & 31));
        } else {
            const int sy = dy > 0 ? 1 : -1;
            int x = x0, y = y0;
            atomicOr(&border[(long)y * pw + (x >> 5)], 1u << (x & 31));
            for (int s = 0; s < dx; s++) {                      // x step first, then y step
                x++;
                atomicOr(&border[(long)y * pw + (x >> 5)], 1u << (x & 31));
                y += sy;
                atomicOr(&border[(long)y * pw + (x >> 5)], 1u << (x & 31));
            }
        }
    }
    if (GLOBAL) __threadfence();
    __syncthreads();
    // mask row = prefix_xor(crossings) | border   (a span [c1, c2] of the reference = parity bits c1..c2-1 plus border pixel c2)
    for (int y = tid; y < bh; y += CT_THREADS) {
        unsigned carry = 0;
        for (int w = 0; w < pw; w++) {
            unsigned t = plane_ld<GLOBAL>(&toggle[(long)y * pw + w]);
            t ^= t << 1; t ^= t << 2; t ^= t << 4; t ^= t << 8; t ^= t << 16;
            t ^= carry;
            carry = (t >> 31) ? 0xffffffffu : 0u;
            plane_st<GLOBAL>(&border[(long)y * pw + w], plane_ld<GLOBAL>(&border[(long)y * pw + w]) | t);
        }
    }
    if (GLOBAL) __threadfence();
    __syncthreads();
    double s = 0; int cnt = 0;
    for (long i = tid; i < plane_words; i += CT_THREADS) {
        unsigned m = plane_ld<GLOBAL>(&border[i]);
        if (!m) continue;
        const int y = (int)(i / pw), w = (int)(i - (long)y * pw);
        const float *prow = pimg + (long)(y + ymin) * W + xmin + w * 32;
        cnt += __popc(m);
        while (m) { const int b = __ffs(m) - 1; m &= m - 1; s += (double)prow[b]; }
    }
    *sum_out = block_reduce_sum<double>(s, red_d);
    *cnt_out = block_reduce_sum<int>(cnt, red_i);
}

// same mask, summed by ONE lane in raster order: the order of cv::mean (only when the decision is within rounding)
template <bool GLOBAL>
__device__ double score_mask_raster_order(const unsigned *border, int xmin, int ymin, int bw, int bh, const float *pimg, int W) {
    const int pw = (bw + 31) >> 5;
    double s = 0;
    for (int y = 0; y < bh; y++)
        for (int w = 0; w < pw; w++) {
            unsigned m = plane_ld<GLOBAL>(&border[(long)y * pw + w]);
            const float *prow = pimg + (long)(y + ymin) * W + xmin + w * 32;
            while (m) { const int b = __ffs(m) - 1; m &= m - 1; s += (double)prow[b]; }
        }
    return s;
}

// Two instantiations share this body (contour_kernel / contour_big_kernel below).  SMALL: column tables for borders up to
// 512 px wide, mask planes up to 30720 px and 80 hull candidates -- 17.5 KB of LDS and <= 80 VGPRs, so a dozen workgroups fit a CU and one fits NEXT TO a Winograd workgroup of
// the following batch's forward pass (which leaves 18 KB of LDS and 92 VGPRs per SIMD); a border that exceeds any of the
// limits is marked ST_DEFER.  The full-size instantiation (2048 px, 131072 px, 512 candidates; 53 KB) then handles only those.
template <int MW, int PLANE, int MH, bool SMALL>
__device__ void contour_body(int img, int k, const float *__restrict__ maps, const Cand *__restrict__ cands,
                             const CandInfo *__restrict__ info, const unsigned *__restrict__ pool, unsigned *__restrict__ gslots,
                             int *__restrict__ slot_locks, Result *__restrict__ results, int *__restrict__ flags,
                             const int *__restrict__ src_wh, float box_thresh, float unclip_ratio, long slot_words,
                             int use_padding_resize, const DbpostDims &d) {
    const int tid = threadIdx.x;
    Result *res = &results[(long)img * MAX_CAND + k];
    const CandInfo ci = info[(long)img * MAX_CAND + k];
    if (flags[img] & 4) { if (tid == 0) res->status = ST_NONE; return; }
    if (ci.npts <= 2) { if (tid == 0) res->status = ST_SKIP_NPTS; return; }     // db_postprocess.cpp:255
    const unsigned *pts = pool + (long)img * d.pool_stride + ci.off;
    const int n = ci.npts;
    const int xmin = ci.xmin, xmax = ci.xmax, ymin = ci.ymin, ymax = ci.ymax;
    const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
    // Speckle: both sides of a min-area rectangle are projections of the point set, so neither exceeds its diameter, which is
    // at most the diagonal of the bounding box; a diagonal <= sqrt(8) means ssid < 3 (db_postprocess.cpp:265) without
    // computing the rectangle.  Noise maps are made of thousands of such borders.
    if ((bw - 1) * (bw - 1) + (bh - 1) * (bh - 1) <= 8) { if (tid == 0) res->status = ST_SKIP_SSID; return; }
    if (SMALL && (bw > MW || (long)((bw + 31) >> 5) * bh > PLANE)) { if (tid == 0) { res->status = ST_DEFER; atomicOr(&flags[img], 8); } return; }

    // LDS arena: column tables (hull phase) and mask planes (score phase) are never live together
    constexpr int ARENA = 2 * PLANE > 3 * MW ? 2 * PLANE : 3 * MW;
    __shared__ __attribute__((aligned(16))) unsigned arena[ARENA];
    int *col_lo = reinterpret_cast<int *>(arena);               // [MW] min y of the border points per column
    int *col_hi = col_lo + MW;                                  // [MW] max y
    __shared__ F2 cand_pts[MH];
    __shared__ F2 hull_pts[MH];
    __shared__ int stack[2 * (MH + 2)];
    __shared__ float cal_scratch[3 * MH];
    __shared__ double red_d[CT_THREADS];
    __shared__ int red_i[CT_THREADS];
    __shared__ int wave_cnt[CT_THREADS / 64];
    __shared__ int sh_n, sh_status;
    __shared__ float sh_mini[4][2];

    // ---- 1. per-column extremes of the border points (every strict hull vertex is a column extreme)
    for (int i = tid; i < bw; i += CT_THREADS) { col_lo[i] = 0x7fffffff; col_hi[i] = -0x7fffffff; }
    if (tid == 0) sh_n = 0;
    __syncthreads();
    for (int i = tid; i < n; i += CT_THREADS) {
        const unsigned p = pts[i];
        const int x = (int)(p & 0xffff) - xmin, y = (int)(p >> 16);
        atomicMin(&col_lo[x], y);
        atomicMax(&col_hi[x], y);
    }
    __syncthreads();
    // ---- 1b. compact the non-empty columns in place (a CHAIN_APPROX_SIMPLE border touches few columns: 8 of 169 is
    //          typical for a clean text blob), so that the quadratic filter below runs over m columns, not bw
    int *col_x = col_hi + MW;                                   // [MW] x (relative to xmin) of compacted column a
    const int lane = tid & 63, wave = tid >> 6;
    int m_cols = 0;
    {
        constexpr int ROUNDS = MW / CT_THREADS;                 // bw <= MW
        int v_lo[ROUNDS], v_hi[ROUNDS], v_pos[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            const int i = r * CT_THREADS + tid;
            const bool has = i < bw && col_lo[i] != 0x7fffffff;
            v_lo[r] = has ? col_lo[i] : 0; v_hi[r] = has ? col_hi[i] : 0;
            const unsigned long long bal = __ballot(has);
            if (lane == 0) wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int pos = m_cols + __popcll(bal & ((1ull << lane) - 1));
            for (int w = 0; w < wave; w++) pos += wave_cnt[w];
            v_pos[r] = has ? pos : -1;
            for (int w = 0; w < CT_THREADS / 64; w++) m_cols += wave_cnt[w];
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++)
            if (v_pos[r] >= 0) { col_lo[v_pos[r]] = v_lo[r]; col_hi[v_pos[r]] = v_hi[r]; col_x[v_pos[r]] = r * CT_THREADS + tid; }
        __syncthreads();
    }
    // ---- 2. keep a column extreme only if it is a strict vertex of its chain: for the min-y chain, point i survives
    //         iff it lies strictly on the outer side of every chord (j, k), j < i < k.  It is enough to test the chord
    //         through the steepest predecessor and the steepest successor.  Exact integer arithmetic, O(m^2 / 256).
    //         End columns always survive.  Survivors are emitted in (x, y) order: the order cv::convexHull sorts to.
    for (int base = 0; base < m_cols; base += CT_THREADS) {
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
        int pos = sh_n + incl - mine;
        for (int w = 0; w < wave; w++) pos += wave_cnt[w];
        if (keep_lo) { if (pos < MH) { cand_pts[pos].x = (float)(xi + xmin); cand_pts[pos].y = (float)ylo; } pos++; }
        if (keep_hi) { if (pos < MH) { cand_pts[pos].x = (float)(xi + xmin); cand_pts[pos].y = (float)yhi; } pos++; }
        __syncthreads();
        if (tid == CT_THREADS - 1) sh_n = pos;
        __syncthreads();
    }
    // ---- 3. min-area rect of the border, mini-box, first size filter (lane 0; a few dozen vertices)
    if (tid == 0) {
        int status = ST_OK;
        if (sh_n > MH) { if (SMALL) { status = ST_DEFER; atomicOr(&flags[img], 8); } else { atomicOr(&flags[img], 4); status = ST_NONE; } }
        else {
            const RRect box = min_area_rect_sorted(cand_pts, sh_n, hull_pts, stack, cal_scratch);
            float ssid;
            get_mini_boxes(box, sh_mini, &ssid);
            res->rect[0] = box.cx; res->rect[1] = box.cy; res->rect[2] = box.w; res->rect[3] = box.h; res->rect[4] = box.angle;
            if (ssid < 3) status = ST_SKIP_SSID;                // min_size, db_postprocess.cpp:265
        }
        sh_status = status;
    }
    __syncthreads();
    if (sh_status != ST_OK) { if (tid == 0) res->status = sh_status; return; }

    // ---- 4. BoxScore (db_postprocess.cpp:194-229)
    const float *pimg = maps + (long)img * d.HW;
    const long plane_words = (long)((bw + 31) >> 5) * bh;
    double total; int npix;
    float score;
    if (plane_words <= PLANE) {
        unsigned *border = arena, *toggle = arena + PLANE;
        score_mask<false>(pts, n, xmin, ymin, bw, bh, border, toggle, pimg, d.W, red_d, red_i, &total, &npix);
        score = (float)(npix ? total / npix : 0.0);
        if (fabs((double)score - (double)box_thresh) <= 1e-6) {
            if (tid == 0) { red_d[0] = score_mask_raster_order<false>(border, xmin, ymin, bw, bh, pimg, d.W); atomicOr(&flags[img], 2); }
            __syncthreads();
            score = (float)(npix ? red_d[0] / npix : 0.0);
        }
    } else {
        const int slot = (int)(((long)img * MAX_CAND + k) % NSLOTS);
        unsigned *border = gslots + (long)slot * 2 * slot_words, *toggle = border + slot_words;
        if (tid == 0) { while (atomicCAS(&slot_locks[slot], 0, 1) != 0) __builtin_amdgcn_s_sleep(32); __threadfence(); }
        __syncthreads();
        score_mask<true>(pts, n, xmin, ymin, bw, bh, border, toggle, pimg, d.W, red_d, red_i, &total, &npix);
        score = (float)(npix ? total / npix : 0.0);
        if (fabs((double)score - (double)box_thresh) <= 1e-6) {
            if (tid == 0) { red_d[0] = score_mask_raster_order<true>(border, xmin, ymin, bw, bh, pimg, d.W); atomicOr(&flags[img], 2); }
            __syncthreads();
            score = (float)(npix ? red_d[0] / npix : 0.0);
        }
        __syncthreads();
        if (tid == 0) { __threadfence(); atomicExch(&slot_locks[slot], 0); }
    }
    if (tid != 0) return;
    res->score = score; res->npix = npix;
    if (score < box_thresh) { res->status = ST_SKIP_SCORE; return; }           // db_postprocess.cpp:272

    // ---- 5. UnClip (db_postprocess.cpp:16-64) and the final box (:283-311), lane 0
    float area = 0.0f, dist = 0.0f;
    for (int i = 0; i < 4; i++) {
        const int nn = (i + 1) % 4;
        area += sh_mini[i][0] * sh_mini[nn][1] - sh_mini[i][1] * sh_mini[nn][0];
        dist += sqrtf((sh_mini[i][0] - sh_mini[nn][0]) * (sh_mini[i][0] - sh_mini[nn][0]) +
                      (sh_mini[i][1] - sh_mini[nn][1]) * (sh_mini[i][1] - sh_mini[nn][1]));
    }
    area = (float)fabs((double)(float)(area / 2.0));
    const float distance = area * unclip_ratio / dist;
    res->distance = distance;
    if (distance < 0.75f) atomicOr(&flags[img], 1);            // sub-pixel sliver: Clipper's union clean-up not reproduced
    CPt path[4];
    for (int i = 0; i < 4; i++) { path[i].X = (long long)(int)sh_mini[i][0]; path[i].Y = (long long)(int)sh_mini[i][1]; }
    const int np = clipper_offset_round(path, (double)distance, cand_pts, MH);
    RRect ub;
    if (np > MH) { if (SMALL) { res->status = ST_DEFER; atomicOr(&flags[img], 8); } else { atomicOr(&flags[img], 4); res->status = ST_NONE; } return; }
    if (np <= 0) { ub.cx = 0; ub.cy = 0; ub.w = 1; ub.h = 1; ub.angle = 0; }
    else {
        for (int i = 1; i < np; i++) {                          // sort by (x, y) like cv::convexHull
            const F2 t = cand_pts[i];
            int j = i;
            for (; j > 0 && (t.x < cand_pts[j - 1].x || (t.x == cand_pts[j - 1].x && t.y < cand_pts[j - 1].y)); j--) cand_pts[j] = cand_pts[j - 1];
            cand_pts[j] = t;
        }
        ub = min_area_rect_sorted(cand_pts, np, hull_pts, stack, cal_scratch);
    }
    if (ub.h < 1.001 && ub.w < 1.001) { res->status = ST_SKIP_UNCLIP; return; }
    float clip[4][2], ssid;
    get_mini_boxes(ub, clip, &ssid);
    if (ssid < 5) { res->status = ST_SKIP_SSID2; return; }                        // min_size + 2
    const int src_w = src_wh[2 * img], src_h = src_wh[2 * img + 1];
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
    res->status = ST_OK;
}

// small-footprint pass: one workgroup per border
__global__ __launch_bounds__(CT_THREADS, 6) void contour_kernel(const float *__restrict__ maps, const Cand *__restrict__ cands,
                                                                const int *__restrict__ totals, const CandInfo *__restrict__ info,
                                                                const unsigned *__restrict__ pool, unsigned *__restrict__ gslots,
                                                                int *__restrict__ slot_locks, Result *__restrict__ results,
                                                                int *__restrict__ flags, const int *__restrict__ src_wh,
                                                                float box_thresh, float unclip_ratio, long slot_words, int use_padding_resize,
                                                                DbpostDims d) {
    const int img = blockIdx.y, k = blockIdx.x;
    if (k >= min(totals[img], MAX_CAND)) return;
    contour_body<512, 960, 80, true>(img, k, maps, cands, info, pool, gslots, slot_locks, results, flags, src_wh, box_thresh, unclip_ratio,
                                     slot_words, use_padding_resize, d);
}

// full-size pass: a few workgroups per image walk the borders the small pass deferred (usually none)
__global__ __launch_bounds__(CT_THREADS, 1) void contour_big_kernel(const float *__restrict__ maps, const Cand *__restrict__ cands,
                                                                    const int *__restrict__ totals, const CandInfo *__restrict__ info,
                                                                    const unsigned *__restrict__ pool, unsigned *__restrict__ gslots,
                                                                    int *__restrict__ slot_locks, Result *__restrict__ results,
                                                                    int *__restrict__ flags, const int *__restrict__ src_wh,
                                                                    float box_thresh, float unclip_ratio, long slot_words,
                                                                    int use_padding_resize, DbpostDims d) {
    const int img = blockIdx.y;
    if (!(flags[img] & 8)) return;                        // internal bit 3: the small pass deferred at least one border of this image
    const int num = min(totals[img], MAX_CAND);
    for (int k = blockIdx.x; k < num; k += gridDim.x) {
        if (results[(long)img * MAX_CAND + k].status != ST_DEFER) continue;       // uniform over the workgroup
        contour_body<MAXW, LDS_PLANE_WORDS, MAXHULL, false>(img, k, maps, cands, info, pool, gslots, slot_locks, results, flags, src_wh,
                                                            box_thresh, unclip_ratio, slot_words, use_padding_resize, d);
        __syncthreads();                                                          // the body's LDS is reused by the next border
    }
}

// boxes of one image in candidate order -> dense int16 list + count
__global__ __launch_bounds__(1024) void compact_kernel(const Result *__restrict__ results, const int *__restrict__ totals,
                                                       short *__restrict__ boxes, int *__restrict__ counts, int max_boxes) {
    const int img = blockIdx.x, k = threadIdx.x;
    __shared__ int sh[1024];
    const int num = min(totals[img], MAX_CAND);
    const bool ok = k < num && results[(long)img * MAX_CAND + k].status == ST_OK;
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
        if (pos < max_boxes)
            for (int j = 0; j < 8; j++) boxes[((long)img * max_boxes + pos) * 8 + j] = (short)results[(long)img * MAX_CAND + k].box[j];
    }
    if (k == 1023) counts[img] = sh[1023];
}

}  // namespace ptocr

using namespace ptocr;

struct ptocr_dbpost {
    int max_n, max_h, max_w;
    unsigned *bits; unsigned *bits2; int *labels; int *chunk_cnt; int *totals; int *strip_totals; Cand *cands; CandInfo *info; unsigned *pool;
    unsigned *gslots; int *slot_locks; long slot_words; Result *results; int *flags; int *src_wh; short *boxes; int *counts;
    int boxes_cap;
    long pool_cap;
    hipEvent_t ev0, ev1;          // device time of the last call's kernels (ptocr_dbpost_last_device_ms)
    int timed;
};

extern "C" int ptocr_dbpost_create(ptocr_dbpost_t *out, int max_n, int max_h, int max_w) {
    PT_CHECK(out && max_n > 0 && max_h > 0 && max_w > 0, "ptocr_dbpost_create: bad arguments");
    PT_CHECK(max_w <= MAXW && max_h < 32768 && (long)max_h * max_w < (1L << 31), "ptocr_dbpost_create: map larger than %d wide / 32767 high", MAXW);
    ptocr_dbpost *h = new ptocr_dbpost();
    memset(h, 0, sizeof *h);
    h->max_n = max_n; h->max_h = max_h; h->max_w = max_w;
    const long hw = (long)max_h * max_w, ww = cdiv(max_w, 32);
    h->pool_cap = 2 * hw;
    h->boxes_cap = MAX_CAND;
    const long nch = (hw + CHUNK - 1) / CHUNK;
    PT_HIP(hipMalloc(&h->bits, sizeof(unsigned) * max_n * max_h * ww));
    PT_HIP(hipMalloc(&h->bits2, sizeof(unsigned) * max_n * max_h * ww));
    PT_HIP(hipMalloc(&h->labels, sizeof(int) * max_n * hw));
    PT_HIP(hipMalloc(&h->chunk_cnt, sizeof(int) * max_n * nch));
    PT_HIP(hipMalloc(&h->totals, sizeof(int) * max_n));
    PT_HIP(hipMalloc(&h->strip_totals, sizeof(int) * max_n));
    PT_HIP(hipMalloc(&h->cands, sizeof(Cand) * max_n * MAX_CAND));
    PT_HIP(hipMalloc(&h->info, sizeof(CandInfo) * max_n * MAX_CAND));
    PT_HIP(hipMalloc(&h->pool, sizeof(unsigned) * max_n * (QOFF + h->pool_cap)));
    h->slot_words = (long)max_h * ww + 64;
    PT_HIP(hipMalloc(&h->gslots, sizeof(unsigned) * NSLOTS * 2 * h->slot_words));
    PT_HIP(hipMalloc(&h->slot_locks, sizeof(int) * NSLOTS));
    PT_HIP(hipMemset(h->slot_locks, 0, sizeof(int) * NSLOTS));
    PT_HIP(hipMalloc(&h->results, sizeof(Result) * max_n * MAX_CAND));
    PT_HIP(hipMalloc(&h->flags, sizeof(int) * max_n));
    PT_HIP(hipMalloc(&h->src_wh, sizeof(int) * 2 * max_n));
    PT_HIP(hipMalloc(&h->boxes, sizeof(short) * 8 * max_n * MAX_CAND));
    PT_HIP(hipMalloc(&h->counts, sizeof(int) * max_n));
    PT_HIP(hipEventCreate(&h->ev0));
    PT_HIP(hipEventCreate(&h->ev1));
    *out = h;
    return 0;
}

extern "C" int ptocr_dbpost_destroy(ptocr_dbpost_t h) {
    if (!h) return 0;
    void *bufs[] = {h->bits, h->bits2, h->labels, h->chunk_cnt, h->totals, h->strip_totals, h->cands, h->info, h->pool, h->gslots, h->slot_locks,
                    h->results, h->flags, h->src_wh, h->boxes, h->counts};
    for (void *b : bufs) (void)hipFree(b);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return 0;
}

// debug/inspection hook used by the parity tests: copies the per-candidate records of the last call
extern "C" int ptocr_dbpost_debug_results(ptocr_dbpost_t h, int img, int32_t *h_total, void *h_results /* MAX_CAND x Result */,
                                          void *h_cands /* MAX_CAND x Cand */, void *h_info /* MAX_CAND x CandInfo */) {
    PT_CHECK(h && img >= 0 && img < h->max_n, "ptocr_dbpost_debug_results: bad arguments");
    PT_HIP(hipDeviceSynchronize());
    PT_HIP(hipMemcpy(h_total, h->totals + img, sizeof(int), hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_results, h->results + (long)img * MAX_CAND, sizeof(Result) * MAX_CAND, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_cands, h->cands + (long)img * MAX_CAND, sizeof(Cand) * MAX_CAND, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(h_info, h->info + (long)img * MAX_CAND, sizeof(CandInfo) * MAX_CAND, hipMemcpyDeviceToHost));
    return 0;
}

// device time (HIP events on the call's stream) from the first to the last kernel of the last call on this workspace
extern "C" int ptocr_dbpost_last_device_ms(ptocr_dbpost_t h, float *ms) {
    PT_CHECK(h && ms && h->timed, "ptocr_dbpost_last_device_ms: no completed call on this workspace");
    PT_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
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
    DbpostDims d;
    d.N = N; d.H = H; d.W = W; d.WW = cdiv(W, 32); d.HW = (long)H * W;
    d.nchunks = (int)((d.HW + CHUNK - 1) / CHUNK);
    d.pool_cap = h->pool_cap;
    d.pool_stride = QOFF + h->pool_cap;
    PT_HIP(hipMemcpyAsync(h->src_wh, h_src_wh, sizeof(int) * 2 * N, hipMemcpyHostToDevice, s));
    PT_HIP(hipEventRecord(h->ev0, s));
    PT_HIP(hipMemsetAsync(h->flags, 0, sizeof(int) * N, s));
    const dim3 row_grid(cdiv(W, 1024), H, N);
    if (d_bitmap) hipLaunchKernelGGL(pack_u8_kernel, row_grid, dim3(256), 0, s, d_bitmap, h->bits, d);
    else hipLaunchKernelGGL(binarize_kernel, row_grid, dim3(256), 0, s, d_maps, h->bits, d, thresh);
    unsigned *bits = h->bits;
    if (use_dilation) {
        hipLaunchKernelGGL(dilate2x2_kernel, dim3(cdiv(d.WW, 256), H, N), dim3(256), 0, s, h->bits, h->bits2, d);
        bits = h->bits2;
    }
    PT_HIP(hipMemsetAsync(h->chunk_cnt, 0, sizeof(int) * (size_t)N * d.nchunks, s));
    const int strip_y = H > 2 * STRIP_ROWS ? H - STRIP_ROWS : 0;      // small maps: one pass over everything
    for (int pass = strip_y ? 0 : 1; pass < 2; pass++) {
        CclPass ps;
        ps.y_first = pass == 0 ? strip_y : 0;
        ps.skip_if_full = (pass == 1 && strip_y) ? h->strip_totals : nullptr;
        const int words = (H - ps.y_first) * d.WW;
        const dim3 word_grid(cdiv(words > d.nchunks ? words : d.nchunks, 256), N);
        hipLaunchKernelGGL(ccl_init_kernel, word_grid, dim3(256), 0, s, bits, h->labels, h->chunk_cnt, d, ps);
        hipLaunchKernelGGL(ccl_merge_kernel, dim3(cdiv(4 * words, 256), N), dim3(256), 0, s, bits, h->labels, d, ps);
        hipLaunchKernelGGL(ccl_flatten_kernel, word_grid, dim3(256), 0, s, bits, h->labels, h->chunk_cnt, d, ps);
        hipLaunchKernelGGL(chunk_suffix_kernel, dim3(N), dim3(1024), 0, s, h->chunk_cnt, h->totals, d, ps, h->strip_totals);
    }
    hipLaunchKernelGGL(gather_starts_kernel, dim3(d.nchunks, N), dim3(256), 0, s, bits, h->labels, h->chunk_cnt, h->cands, d);
    hipLaunchKernelGGL(trace_count_kernel, dim3(MAX_CAND, N), dim3(64), 0, s, bits, h->cands, h->totals, h->info, h->pool, d);
    hipLaunchKernelGGL(pool_offsets_kernel, dim3(N), dim3(1024), 0, s, h->info, h->totals, h->flags, d);
    hipLaunchKernelGGL(trace_write_kernel, dim3(MAX_CAND, N), dim3(64), 0, s, bits, h->cands, h->totals, h->info,
                       h->pool, h->flags, d);
    hipLaunchKernelGGL(contour_kernel, dim3(MAX_CAND, N), dim3(CT_THREADS), 0, s, d_maps, h->cands, h->totals, h->info, h->pool, h->gslots,
                       h->slot_locks, h->results, h->flags, h->src_wh, box_thresh, unclip_ratio, h->slot_words, use_padding_resize, d);
    hipLaunchKernelGGL(contour_big_kernel, dim3(64, N), dim3(CT_THREADS), 0, s, d_maps, h->cands, h->totals, h->info, h->pool, h->gslots,
                       h->slot_locks, h->results, h->flags, h->src_wh, box_thresh, unclip_ratio, h->slot_words, use_padding_resize, d);
    hipLaunchKernelGGL(compact_kernel, dim3(N), dim3(1024), 0, s, h->results, h->totals, h->boxes, h->counts, max_boxes);
    if (int e = launch_ok("dbpost kernels")) return e;
    PT_HIP(hipEventRecord(h->ev1, s));
    h->timed = 1;
    PT_HIP(hipMemcpyAsync(h_counts, h->counts, sizeof(int) * N, hipMemcpyDeviceToHost, s));
    PT_HIP(hipMemcpyAsync(h_flags, h->flags, sizeof(int) * N, hipMemcpyDeviceToHost, s));
    PT_HIP(hipMemcpyAsync(h_boxes, h->boxes, sizeof(short) * 8 * (size_t)N * max_boxes, hipMemcpyDeviceToHost, s));
    PT_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < N; i++) h_flags[i] &= 7;           // bit 3 is internal (deferred borders)
    for (int i = 0; i < N; i++)
        if (h_flags[i] & 4) return fail("ptocr_db_postprocess: internal capacity exceeded on image %d (point pool %ld points or hull "
                                        "candidates %d)", i, h->pool_cap, MAXHULL);
    return 0;
}
