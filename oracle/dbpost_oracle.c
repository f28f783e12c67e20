/*
 * ORACLE -- test infrastructure only.  Nothing under pytorchocr_amd/ may include, link or call this file.
 *
 * Plain-C, single-thread CPU restatement of the reference's DB box extraction with cpp_speedup=True:
 *   pytocr/postprocess/db_postprocess_fast/src/db_postprocess.cpp
 *     :16-32   GetContourArea      -> unclip_distance()
 *     :34-64   UnClip              -> unclip_rect()
 *     :147-151 XsortFp32, :159-192 GetMiniBoxes -> get_mini_boxes()
 *     :194-229 BoxScore            -> box_score()
 *     :231-317 BoxesFromBitmap     -> dbpost_oracle_run()
 *   pytocr/postprocess/db_postprocess_fast/src/clipper.cpp (Clipper 6.4.2, vendored)
 *     :136 Round, :399-411 Area, :3799-3811 GetUnitNormal, :3837-3879 AddPath, :3889-3913 FixOrientations,
 *     :3987-4020 DoOffset (arc steps), :4160-4201 OffsetPoint, :4225-4244 DoRound  -> clipper_offset_round()
 *     :3916-3943 ClipperOffset::Execute's union (Clipper::Execute, ctUnion / pftPositive), restated as far as the hull of the
 *     solution goes: :1097-1130 AddPath's clean-up, :623-627 TopX, :3102-3136 the shared-edge join  -> dbpost_oracle_clipper_union()
 *     Offset + union are PINNED against the real vendored Clipper compiled into oracle/_ref/
 *     (tests/test_oracle_clipper.py): same minAreaRect input hull and emptiness on random, degenerate and sub-pixel boxes.
 *
 * The OpenCV calls of the reference are NOT under /root/reference (un-vendored dependency, pinned by the
 * reference at opencv 3.4.2 / opencv-python 4.1.2.30) and OpenCV is absent from this image, so they are
 * restated from OpenCV's published algorithms.  PARITY UNPINNED at this boundary:
 *   cv::findContours(RETR_LIST, CHAIN_APPROX_SIMPLE)  Suzuki-Abe border following   -> find_contours()
 *   cv::minAreaRect   (Sklansky hull on x-sorted points + float32 rotating calipers) -> min_area_rect_*()
 *   cv::boxPoints                                                                    -> box_points()
 *   cv::fillPoly(mask, pts, 1, lineType=1) (edge lines 4-connected + even-odd scanline fill) -> fill_poly()
 *   cv::mean(crop, mask)  (raster-order sum in double / count)                       -> inside box_score()
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: float expressions must not be fused).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dbpost_oracle.h"

/* ------------------------------------------------------------------------------------------------
 * findContours(RETR_LIST, CHAIN_APPROX_SIMPLE)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int x, y; } ipt;

typedef struct {
    ipt *pts;
    int n, cap;
    int is_hole;
    int trig_x, trig_y;   /* raster position of the pixel whose scan triggered the trace */
} contour_t;

static void contour_push(contour_t *c, int x, int y) {
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 16;
        c->pts = (ipt *)realloc(c->pts, (size_t)c->cap * sizeof(ipt));
    }
    c->pts[c->n].x = x; c->pts[c->n].y = y; c->n++;
}

/* 8-neighbour code deltas: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (image y grows downwards) */
static const int code_dx[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
static const int code_dy[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

/* Border following from start pixel i0 (in the padded label image), writing the points at which the
 * chain direction changes.  Marks visited pixels: 2 = visited, -126 = visited and its east neighbour was
 * examined as zero ("right bound"), exactly what the raster scan needs to start every border once. */
static void fetch_contour(signed char *img, int step, int px, int py, int is_hole, contour_t *out,
                          int off_x, int off_y) {
    const signed char nbd = 2;
    int deltas[16];
    signed char *i0 = img + (size_t)py * step + px, *i1, *i3, *i4 = 0;
    int prev_s, s, s_end, k;
    int x = px + off_x, y = py + off_y;
    for (k = 0; k < 8; k++) deltas[k] = code_dy[k] * step + code_dx[k];
    for (k = 0; k < 8; k++) deltas[k + 8] = deltas[k];

    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) {                 /* single pixel domain */
        *i0 = (signed char)(nbd | -128);
        contour_push(out, x, y);
        return;
    }
    i3 = i0;
    prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1) *i3 = nbd;

        if (s != prev_s) {
            contour_push(out, x, y);
            prev_s = s;
        }
        x += code_dx[s]; y += code_dy[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* Returns contours in OpenCV's RETR_LIST output order: the contour discovered LAST comes first. */
static contour_t *find_contours(const uint8_t *bitmap, int H, int W, int *count_out) {
    int step = W + 2, x, y, n = 0, cap = 0, i;
    signed char *img = (signed char *)calloc((size_t)(H + 2) * step, 1);
    contour_t *list = 0;
    for (y = 0; y < H; y++)
        for (x = 0; x < W; x++) img[(size_t)(y + 1) * step + x + 1] = bitmap[(size_t)y * W + x] ? 1 : 0;

    for (y = 1; y <= H; y++) {
        signed char *row = img + (size_t)y * step;
        int prev = 0;
        for (x = 1; x <= W; x++) {
            int p = row[x];
            if (p != prev) {
                int is_hole = 0, start = 0;
                if (prev == 0 && p == 1) start = 1;                 /* outer border */
                else if (p == 0 && prev >= 1) { start = 1; is_hole = 1; }  /* hole border */
                if (start) {
                    contour_t c; memset(&c, 0, sizeof c);
                    c.is_hole = is_hole; c.trig_x = x - 1; c.trig_y = y - 1;
                    fetch_contour(img, step, x - is_hole, y, is_hole, &c, -1, -1);
                    if (n == cap) { cap = cap ? cap * 2 : 64; list = (contour_t *)realloc(list, (size_t)cap * sizeof(contour_t)); }
                    list[n++] = c;
                    p = row[x];                                     /* the start pixel may be marked now */
                }
                prev = p;
            }
        }
    }
    free(img);
    for (i = 0; i < n / 2; i++) { contour_t t = list[i]; list[i] = list[n - 1 - i]; list[n - 1 - i] = t; }
    *count_out = n;
    return list;
}

/* ------------------------------------------------------------------------------------------------
 * minAreaRect = convexHull (Sklansky on points sorted by x then y) + rotating calipers in float32
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } fpt;
typedef struct { float cx, cy, w, h, angle; } rrect;

static int sign_d(double v) { return (v > 0) - (v < 0); }

/* Upper/lower chain scan over the x-sorted array (indices start..end step +-1). */
static int sklansky(const fpt *a, int start, int end, int *stack, int nsign, int sign2) {
    int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (a[start].x == a[end].x && a[start].y == a[end].y)) {
        stack[0] = start;
        return 1;
    }
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        float cury = a[pcur].y, nexty = a[pnext].y;
        float by = nexty - cury;
        if (sign_d(by) != nsign) {
            float ax = a[pcur].x - a[pprev].x;
            float bx = a[pnext].x - a[pcur].x;
            float ay = cury - a[pprev].y;
            double convexity = (double)ay * bx - (double)ax * by;
            if (sign_d(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pcur = pnext; pnext += incr;
                stack[stacksize] = pnext; stacksize++;
            } else {
                if (pprev == stack[0]) {
                    pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
                } else {
                    stack[stacksize - 2] = pnext;
                    pcur = pprev; pprev = stack[stacksize - 4];
                    stacksize--;
                }
            }
        } else {
            pnext += incr;
            stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

static int cmp_fpt(const void *pa, const void *pb) {
    const fpt *a = (const fpt *)pa, *b = (const fpt *)pb;
    if (a->x < b->x) return -1;
    if (a->x > b->x) return 1;
    if (a->y < b->y) return -1;
    if (a->y > b->y) return 1;
    return 0;
}

/* convexHull(points, clockwise=true, returnPoints=true).  hull must hold n points; returns hull size. */
static int convex_hull(const fpt *pts_in, int n, fpt *hull) {
    fpt *a = (fpt *)malloc((size_t)n * sizeof(fpt));
    int *stack = (int *)malloc((size_t)(n + 2) * sizeof(int) * 2);
    int i, nout = 0, miny_ind = 0, maxy_ind = 0;
    memcpy(a, pts_in, (size_t)n * sizeof(fpt));
    qsort(a, (size_t)n, sizeof(fpt), cmp_fpt);
    for (i = 1; i < n; i++) {
        float y = a[i].y;
        if (a[miny_ind].y > y) miny_ind = i;
        if (a[maxy_ind].y < y) maxy_ind = i;
    }
    if (a[0].x == a[n - 1].x && a[0].y == a[n - 1].y) {
        hull[nout++] = a[0];
    } else {
        int *tl_stack = stack;
        int tl_count = sklansky(a, 0, maxy_ind, tl_stack, -1, 1);
        int *tr_stack = stack + tl_count;
        int tr_count = sklansky(a, n - 1, maxy_ind, tr_stack, -1, -1);
        int stop_idx, *bl_stack, *br_stack, bl_count, br_count;
        /* clockwise == true: no swap of the upper chains */
        for (i = 0; i < tl_count - 1; i++) hull[nout++] = a[tl_stack[i]];
        for (i = tr_count - 1; i > 0; i--) hull[nout++] = a[tr_stack[i]];
        stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;

        /* note: the upper chains have been copied out, the stack is reused */
        bl_stack = stack;
        bl_count = sklansky(a, 0, miny_ind, bl_stack, 1, -1);
        br_stack = stack + bl_count;
        br_count = sklansky(a, n - 1, miny_ind, br_stack, 1, 1);
        { int *ts = bl_stack; int tc = bl_count; bl_stack = br_stack; bl_count = br_count; br_stack = ts; br_count = tc; }
        if (stop_idx >= 0) {
            int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
            if (check_idx == stop_idx || (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
                bl_count = bl_count < 2 ? bl_count : 2;
                br_count = br_count < 2 ? br_count : 2;
            }
        }
        for (i = 0; i < bl_count - 1; i++) hull[nout++] = a[bl_stack[i]];
        for (i = br_count - 1; i > 0; i--) hull[nout++] = a[br_stack[i]];
    }
    free(a); free(stack);
    return nout;
}

static void rotating_calipers_minarea(const fpt *points, int n, float *out /*6*/) {
    float minarea = 3.402823466e+38f;
    float *inv_vect_length = (float *)malloc((size_t)n * 3 * sizeof(float));
    fpt *vect = (fpt *)(inv_vect_length + n);
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4] = { -1, -1, -1, -1 };
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    fpt pt0 = points[0];
    int i, k;
    int buf_left = 0, buf_bottom = 0;
    float buf_a = 0, buf_w = 0, buf_b = 0, buf_h = 0;

    left_x = right_x = pt0.x;
    top_y = bottom_y = pt0.y;
    for (i = 0; i < n; i++) {
        double dx, dy;
        fpt pt;
        if (pt0.x < left_x) { left_x = pt0.x; left = i; }
        if (pt0.x > right_x) { right_x = pt0.x; right = i; }
        if (pt0.y > top_y) { top_y = pt0.y; top = i; }
        if (pt0.y < bottom_y) { bottom_y = pt0.y; bottom = i; }
        pt = points[(i + 1) < n ? (i + 1) : 0];
        dx = pt.x - pt0.x;
        dy = pt.y - pt0.y;
        vect[i].x = (float)dx;
        vect[i].y = (float)dy;
        inv_vect_length[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0 = pt;
    }
    {
        double ax = vect[n - 1].x, ay = vect[n - 1].y;
        for (i = 0; i < n; i++) {
            double bx = vect[i].x, by = vect[i].y;
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;

    for (k = 0; k < n; k++) {
        float dp[4];
        float maxcos;
        int main_element = 0;
        dp[0] = +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y;
        dp[1] = -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y;
        dp[2] = -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y;
        dp[3] = +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y;
        maxcos = dp[0] * inv_vect_length[seq[0]];
        for (i = 1; i < 4; ++i) {
            float cosalpha = dp[i] * inv_vect_length[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vect[pindex].x * inv_vect_length[pindex];
            float lead_y = vect[pindex].y * inv_vect_length[pindex];
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
            float height, area, width;
            float dx = points[seq[1]].x - points[seq[3]].x;
            float dy = points[seq[1]].y - points[seq[3]].y;
            width = dx * base_a + dy * base_b;
            dx = points[seq[2]].x - points[seq[0]].x;
            dy = points[seq[2]].y - points[seq[0]].y;
            height = -dx * base_b + dy * base_a;
            area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_left = seq[3]; buf_a = base_a; buf_w = width; buf_b = base_b; buf_h = height; buf_bottom = seq[0];
            }
        }
    }
    {
        float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
        float C1 = A1 * points[buf_left].x + points[buf_left].y * B1;
        float C2 = A2 * points[buf_bottom].x + points[buf_bottom].y * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float px = (C1 * B2 - C2 * B1) * idet;
        float py = (A1 * C2 - A2 * C1) * idet;
        out[0] = px; out[1] = py;
        out[2] = A1 * buf_w; out[3] = B1 * buf_w;
        out[4] = A2 * buf_h; out[5] = B2 * buf_h;
    }
    free(inv_vect_length);
}

static const double CV_PI_D = 3.1415926535897932384626433832795;

static rrect min_area_rect_f(const fpt *pts, int n) {
    rrect box; fpt *hull; int hn; float out[6];
    memset(&box, 0, sizeof box);
    if (n <= 0) return box;
    hull = (fpt *)malloc((size_t)n * sizeof(fpt));
    hn = convex_hull(pts, n, hull);
    if (hn > 2) {
        rotating_calipers_minarea(hull, hn, out);
        box.cx = out[0] + (out[2] + out[4]) * 0.5f;
        box.cy = out[1] + (out[3] + out[5]) * 0.5f;
        box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box.angle = (float)atan2((double)out[3], (double)out[2]);
    } else if (hn == 2) {
        double dx, dy;
        box.cx = (hull[0].x + hull[1].x) * 0.5f;
        box.cy = (hull[0].y + hull[1].y) * 0.5f;
        dx = hull[1].x - hull[0].x; dy = hull[1].y - hull[0].y;
        box.w = (float)sqrt(dx * dx + dy * dy);
        box.h = 0;
        box.angle = (float)atan2(dy, dx);
    } else if (hn == 1) {
        box.cx = hull[0].x; box.cy = hull[0].y;
    }
    box.angle = (float)(box.angle * 180 / CV_PI_D);
    free(hull);
    return box;
}

static rrect min_area_rect_i(const ipt *pts, int n) {
    fpt *f = (fpt *)malloc((size_t)(n > 0 ? n : 1) * sizeof(fpt));
    rrect r; int i;
    for (i = 0; i < n; i++) { f[i].x = (float)pts[i].x; f[i].y = (float)pts[i].y; }
    r = min_area_rect_f(f, n);
    free(f);
    return r;
}

static void box_points(rrect r, fpt pt[4]) {
    double _angle = r.angle * CV_PI_D / 180.;
    float b = (float)cos(_angle) * 0.5f;
    float a = (float)sin(_angle) * 0.5f;
    pt[0].x = r.cx - a * r.h - b * r.w;
    pt[0].y = r.cy + b * r.h - a * r.w;
    pt[1].x = r.cx + a * r.h - b * r.w;
    pt[1].y = r.cy - b * r.h - a * r.w;
    pt[2].x = 2 * r.cx - pt[0].x;
    pt[2].y = 2 * r.cy - pt[0].y;
    pt[3].x = 2 * r.cx - pt[1].x;
    pt[3].y = 2 * r.cy - pt[1].y;
}

/* db_postprocess.cpp:159-192.  std::sort on 4 elements is an insertion sort => stable on x ties. */
static void get_mini_boxes(rrect box, float out[4][2], float *ssid) {
    fpt p[4], t; int i, j;
    fpt idx1, idx2, idx3, idx4;
    *ssid = box.w > box.h ? box.w : box.h;   /* std::max(width, height) */
    box_points(box, p);
    for (i = 1; i < 4; i++) {
        t = p[i];
        for (j = i; j > 0 && t.x < p[j - 1].x; j--) p[j] = p[j - 1];
        p[j] = t;
    }
    if (p[3].y <= p[2].y) { idx2 = p[3]; idx3 = p[2]; } else { idx2 = p[2]; idx3 = p[3]; }
    if (p[1].y <= p[0].y) { idx1 = p[1]; idx4 = p[0]; } else { idx1 = p[0]; idx4 = p[1]; }
    out[0][0] = idx1.x; out[0][1] = idx1.y;
    out[1][0] = idx2.x; out[1][1] = idx2.y;
    out[2][0] = idx3.x; out[2][1] = idx3.y;
    out[3][0] = idx4.x; out[3][1] = idx4.y;
}

/* ------------------------------------------------------------------------------------------------
 * fillPoly(mask, {poly}, 1, lineType = 1): Line() maps connectivity 1 -> 4-connected; then even-odd fill
 * ---------------------------------------------------------------------------------------------- */
static void draw_line4(uint8_t *mask, int mw, int mh, int x1, int y1, int x2, int y2) {
    /* LineIterator(img, pt1, pt2, connectivity=4, leftToRight=true) restated on coordinates */
    int dx = x2 - x1, dy = y2 - y1, sx = 1, sy, count, err, plusDelta, minusDelta, i;
    int swap_xy;
    int x, y;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }   /* always go left to right */
    sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    swap_xy = dy > dx;
    if (swap_xy) { int t = dx; dx = dy; dy = t; }
    err = 0;
    plusDelta = (dx + dx) + (dy + dy);
    minusDelta = -(dy + dy);
    count = dx + dy + 1;
    x = x1; y = y1;
    for (i = 0; i < count; i++) {
        int m;
        if (x >= 0 && x < mw && y >= 0 && y < mh) mask[(size_t)y * mw + x] = 1;
        m = err < 0 ? -1 : 0;
        err += minusDelta + (plusDelta & m);
        /* ptr += minusStep + (plusStep & mask): minusStep = major-axis step, plusStep = minor - major */
        if (m) { if (swap_xy) x += sx; else y += sy; }
        else   { if (swap_xy) y += sy; else x += sx; }
    }
}

typedef struct { int y0, y1; int64_t x, dx; } pedge;

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

static void fill_poly(uint8_t *mask, int mw, int mh, const ipt *v, int count) {
    const int XY_SHIFT = 16; const int64_t XY_ONE = 1 << 16;
    pedge *edges = (pedge *)malloc((size_t)(count + 1) * sizeof(pedge));
    int64_t *xs = (int64_t *)malloc((size_t)(count + 1) * sizeof(int64_t));
    int ne = 0, i, y, ymin = 0x7fffffff, ymax = -0x7fffffff;
    ipt p0 = v[count - 1];
    for (i = 0; i < count; i++) {
        ipt p1 = v[i];
        draw_line4(mask, mw, mh, p0.x, p0.y, p1.x, p1.y);
        if (p0.y != p1.y) {
            pedge e;
            int64_t x0 = (int64_t)p0.x << XY_SHIFT, x1 = (int64_t)p1.x << XY_SHIFT;
            if (p0.y < p1.y) { e.y0 = p0.y; e.y1 = p1.y; e.x = x0; }
            else             { e.y0 = p1.y; e.y1 = p0.y; e.x = x1; }
            e.dx = (x1 - x0) / (p1.y - p0.y);
            edges[ne++] = e;
            if (e.y0 < ymin) ymin = e.y0;
            if (e.y1 > ymax) ymax = e.y1;
        }
        p0 = p1;
    }
    if (ne >= 2) {
        if (ymax > mh) ymax = mh;
        for (y = ymin; y < ymax; y++) {
            int na = 0, k;
            for (i = 0; i < ne; i++)
                if (edges[i].y0 <= y && y < edges[i].y1) xs[na++] = edges[i].x + (int64_t)(y - edges[i].y0) * edges[i].dx;
            if (y < 0) continue;
            qsort(xs, (size_t)na, sizeof(int64_t), cmp_i64);
            for (k = 0; k + 1 < na; k += 2) {
                int xa = (int)((xs[k] + XY_ONE - 1) >> XY_SHIFT);
                int xb = (int)(xs[k + 1] >> XY_SHIFT);
                if (xa < mw && xb >= 0) {
                    int xx;
                    if (xa < 0) xa = 0;
                    if (xb >= mw) xb = mw - 1;
                    for (xx = xa; xx <= xb; xx++) mask[(size_t)y * mw + xx] = 1;
                }
            }
        }
    }
    free(edges); free(xs);
}

/* db_postprocess.cpp:194-229 */
static float box_score(const contour_t *c, const float *pred, int H, int W, int *npix_out) {
    int xmin = W, xmax = -1, ymin = H, ymax = -1, i, mw, mh, x, y, cnt = 0;
    ipt *poly = (ipt *)malloc((size_t)c->n * sizeof(ipt));
    uint8_t *mask;
    double s = 0;
    for (i = 0; i < c->n; i++) {
        xmin = xmin > c->pts[i].x ? c->pts[i].x : xmin;
        xmax = xmax < c->pts[i].x ? c->pts[i].x : xmax;
        ymin = ymin > c->pts[i].y ? c->pts[i].y : ymin;
        ymax = ymax < c->pts[i].y ? c->pts[i].y : ymax;
    }
    xmax = xmax > 0 ? xmax : 0; xmax = xmax < W - 1 ? xmax : W - 1;
    xmin = xmin < W - 1 ? xmin : W - 1; xmin = xmin > 0 ? xmin : 0;
    ymax = ymax > 0 ? ymax : 0; ymax = ymax < H - 1 ? ymax : H - 1;
    ymin = ymin < H - 1 ? ymin : H - 1; ymin = ymin > 0 ? ymin : 0;
    for (i = 0; i < c->n; i++) { poly[i].x = c->pts[i].x - xmin; poly[i].y = c->pts[i].y - ymin; }
    mw = xmax - xmin + 1; mh = ymax - ymin + 1;
    mask = (uint8_t *)calloc((size_t)mw * mh, 1);
    fill_poly(mask, mw, mh, poly, c->n);
    for (y = 0; y < mh; y++)
        for (x = 0; x < mw; x++)
            if (mask[(size_t)y * mw + x]) { s += pred[(size_t)(y + ymin) * W + x + xmin]; cnt++; }
    free(mask); free(poly);
    if (npix_out) *npix_out = cnt;
    return (float)(cnt ? s / cnt : 0.0);
}

/* ------------------------------------------------------------------------------------------------
 * Clipper 6.4.2 ClipperOffset(miter 2.0, arc 0.25).AddPath(jtRound, etClosedPolygon).Execute(delta)
 * restated for ONE closed path.  Returns the offset polygon BEFORE the final union (dbpost_oracle_clipper_union below).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int64_t X, Y; } cpt;
typedef struct { double X, Y; } dpt;

static int64_t cl_round(double val) { return (val < 0) ? (int64_t)(val - 0.5) : (int64_t)(val + 0.5); }

static dpt unit_normal(cpt p1, cpt p2) {
    dpt r; double Dx, dy, f;
    if (p2.X == p1.X && p2.Y == p1.Y) { r.X = 0; r.Y = 0; return r; }
    Dx = (double)(p2.X - p1.X); dy = (double)(p2.Y - p1.Y);
    f = 1 * 1.0 / sqrt(Dx * Dx + dy * dy);
    Dx *= f; dy *= f;
    r.X = dy; r.Y = -Dx;
    return r;
}

int dbpost_oracle_clipper_offset(const long long *path_xy, int npts, double delta, long long *out_xy, int out_cap) {
    const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc_tolerance = 0.25;
    const double ArcTolerance = 0.25;      /* ClipperOffset() defaults: MiterLimit 2.0, ArcTolerance 0.25 */
    cpt src[8]; dpt normals[8];
    int highI = npts - 1, i, j = 0, len, nout = 0, k;
    double y, steps, m_sin, m_cos, steps_per_rad, a_area;
    if (npts <= 0 || npts > 8) return 0;
    /* AddPath: strip duplicate points (closing duplicates first, then consecutive) */
    while (highI > 0 && path_xy[0] == path_xy[2 * highI] && path_xy[1] == path_xy[2 * highI + 1]) highI--;
    src[0].X = path_xy[0]; src[0].Y = path_xy[1];
    for (i = 1; i <= highI; i++)
        if (src[j].X != path_xy[2 * i] || src[j].Y != path_xy[2 * i + 1]) {
            j++; src[j].X = path_xy[2 * i]; src[j].Y = path_xy[2 * i + 1];
        }
    if (j < 2) return 0;                     /* etClosedPolygon with < 3 vertices is dropped */
    len = j + 1;
    /* FixOrientations: only one path, so it is the one holding the lowest vertex */
    a_area = 0;
    for (i = 0, k = len - 1; i < len; ++i) { a_area += ((double)src[k].X + src[i].X) * ((double)src[k].Y - src[i].Y); k = i; }
    if (!(-a_area * 0.5 >= 0)) {
        for (i = 0; i < len / 2; i++) { cpt t = src[i]; src[i] = src[len - 1 - i]; src[len - 1 - i] = t; }
    }
    /* DoOffset */
    if (delta > -1.0e-20 && delta < 1.0e-20) {
        for (i = 0; i < len && nout < out_cap; i++) { out_xy[2 * nout] = src[i].X; out_xy[2 * nout + 1] = src[i].Y; nout++; }
        return nout;
    }
    if (ArcTolerance > fabs(delta) * def_arc_tolerance) y = fabs(delta) * def_arc_tolerance; else y = ArcTolerance;
    steps = pi / acos(1 - y / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    m_sin = sin(two_pi / steps);
    m_cos = cos(two_pi / steps);
    steps_per_rad = steps / two_pi;
    if (delta < 0.0) m_sin = -m_sin;
    if (delta <= 0 && len < 3) return 0;
    for (i = 0; i < len - 1; ++i) normals[i] = unit_normal(src[i], src[i + 1]);
    normals[len - 1] = unit_normal(src[len - 1], src[0]);
#define PUSH(px, py) do { if (nout < out_cap) { out_xy[2 * nout] = (px); out_xy[2 * nout + 1] = (py); } nout++; } while (0)
    k = len - 1;
    for (j = 0; j < len; ++j) {
        double sinA = normals[k].X * normals[j].Y - normals[j].X * normals[k].Y;
        int done = 0;
        if (fabs(sinA * delta) < 1.0) {
            double cosA = normals[k].X * normals[j].X + normals[j].Y * normals[k].Y;
            if (cosA > 0) {
                PUSH(cl_round(src[j].X + normals[k].X * delta), cl_round(src[j].Y + normals[k].Y * delta));
                done = 1;                     /* NB: returns before k = j, as the original does */
            }
        } else if (sinA > 1.0) sinA = 1.0;
        else if (sinA < -1.0) sinA = -1.0;
        if (done) continue;
        if (sinA * delta < 0) {
            PUSH(cl_round(src[j].X + normals[k].X * delta), cl_round(src[j].Y + normals[k].Y * delta));
            PUSH(src[j].X, src[j].Y);
            PUSH(cl_round(src[j].X + normals[j].X * delta), cl_round(src[j].Y + normals[j].Y * delta));
        } else {                               /* jtRound -> DoRound(j, k) */
            double a = atan2(sinA, normals[k].X * normals[j].X + normals[k].Y * normals[j].Y);
            int64_t r = cl_round(steps_per_rad * fabs(a));
            int nsteps = (int)r > 1 ? (int)r : 1, s;
            double X = normals[k].X, Y = normals[k].Y, X2;
            for (s = 0; s < nsteps; ++s) {
                PUSH(cl_round(src[j].X + X * delta), cl_round(src[j].Y + Y * delta));
                X2 = X;
                X = X * m_cos - m_sin * Y;
                Y = X2 * m_sin + Y * m_cos;
            }
            PUSH(cl_round(src[j].X + normals[j].X * delta), cl_round(src[j].Y + normals[j].Y * delta));
        }
        k = j;
    }
#undef PUSH
    return nout;
}

/* ------------------------------------------------------------------------------------------------
 * The union that ClipperOffset::Execute runs over its offset polygon (clipper.cpp:3916-3943: Clipper::Execute(ctUnion,
 * pftPositive) on the one path), restated for what it can do to the offset of a convex quadrilateral as far as
 * cv::minAreaRect can see (only the hull of the vertices and emptiness reach it):
 *   1. AddPath never turns duplicate or collinear vertices (spikes included) into edges (clipper.cpp:1097-1130).
 *   2. The polygon is two y-monotone bounds from the bottom (largest Y; a horizontal edge allowed there) to the top.  When
 *      the sweep promotes an intermediate vertex `a` of one bound (ProcessEdgesAtTopOfScanbeam, clipper.cpp:3102-3136) and the
 *      other bound's edge, which strictly spans that scan line, has its ROUNDED position TopX (clipper.cpp:623-627:
 *      Bot.X + Round(Dx * (Y - Bot.Y))) on `a`, and the rest of that edge, seen from that rounded position, has the slope of
 *      the edge leaving `a` (SlopesEqual on e->Curr, e->Top), the two output polygons "share an edge": AddJoin, and
 *      JoinCommonEdges later pinches the polygon at `a`.  What lies above `a` is bounded by two coincident edges, has no
 *      area and is discarded; what lies below keeps its vertices.  Fewer than three distinct vertices left: no solution.
 *   3. A polygon without area has no solution.
 * Anything else the sweep does (dropping further collinear points, choosing the start vertex) leaves the hull alone.
 * Pinned against the reference's own Clipper (oracle/_ref): 2 000 000 random boxes incl. 950 000 with an unclip distance
 * below 0.75 px, identical minAreaRect input hull and emptiness on every one (tests/test_oracle_clipper.py runs a sample).
 * In place on xy[2n]; returns the number of vertices kept (0: no solution).
 * ---------------------------------------------------------------------------------------------- */
static int cu_dedupe(long long *q, int m) {
    int i, k = 0;
    for (i = 0; i < m; i++)
        if (k == 0 || q[2 * (k - 1)] != q[2 * i] || q[2 * (k - 1) + 1] != q[2 * i + 1]) { q[2 * k] = q[2 * i]; q[2 * k + 1] = q[2 * i + 1]; k++; }
    while (k > 1 && q[0] == q[2 * (k - 1)] && q[1] == q[2 * (k - 1) + 1]) k--;
    return k;
}

int dbpost_oracle_clipper_union(long long *xy, int n) {
    long long q[2 * 512];
    int chainA[512], chainB[512];
    int m, i, j, nb = 0, nt = 0, b0, b1, t0, t1, la = 0, lb = 0, ia, ib;
    long long ymax, ymin, area2 = 0;
    if (n > 512) return n;
    for (i = 0; i < 2 * n; i++) q[i] = xy[i];
    m = cu_dedupe(q, n);
    for (;;) {                                          /* collinear vertices, one at a time */
        int found = 0;
        if (m < 3) return 0;
        for (i = 0; i < m && !found; i++) {
            const long long *a = q + 2 * ((i + m - 1) % m), *b = q + 2 * i, *c = q + 2 * ((i + 1) % m);
            if ((b[1] - a[1]) * (c[0] - b[0]) == (b[0] - a[0]) * (c[1] - b[1])) {
                for (j = i; j < m - 1; j++) { q[2 * j] = q[2 * j + 2]; q[2 * j + 1] = q[2 * j + 3]; }
                m = cu_dedupe(q, m - 1);
                found = 1;
            }
        }
        if (!found) break;
    }
    for (i = 0; i < m; i++) { j = (i + 1) % m; area2 += q[2 * i] * q[2 * j + 1] - q[2 * j] * q[2 * i + 1]; }
    if (area2 == 0) return 0;
    ymax = ymin = q[1];
    for (i = 1; i < m; i++) { if (q[2 * i + 1] > ymax) ymax = q[2 * i + 1]; if (q[2 * i + 1] < ymin) ymin = q[2 * i + 1]; }
    b0 = b1 = t0 = t1 = -1;
    for (i = 0; i < m; i++) {
        if (q[2 * i + 1] == ymax) { nb++; if (b0 < 0) b0 = i; else b1 = i; }
        if (q[2 * i + 1] == ymin) { nt++; if (t0 < 0) t0 = i; else t1 = i; }
    }
    /* the bottom (top) is one vertex or two neighbours; b0 -> b1 and t0 -> t1 in the direction of increasing index */
    if (nb == 1) b1 = b0; else if (nb == 2) { if ((b0 + 1) % m == b1) { } else if ((b1 + 1) % m == b0) { int t = b0; b0 = b1; b1 = t; } else return n; } else return n;
    if (nt == 1) t1 = t0; else if (nt == 2) { if ((t0 + 1) % m == t1) { } else if ((t1 + 1) % m == t0) { int t = t0; t0 = t1; t1 = t; } else return n; } else return n;
    for (i = b1;; i = (i + 1) % m) { chainA[la++] = i; if (i == t0) break; if (la > m) return n; }
    for (i = b0;; i = (i + m - 1) % m) { chainB[lb++] = i; if (i == t1) break; if (lb > m) return n; }
    for (i = 0; i + 1 < la; i++) if (!(q[2 * chainA[i + 1] + 1] < q[2 * chainA[i] + 1])) return n;
    for (i = 0; i + 1 < lb; i++) if (!(q[2 * chainB[i + 1] + 1] < q[2 * chainB[i] + 1])) return n;
    if (la + lb - (b0 == b1) - (t0 == t1) != m) return n;
    /* intermediate vertices of both bounds from the bottom up */
    ia = 1; ib = 1;
    while (ia < la - 1 || ib < lb - 1) {
        const int *C, *O; int k, lo, e;
        long long y, ax, ex, ey;
        int takeA;
        if (ia >= la - 1) takeA = 0; else if (ib >= lb - 1) takeA = 1; else takeA = q[2 * chainA[ia] + 1] >= q[2 * chainB[ib] + 1];
        if (takeA) { C = chainA; O = chainB; lo = lb; k = ia++; } else { C = chainB; O = chainA; lo = la; k = ib++; }
        ax = q[2 * C[k]]; y = q[2 * C[k] + 1]; ex = q[2 * C[k + 1]]; ey = q[2 * C[k + 1] + 1];
        for (e = 0; e + 1 < lo; e++) {
            const long long bx = q[2 * O[e]], by = q[2 * O[e] + 1], tx = q[2 * O[e + 1]], ty = q[2 * O[e + 1] + 1];
            if (by > y && y > ty) {
                const double dx = (double)(tx - bx) / (double)(ty - by);
                const long long x = bx + cl_round(dx * (double)(y - by));
                if (x == ax && (ey - y) * (tx - x) == (ex - x) * (ty - y)) {
                    int kept = 0, distinct = 0, u, v;
                    for (u = 0; u < m; u++)
                        if (q[2 * u + 1] >= y) { xy[2 * kept] = q[2 * u]; xy[2 * kept + 1] = q[2 * u + 1]; kept++; }
                    for (u = 0; u < kept; u++) {
                        for (v = 0; v < u; v++) if (xy[2 * v] == xy[2 * u] && xy[2 * v + 1] == xy[2 * u + 1]) break;
                        distinct += v == u;
                    }
                    return distinct >= 3 ? kept : 0;
                }
                break;
            }
        }
    }
    return n;
}

/* offset + union: the vertices of ClipperOffset::Execute's solution as far as their hull goes */
int dbpost_oracle_clipper_unclip(const long long *path_xy, int npts, double delta, long long *out_xy, int out_cap) {
    int n = dbpost_oracle_clipper_offset(path_xy, npts, delta, out_xy, out_cap);
    if (n <= 0 || n > out_cap) return n;
    return dbpost_oracle_clipper_union(out_xy, n);
}

/* db_postprocess.cpp:16-32 */
static float unclip_distance(float box[4][2], float unclip_ratio) {
    int i; float area = 0.0f, dist = 0.0f;
    for (i = 0; i < 4; i++) {
        int n = (i + 1) % 4;
        area += box[i][0] * box[n][1] - box[i][1] * box[n][0];
        dist += sqrtf((box[i][0] - box[n][0]) * (box[i][0] - box[n][0]) +
                      (box[i][1] - box[n][1]) * (box[i][1] - box[n][1]));
    }
    area = (float)fabs((double)(float)(area / 2.0));
    return area * unclip_ratio / dist;
}

/* Optional hook: the reference's own vendored Clipper (oracle/_ref/libclipper_ref.so, symbol
 * clipper_ref_offset).  When installed, the unclip polygon is exactly ClipperOffset::Execute's solution
 * (offset + union clean-up), gathered with the reference's loop db_postprocess.cpp:52-56. */
typedef int (*clipper_ref_fn)(const long long *, int, double, long long *, int, int *, int);
static clipper_ref_fn g_clipper_ref = 0;
void dbpost_oracle_set_clipper_ref(void *fn) { g_clipper_ref = (clipper_ref_fn)fn; }

/* db_postprocess.cpp:34-64 */
static rrect unclip_rect(float box[4][2], float unclip_ratio, float *distance_out, int *npoly_out) {
    long long path[8], out[2 * 512]; fpt pts[512]; int n, i; rrect res;
    float distance = unclip_distance(box, unclip_ratio);
    for (i = 0; i < 4; i++) { path[2 * i] = (long long)(int)box[i][0]; path[2 * i + 1] = (long long)(int)box[i][1]; }
    if (g_clipper_ref) {
        int sizes[64], r, npaths, tot, j, off = 0, m, k = 0;
        long long tmp[2 * 512];
        r = g_clipper_ref(path, 4, (double)distance, tmp, 512, sizes, 64);
        npaths = r / 100000; tot = r % 100000; (void)tot;
        m = npaths > 0 ? sizes[npaths - 1] : 0;        /* inner bound = size of the LAST path (reference quirk) */
        for (j = 0; j < npaths; j++) {
            for (i = 0; i < m && i < sizes[j] && k < 512; i++) { out[2 * k] = tmp[2 * (off + i)]; out[2 * k + 1] = tmp[2 * (off + i) + 1]; k++; }
            off += sizes[j];
        }
        n = k;
    } else {
        n = dbpost_oracle_clipper_unclip(path, 4, (double)distance, out, 512);
    }
    if (n > 512) n = 512;
    if (distance_out) *distance_out = distance;
    if (npoly_out) *npoly_out = n;
    if (n <= 0) { res.cx = 0; res.cy = 0; res.w = 1; res.h = 1; res.angle = 0; return res; }
    for (i = 0; i < n; i++) { pts[i].x = (float)out[2 * i]; pts[i].y = (float)out[2 * i + 1]; }
    return min_area_rect_f(pts, n);
}

static float clampf(float x, float lo, float hi) { if (x > hi) return hi; if (x < lo) return lo; return x; }

/* db_postprocess.cpp:111-145 get_affine_transform(center, img_maxsize, target_size, inv=1).t() + transform_preds.
 * The three point pairs describe a uniform scale about the centres (the third pair is consistent with it), so
 * cv::getAffineTransform's 6x6 LU solve is restated in closed form, in double on the float32 triangle coordinates. */
static void padding_resize_point(float x, float y, int src_w, int src_h, int target, float *ox, float *oy) {
    const float cx = (float)(src_w / 2.0), cy = (float)(src_h / 2.0);
    const int img_maxsize = src_w > src_h ? src_w : src_h;
    const float s0y = cy, s1y = cy + (float)((float)img_maxsize / 2.0);          /* srcTriangle[0].y, [1].y */
    const float d0 = (float)((float)target / 2.0), d1y = d0 + (float)((float)target / 2.0);
    const double scale = ((double)s1y - (double)s0y) / ((double)d1y - (double)d0);
    *ox = (float)(scale * ((double)x - (double)d0) + (double)cx);
    *oy = (float)(scale * ((double)y - (double)d0) + (double)cy);
}

static int g_use_padding_resize = 0;
void dbpost_oracle_set_padding_resize(int on) { g_use_padding_resize = on; }

/* cv2.dilate(mask, [[1,1],[1,1]]) of db_postprocess.py:52-55: anchor (1,1) => OR of the pixel with its left, upper and
 * upper-left neighbours (BORDER_CONSTANT with the neutral value). */
void dbpost_oracle_dilate2x2(const uint8_t *in, int H, int W, uint8_t *out) {
    int x, y;
    for (y = 0; y < H; y++)
        for (x = 0; x < W; x++) {
            int v = in[(size_t)y * W + x];
            if (x > 0) v |= in[(size_t)y * W + x - 1];
            if (y > 0) v |= in[(size_t)(y - 1) * W + x];
            if (x > 0 && y > 0) v |= in[(size_t)(y - 1) * W + x - 1];
            out[(size_t)y * W + x] = v ? 1 : 0;
        }
}

/* db_postprocess.cpp:231-317 */
int dbpost_oracle_run(const float *pred, const uint8_t *bitmap, int H, int W, float box_thresh,
                      float unclip_ratio, int src_w, int src_h, int *boxes_out, int max_boxes,
                      dbpost_oracle_dbg *dbg, int dbg_cap, int *n_contours_out) {
    const int min_size = 3, max_candidates = 1000;
    int ncont = 0, i, j, nboxes = 0, num;
    contour_t *cs = find_contours(bitmap, H, W, &ncont);
    num = ncont >= max_candidates ? max_candidates : ncont;
    if (n_contours_out) *n_contours_out = ncont;
    for (i = 0; i < num; i++) {
        dbpost_oracle_dbg d; float ssid, score, arr[4][2], clip[4][2]; rrect box, ub;
        memset(&d, 0, sizeof d);
        d.is_hole = cs[i].is_hole; d.trig_x = cs[i].trig_x; d.trig_y = cs[i].trig_y; d.npts = cs[i].n;
        d.start_x = cs[i].pts[0].x; d.start_y = cs[i].pts[0].y;
        d.status = DBPO_OK;
        do {
            if (cs[i].n <= 2) { d.status = DBPO_SKIP_NPTS; break; }
            box = min_area_rect_i(cs[i].pts, cs[i].n);
            d.rect[0] = box.cx; d.rect[1] = box.cy; d.rect[2] = box.w; d.rect[3] = box.h; d.rect[4] = box.angle;
            get_mini_boxes(box, arr, &ssid);
            memcpy(d.minibox, arr, sizeof arr);
            if (ssid < min_size) { d.status = DBPO_SKIP_SSID; break; }
            score = box_score(&cs[i], pred, H, W, &d.npix);
            d.score = score;
            if (score < box_thresh) { d.status = DBPO_SKIP_SCORE; break; }
            ub = unclip_rect(arr, unclip_ratio, &d.distance, &d.npoly);
            d.urect[0] = ub.cx; d.urect[1] = ub.cy; d.urect[2] = ub.w; d.urect[3] = ub.h; d.urect[4] = ub.angle;
            if (ub.h < 1.001 && ub.w < 1.001) { d.status = DBPO_SKIP_UNCLIP; break; }
            get_mini_boxes(ub, clip, &ssid);
            if (ssid < min_size + 2) { d.status = DBPO_SKIP_SSID2; break; }
            for (j = 0; j < 4; j++) {
                if (g_use_padding_resize) {
                    float tx, ty;
                    padding_resize_point(clip[j][0], clip[j][1], src_w, src_h, H, &tx, &ty);
                    d.box[2 * j]     = (int)clampf(roundf(tx), 0, (float)src_w);
                    d.box[2 * j + 1] = (int)clampf(roundf(ty), 0, (float)src_h);
                } else {
                    d.box[2 * j]     = (int)clampf(roundf(clip[j][0] / (float)W * (float)src_w), 0, (float)src_w);
                    d.box[2 * j + 1] = (int)clampf(roundf(clip[j][1] / (float)H * (float)src_h), 0, (float)src_h);
                }
            }
            if (nboxes < max_boxes) memcpy(boxes_out + 8 * nboxes, d.box, 8 * sizeof(int));
            nboxes++;
        } while (0);
        if (dbg && i < dbg_cap) dbg[i] = d;
    }
    for (i = 0; i < ncont; i++) free(cs[i].pts);
    free(cs);
    return nboxes;
}

/* Contours only (tests of the GPU trace stage): flattens up to pts_cap points; returns #contours. */
int dbpost_oracle_contours(const uint8_t *bitmap, int H, int W, int *npts /*[cap]*/, int *is_hole, int *trig_xy,
                           int cap, int *pts_xy, int pts_cap) {
    int ncont = 0, i, k, off = 0;
    contour_t *cs = find_contours(bitmap, H, W, &ncont);
    for (i = 0; i < ncont; i++) {
        if (i < cap) {
            npts[i] = cs[i].n; is_hole[i] = cs[i].is_hole;
            trig_xy[2 * i] = cs[i].trig_x; trig_xy[2 * i + 1] = cs[i].trig_y;
            for (k = 0; k < cs[i].n; k++)
                if (off < pts_cap) { pts_xy[2 * off] = cs[i].pts[k].x; pts_xy[2 * off + 1] = cs[i].pts[k].y; off++; }
        }
        free(cs[i].pts);
    }
    free(cs);
    return ncont;
}

/* minAreaRect + GetMiniBoxes on float points (tests of the GPU geometry stage). */
void dbpost_oracle_min_area_rect(const float *pts_xy, int n, float *rect5, float *minibox8, float *ssid) {
    rrect r = min_area_rect_f((const fpt *)pts_xy, n);
    float arr[4][2];
    rect5[0] = r.cx; rect5[1] = r.cy; rect5[2] = r.w; rect5[3] = r.h; rect5[4] = r.angle;
    get_mini_boxes(r, arr, ssid);
    memcpy(minibox8, arr, sizeof arr);
}

/* P1 (db_postprocess.py:45-46): segmentation = pred > thresh, compared in float32. */
void dbpost_oracle_binarize(const float *pred, size_t n, float thresh, uint8_t *bitmap) {
    size_t i;
    for (i = 0; i < n; i++) bitmap[i] = pred[i] > thresh ? 1 : 0;
}
