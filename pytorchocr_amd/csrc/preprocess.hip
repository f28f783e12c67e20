// GPU pre-process next to the path: u8 image -> bilinear resize (cv2.resize INTER_LINEAR in resize.cpp's u8
// arithmetic: 11-bit coefficient tables, HResizeLinear, the truncating VResizeLinear, the 2x2 area re-route) -> /255 -> mean/std -> fp32 NHWC(4) network input, and the batched
// perspective crop of run_ocr.  Replaces DetResizeForTest+ToTensor+Normalize (pytocr/data/imaug/operators.py:41-112,155-252),
// resize_norm_img (rec_img_aug.py:108-134, incl. BGR2GRAY and right zero padding), get_part_img (utils/utility.py:53-78), and is
// the MI355X counterpart of the reference's own CUDA normalisation kernel (deploy/trt_utils.py:43-52).  HBM-bound, one thread
// per destination pixel, all items of a batch in one launch through a descriptor array.
#include "common.h"

namespace ptocr {

struct PreItem {            // one image / crop
    long src_off;           // byte offset of the u8 HxWx3 (BGR) source inside d_src
    int sh, sw;             // source size
    int rh, rw;             // resized size (<= dh, dw); the rest of the destination is zero padding
    long dst_off;           // float offset of the destination f32[dh][dw][cpad] inside d_dst
    int dh, dw;
    int flip;               // 1: the source is read rotated by 180 degrees (cv2.rotate(ROTATE_180) of run_ocr.py:209-211 before the resize)
    int pad_;
};

// resize.cpp's per-axis table for INTER_LINEAR, 8-bit path: offset and the two 11-bit coefficients (both saturate_cast<short> of a
// float product).  clamp = the x axis (offset clamped, fraction zeroed); the y axis keeps its offset and the ROWS are clipped.
__device__ __forceinline__ void lin_coef(int d, int dn, int sn, bool clamp, int *s0, int *a0, int *a1) {
    const double scale = 1.0 / ((double)dn / (double)sn);        // hal::resize: scale = 1. / inv_scale
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (clamp) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= sn - 1) { f = 0.f; s = sn - 1; }
    }
    *s0 = s;
    *a0 = min(max((int)rintf((1.f - f) * 2048.f), -32768), 32767);
    *a1 = min(max((int)rintf(f * 2048.f), -32768), 32767);
}

__device__ __forceinline__ int gray_bgr(const uint8_t *p) {        // cv2 COLOR_BGR2GRAY of the 4.1.x line: CV_DESCALE(.., 14)
    return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14;
}

// HResizeLinear (S[sx] * a0 + S[sx + 1] * a1, scale 2^11) then the u8 VResizeLinear: two truncating products, (+ 2) >> 2
__device__ __forceinline__ int lin_mix(int v00, int v01, int v10, int v11, int ax0, int ax1, int by0, int by1) {
    const int r0 = v00 * ax0 + v01 * ax1, r1 = v10 * ax0 + v11 * ax1;
    return ((((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2) & 255;
}

// mode 0: 3 channels out (RGB order if swap_rb) normalised with mean/std; mode 1: 1 gray channel, (x/255 - 0.5)/0.5
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ src, float *__restrict__ dst, const PreItem *__restrict__ items,
                                                         int mode, int swap_rb, int cpad, float m0, float m1, float m2, float s0, float s1, float s2) {
    const PreItem it = items[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= it.dh * it.dw) return;
    const int y = p / it.dw, x = p - y * it.dw;
    float *o = dst + it.dst_off + (long)p * cpad;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y < it.rh && x < it.rw) {
        const uint8_t *s = src + it.src_off;
        const bool same = it.rh == it.sh && it.rw == it.sw;
        const bool area = it.sw == 2 * it.rw && it.sh == 2 * it.rh;      // INTER_LINEAR at exactly 2x2 runs as INTER_AREA (fast): (a+b+c+d+2)>>2
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        if (area) {
            x0 = 2 * x; x1 = x0 + 1; y0 = 2 * y; y1 = y0 + 1;
            ax0 = ax1 = by0 = by1 = 0;
        } else {
            lin_coef(x, it.rw, it.sw, true, &x0, &ax0, &ax1);
            x1 = min(x0 + 1, it.sw - 1);
            lin_coef(y, it.rh, it.sh, false, &y0, &by0, &by1);
            y1 = min(max(y0 + 1, 0), it.sh - 1);
            y0 = min(max(y0, 0), it.sh - 1);
        }
        if (it.flip) {
            x0 = it.sw - 1 - x0; x1 = it.sw - 1 - x1;
            y0 = it.sh - 1 - y0; y1 = it.sh - 1 - y1;
        }
        const uint8_t *p00 = s + ((long)y0 * it.sw + x0) * 3, *p01 = s + ((long)y0 * it.sw + x1) * 3;
        const uint8_t *p10 = s + ((long)y1 * it.sw + x0) * 3, *p11 = s + ((long)y1 * it.sw + x1) * 3;
        if (mode == 1) {
            int r;
            if (same) r = gray_bgr(p00);
            else if (area) r = (gray_bgr(p00) + gray_bgr(p01) + gray_bgr(p10) + gray_bgr(p11) + 2) >> 2;
            else r = lin_mix(gray_bgr(p00), gray_bgr(p01), gray_bgr(p10), gray_bgr(p11), ax0, ax1, by0, by1);
            v[0] = ((float)r / 255.f - 0.5f) / 0.5f;
        } else {
            const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
            for (int c = 0; c < 3; c++) {
                const int sc = swap_rb ? 2 - c : c;
                int r;
                if (same) r = p00[sc];
                else if (area) r = (p00[sc] + p01[sc] + p10[sc] + p11[sc] + 2) >> 2;
                else r = lin_mix(p00[sc], p01[sc], p10[sc], p11[sc], ax0, ax1, by0, by1);
                v[c] = ((float)r / 255.f - mean[c]) / sd[c];
            }
        }
    }
    for (int c = 0; c < cpad; c++) o[c] = c < 4 ? v[c] : 0.f;
}

struct WarpItem {           // one text box of the source image
    double minv[9];         // inverse perspective matrix (crop -> source-crop coordinates), row major
    int left, top;          // origin of the axis-aligned source crop
    int cw, ch;             // size of the source crop (clamp range) == size of the warped crop before rotation
    int rot90;              // 1: store rotated 90 degrees counter-clockwise (np.rot90(img, 1)) -> output is cw rows x ch cols
    int img;                // index of the source image inside d_img (a stack of equally sized u8 HxWx3 images; 0 for one image)
    long dst_off;           // byte offset of the u8 output crop (HxWx3) inside d_dst
};

// cv2.warpPerspective(INTER_LINEAR, BORDER_REPLICATE) on the crop, in imgwarp.cpp's arithmetic: WarpPerspectiveInvoker forms the source
// coordinate of destination pixel (xb + x1, y) as (M0 * xb + M1 * y + M2 + M0 * x1) * (32 / (M6 * xb + M7 * y + M8 + M6 * x1)) with xb the
// first column of its bw0-wide block, rounds it half to even to 1/32 pixel, and remapBilinear mixes the four pixels with the 15-bit
// integer weights of BilinearTab_i and (sum + 2^14) >> 15.  (double arithmetic: the library is built with -ffp-contract=off)
__global__ __launch_bounds__(256) void warp_crops_kernel(const uint8_t *__restrict__ img, int H, int W, uint8_t *__restrict__ dst,
                                                         const WarpItem *__restrict__ items) {
    const WarpItem it = items[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= it.cw * it.ch) return;
    img += (long)it.img * H * W * 3;
    const int y = p / it.cw, x = p - y * it.cw;
    const int bw0 = min(1024 / min(16, it.ch), it.cw);
    const int xb = (x / bw0) * bw0, x1 = x - xb;
    const double X0 = it.minv[0] * xb + it.minv[1] * y + it.minv[2];
    const double Y0 = it.minv[3] * xb + it.minv[4] * y + it.minv[5];
    const double W0 = it.minv[6] * xb + it.minv[7] * y + it.minv[8];
    double Wd = W0 + it.minv[6] * x1;
    Wd = Wd != 0 ? 32.0 / Wd : 0.0;
    const double fx = fmax(-2147483648.0, fmin(2147483647.0, (X0 + it.minv[0] * x1) * Wd));
    const double fy = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + it.minv[3] * x1) * Wd));
    const int X = (int)rint(fx), Y = (int)rint(fy);
    const int x0 = min(max(X >> 5, -32768), 32767), y0 = min(max(Y >> 5, -32768), 32767);
    const int ax = X & 31, ay = Y & 31;
    int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    if ((ax | ay) == 0) { w00 = 32767; w11 = 1; }          // the table entry that does not fit a short, as initInterTab2D leaves it
    const int cx0 = min(max(x0, 0), it.cw - 1) + it.left, cx1 = min(max(x0 + 1, 0), it.cw - 1) + it.left;
    const int cy0 = min(max(y0, 0), it.ch - 1) + it.top, cy1 = min(max(y0 + 1, 0), it.ch - 1) + it.top;
    const uint8_t *p00 = img + ((long)cy0 * W + cx0) * 3, *p01 = img + ((long)cy0 * W + cx1) * 3;
    const uint8_t *p10 = img + ((long)cy1 * W + cx0) * 3, *p11 = img + ((long)cy1 * W + cx1) * 3;
    long o;
    if (it.rot90) o = ((long)(it.cw - 1 - x) * it.ch + y) * 3;      // rot90 ccw: out[cw-1-x][y] = in[y][x]
    else o = ((long)y * it.cw + x) * 3;
    for (int c = 0; c < 3; c++) {
        const int r = (p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11 + (1 << 14)) >> 15;
        dst[it.dst_off + o + c] = (uint8_t)min(max(r, 0), 255);
    }
}

}  // namespace ptocr

using namespace ptocr;

extern "C" int ptocr_preprocess_u8_f32(const uint8_t *d_src, float *d_dst, const void *d_items, int n_items, int max_dst_pixels,
                                       int mode, int swap_rb, int cpad, const float *h_mean3, const float *h_std3, void *stream) {
    PT_CHECK(d_src && d_dst && d_items && n_items >= 1 && cpad >= 1 && (mode == 0 || mode == 1), "ptocr_preprocess_u8_f32: bad arguments");
    PT_CHECK(mode == 1 || (h_mean3 && h_std3 && cpad >= 3), "ptocr_preprocess_u8_f32: mode 0 needs mean/std and cpad >= 3");
    const float m[3] = {mode ? 0.f : h_mean3[0], mode ? 0.f : h_mean3[1], mode ? 0.f : h_mean3[2]};
    const float s[3] = {mode ? 1.f : h_std3[0], mode ? 1.f : h_std3[1], mode ? 1.f : h_std3[2]};
    for (int first = 0; first < n_items; first += 65535) {          // grid.y limit
        const int n = n_items - first < 65535 ? n_items - first : 65535;
        hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(max_dst_pixels, 256), n), dim3(256), 0, (hipStream_t)stream, d_src, d_dst,
                           (const PreItem *)d_items + first, mode, swap_rb, cpad, m[0], m[1], m[2], s[0], s[1], s[2]);
    }
    return launch_ok("preprocess_kernel");
}

extern "C" int ptocr_warp_crops_u8(const uint8_t *d_img, int H, int W, uint8_t *d_dst, const void *d_items, int n_items,
                                   int max_crop_pixels, void *stream) {
    PT_CHECK(d_img && d_dst && d_items && n_items >= 1, "ptocr_warp_crops_u8: bad arguments");
    for (int first = 0; first < n_items; first += 65535) {          // grid.y limit
        const int n = n_items - first < 65535 ? n_items - first : 65535;
        hipLaunchKernelGGL(warp_crops_kernel, dim3(cdiv(max_crop_pixels, 256), n), dim3(256), 0, (hipStream_t)stream, d_img, H, W, d_dst,
                           (const WarpItem *)d_items + first);
    }
    return launch_ok("warp_crops_kernel");
}
