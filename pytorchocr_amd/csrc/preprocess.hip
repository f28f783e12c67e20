// GPU pre-process next to the path: u8 image -> bilinear resize (cv2.resize INTER_LINEAR fixed-point
// semantics: 11-bit coefficients, half-pixel centres) -> /255 -> mean/std -> fp32 NHWC(4) network input, and the batched
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

__device__ __forceinline__ void lin_coef(int d, int sn, double scale, int *s0, int *s1, int *a0, int *a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= sn - 1) { f = 0.f; s = sn - 1; }
    *s0 = s; *s1 = min(s + 1, sn - 1);
    const int c1 = (int)rintf(f * 2048.f);
    *a1 = c1; *a0 = 2048 - c1;
}

__device__ __forceinline__ int gray_bgr(const uint8_t *p) {        // cv2 COLOR_BGR2GRAY, 15-bit fixed point
    return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
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
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        lin_coef(x, it.sw, (double)it.sw / it.rw, &x0, &x1, &ax0, &ax1);
        lin_coef(y, it.sh, (double)it.sh / it.rh, &y0, &y1, &by0, &by1);
        if (it.flip) {
            x0 = it.sw - 1 - x0; x1 = it.sw - 1 - x1;
            y0 = it.sh - 1 - y0; y1 = it.sh - 1 - y1;
        }
        const uint8_t *p00 = s + ((long)y0 * it.sw + x0) * 3, *p01 = s + ((long)y0 * it.sw + x1) * 3;
        const uint8_t *p10 = s + ((long)y1 * it.sw + x0) * 3, *p11 = s + ((long)y1 * it.sw + x1) * 3;
        const bool same = it.rh == it.sh && it.rw == it.sw;
        if (mode == 1) {
            int r;
            if (same) r = gray_bgr(p00);
            else {
                const long r0 = (long)gray_bgr(p00) * ax0 + (long)gray_bgr(p01) * ax1;
                const long r1 = (long)gray_bgr(p10) * ax0 + (long)gray_bgr(p11) * ax1;
                r = (int)((r0 * by0 + r1 * by1 + (1 << 21)) >> 22);
            }
            r = min(max(r, 0), 255);
            v[0] = ((float)r / 255.f - 0.5f) / 0.5f;
        } else {
            const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
            for (int c = 0; c < 3; c++) {
                const int sc = swap_rb ? 2 - c : c;
                int r;
                if (same) r = p00[sc];
                else {
                    const long r0 = (long)p00[sc] * ax0 + (long)p01[sc] * ax1;
                    const long r1 = (long)p10[sc] * ax0 + (long)p11[sc] * ax1;
                    r = (int)((r0 * by0 + r1 * by1 + (1 << 21)) >> 22);
                }
                r = min(max(r, 0), 255);
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

// cv2.warpPerspective(INTER_LINEAR, BORDER_REPLICATE) on the crop: source coordinates quantised to 1/32 pixel
__global__ __launch_bounds__(256) void warp_crops_kernel(const uint8_t *__restrict__ img, int H, int W, uint8_t *__restrict__ dst,
                                                         const WarpItem *__restrict__ items) {
    const WarpItem it = items[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= it.cw * it.ch) return;
    img += (long)it.img * H * W * 3;
    const int y = p / it.cw, x = p - y * it.cw;
    const double den0 = it.minv[6] * x + it.minv[7] * y + it.minv[8];
    const double den = den0 != 0 ? 1.0 / den0 : 0.0;
    const double fx = (it.minv[0] * x + it.minv[1] * y + it.minv[2]) * den;
    const double fy = (it.minv[3] * x + it.minv[4] * y + it.minv[5]) * den;
    const long X = (long)rint(fx * 32), Y = (long)rint(fy * 32);
    const int x0 = (int)(X >> 5), y0 = (int)(Y >> 5);
    const float ax = (float)(X & 31) / 32.f, ay = (float)(Y & 31) / 32.f;
    const int cx0 = min(max(x0, 0), it.cw - 1) + it.left, cx1 = min(max(x0 + 1, 0), it.cw - 1) + it.left;
    const int cy0 = min(max(y0, 0), it.ch - 1) + it.top, cy1 = min(max(y0 + 1, 0), it.ch - 1) + it.top;
    const uint8_t *p00 = img + ((long)cy0 * W + cx0) * 3, *p01 = img + ((long)cy0 * W + cx1) * 3;
    const uint8_t *p10 = img + ((long)cy1 * W + cx0) * 3, *p11 = img + ((long)cy1 * W + cx1) * 3;
    long o;
    if (it.rot90) o = ((long)(it.cw - 1 - x) * it.ch + y) * 3;      // rot90 ccw: out[cw-1-x][y] = in[y][x]
    else o = ((long)y * it.cw + x) * 3;
    for (int c = 0; c < 3; c++) {
        const float top = p00[c] * (1.f - ax) + p01[c] * ax;
        const float bot = p10[c] * (1.f - ax) + p11[c] * ax;
        const float r = rintf(top * (1.f - ay) + bot * ay);
        dst[it.dst_off + o + c] = (uint8_t)fminf(fmaxf(r, 0.f), 255.f);
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
