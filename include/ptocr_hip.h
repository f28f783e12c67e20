/*
 * ptocr_hip.h -- C ABI of libptocr_hip.so, the MI355X (gfx950) implementation of the PyTorchOCR
 * detection + recognition inference hot path.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (HBM) owned by the caller; h_* is host memory.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is stream-ordered, no call
 *     synchronises the device unless its comment says so.
 *   - return value 0 = ok, non-zero = error; ptocr_last_error() gives the message of the last failing call
 *     of the calling thread.  No call falls back to the CPU.
 *   - activations are fp32 NHWC in HBM (channel-contiguous, 16-byte aligned), probability maps are
 *     f32[N,1,H,W] == NHWC with C=1, exactly the tensor the reference model returns under key "maps".
 *
 * What each entry point replaces in the reference (file:line under /root/reference):
 *   ptocr_conv2d_f32 / ptocr_conv3x3_wino_f32 (3x3 s1: Winograd) / ptocr_conv7x7s2_stem_f32 (ResNet stem) /
 *   ptocr_conv1x1_k64_f32, ptocr_conv1x1_small_k_f32 (1x1 with <= 64 inputs) / ptocr_conv3x3_small_relu_pool_f32 (CRNN conv0 +
 *   pool) / ptocr_maxpool2d_f32 / ptocr_nchw_to_nhwc_f32 / ptocr_convt2x2_sigmoid_f32 / ptocr_db_head_tail_f32 /
 *   ptocr_dwconv_f32 / ptocr_se_scale_f32
 *       the ATen conv / BN / ReLU / Hardswish / max_pool / conv_transpose / sigmoid calls issued by
 *       pytocr/modeling/backbones/det_resnet.py:66-82,282-309, det_mobilenet_v3.py:38-151, necks/fpn.py:102-134,
 *       heads/det_db_head.py:9-17,47-50, backbones/rec_vgg.py:78-120 (BN folded into weights/bias at load).  The specialised
 *       entry points compute the same layers as ptocr_conv2d_f32 (same results within fp32 rounding) for the shapes where a
 *       dedicated kernel is faster; the Python host side picks them per layer.
 *   ptocr_asf_scale_channel_spatial_f32
 *       ScaleChannelSpatialAttention.forward + the re-weighted concat of pytocr/modeling/necks/asf.py:63-75,155-162 (DB++).
 *   ptocr_preprocess_u8_f32, ptocr_warp_crops_u8
 *       cv2.resize / to_tensor / normalize of pytocr/data/imaug/operators.py:41-112,155-252 and rec_img_aug.py:108-134
 *       (the reference's own CUDA analogue: deploy/trt_utils.py:43-52); get_part_img of pytocr/utils/utility.py:53-78.
 *   ptocr_db_postprocess
 *       pybind11 `db_postprocess.db_postprocess(pred, bitmap, box_thresh, det_db_unclip_ratio, src_w, src_h,
 *       use_padding_resize)` = DBProcess, pytocr/postprocess/db_postprocess_fast/src/db_postprocess.cpp:319-370,
 *       plus the threshold of pytocr/postprocess/db_postprocess.py:45-46, batched over N images.
 *   ptocr_ctc_greedy_f32
 *       preds.argmax(axis=2) / preds.max(axis=2) of pytocr/postprocess/rec_postprocess.py:80-84.
 *   ptocr_lstm_bidir_f32, ptocr_linear_f32, ptocr_softmax_rows_f32
 *       nn.LSTM(bidirectional) / nn.Linear / F.softmax of pytocr/modeling/necks/rnn.py:18-36, heads/rec_ctc_head.py:17-36.
 */
#ifndef PTOCR_HIP_H
#define PTOCR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char *ptocr_last_error(void);
int ptocr_version(void);
/* digest of the compile flags the library was built with (pytorchocr_amd/build.py); the Python binding refuses a library whose tag is
 * not the default build's unless PTOCR_EXTRA_HIPCC_FLAGS names the experiment flags it carries */
const char *ptocr_build_tag(void);
/* Device memory the library OWNS (the post-process workspaces of ptocr_dbpost_create, the exchange buffers of the split LSTM) comes from
 * this pair, hipMalloc / hipFree by default: a host framework installs its own pool here (torch's caching allocator, a guard-page
 * allocator in tests).  alloc_fn returns 0 and a device pointer aligned to 256 bytes; free_fn may be called from any thread and must not
 * return before the device is done with the range (hipFree's contract).  Refused while buffers of the current allocator are alive;
 * (NULL, NULL) restores the default.  ptocr_live_allocations: buffers the library holds right now. */
typedef int (*ptocr_alloc_fn)(void **d_ptr, size_t bytes);
typedef int (*ptocr_free_fn)(void *d_ptr);
int ptocr_set_allocator(ptocr_alloc_fn alloc_fn, ptocr_free_fn free_fn);
long ptocr_live_allocations(void);
/* fills name (<=255 chars) with the gcnArchName of device `dev`, returns 0 or error */
int ptocr_device_arch(int dev, char *name);

/* ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate) ------------------- */
enum { PTOCR_RES_NONE = 0, PTOCR_RES_ADD_PRE_RELU = 1, PTOCR_RES_ADD_UP2_POST_RELU = 2 };

typedef struct {
    int N, H, W, Cin;          /* input  f32[N,H,W,Cin]; Cin % 32 == 0, or Cin == 4 (zero-padded 1/3-channel images) */
    int Cout, KH, KW;          /* weights packed f32[Cout][Kpad], K index = (kh*KW + kw)*Cin + ci, Kpad = roundup(KH*KW*Cin, 32) */
    int stride, pad_h, pad_w;
    int Ho, Wo;                /* output spatial size (before out_up / convt expansion) */
    int relu;                  /* epilogue activation: 0 none, 1 ReLU, 2 Hardswish */
    int res_mode;              /* PTOCR_RES_*: d_res is f32[N,Ho,Wo,Cout] (PRE_RELU) or f32[N,Ho/2,Wo/2,Cout] (UP2_POST_RELU) */
    int out_up;                /* >= 1: every output pixel is stored to an out_up x out_up block (nearest upsample) */
    int out_ldc, out_coff;     /* output tensor channel stride / channel offset (concat-in-place); ldc >= coff + Cout */
    int convt2x2;              /* 1: ConvTranspose2d k=2 s=2: Cout here = 4*Co, column (a*2+b)*Co + co goes to pixel (2y+a, 2x+b) */
    int cout_store;            /* 0 = all; else only columns < cout_store are written (weights padded to Cout % 64 == 0) */
    int res_ldc;               /* channel stride of d_res (0 = Cout) */
} ptocr_conv_desc;

int ptocr_conv2d_f32(const ptocr_conv_desc *d, const float *d_x, const float *d_w, const float *d_bias,
                     const float *d_res, float *d_y, void *stream);

/* 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2,3x3) on fp32 MFMA (2.25x fewer multiplies, fp32 accuracy).
 * d_u: host-transformed weights U = G g G^T (BN folded), packed f32[Cout/64][Cin/4][16][64][4]; Cin % 16 == 0, Cout % 64 == 0
 * (zero-pad the weights and the bias of a narrower layer; cout_store = channels actually written, a multiple of 4, 0 = Cout).
 * Epilogue: bias, optional pre-ReLU residual (d_res f32[N,H,W,res_ldc]), optional ReLU, store into channels
 * [out_coff, out_coff+Cout) of a tensor with channel stride out_ldc; up > 1 (<= 8, no residual) stores every output pixel
 * to an up x up block of y f32[N,H*up,W*up,out_ldc] (nearest upsample, fpn.py:125-133). */
int ptocr_conv3x3_wino_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                           int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                           int out_ldc, int out_coff, int up, void *stream);
/* The same layers by Winograd F(4x4,3x3) (4x fewer multiplies than direct, 1.78x fewer than F(2x2); maps move by ~1e-6): same
 * arguments and epilogue; d_u: U = G6 g G6^T (6x6 per weight pair, fp64 on the host), packed
 * f32[Cout/64][Cin/4][12][3][64][4]: "wave" w owns the frequencies xi = 3w + e (xi = 6 i + j), lane l = 32 h + n holds
 * {U[xi][c0+2h][n], U[xi][c0+2h+1][n], U[xi][c0+2h][32+n], U[xi][c0+2h+1][32+n]} with c0 = 4 * chunk and n relative to the 64
 * output channels of the block.  ptocr_conv3x3_wino4_patches: workgroups per 64 output channels the kernel needs for
 * N x H x W (to choose between the two forms per layer). */
int ptocr_conv3x3_wino4_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                            int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                            int out_ldc, int out_coff, int up, void *stream);
long ptocr_conv3x3_wino4_patches(int N, int H, int W);
/* CRNN conv1 + relu1 + pooling1 (reference rec_vgg.py:28-35: Conv2d 3x3 + ReLU, MaxPool2d(2, 2)) in one launch: the Winograd F(4x4) kernel's
 * epilogue takes the 2x2 maxima of each 4x4 output tile; the full-resolution tensor is never written.  d_u as for ptocr_conv3x3_wino4_f32;
 * y f32[N, H/2, W/2, out_ldc] (cout_store channels written); H and W even.  Bit-identical to the conv followed by ptocr_maxpool2d_f32. */
int ptocr_conv3x3_wino4_pool2_f32(const float *d_x, const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cin,
                                  int Cout, int cout_store, int out_ldc, void *stream);

/* The same two operations on the round-5 re-cut of the kernel (conv_wino4r.hip: a wave owns one row of the 6x6 frequencies, persistent
 * workgroups, double-buffered output transform; same multiplies, sums of the output transform in another order).  d_u: the same U, packed
 * f32[Cout/64][Cin/4][2 wh][6 wi][3 q][2 kh][32 n][2 jj][2 t] = U[xi = 6 wi + 2 q + jj][cin = 4 chunk + 2 kh + t][cout = 64 cb + 32 wh + n]. */
int ptocr_conv3x3_wino4r_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                             int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                             int out_ldc, int out_coff, int up, void *stream);
int ptocr_conv3x3_wino4r_pool2_f32(const float *d_x, const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cin,
                                   int Cout, int cout_store, int out_ldc, void *stream);
void ptocr_wino4r_set_timing_buffer(void *d_buf);  /* 4 clock samples per patch: start, main loop start, main loop end, end */
/* ptocr_conv3x3_wino4r_f32 on an input that exists only as a PYRAMID: replaces interpolate + cat of the reference's FPN
 * (pytocr/modeling/necks/fpn.py:118-131: fuse = cat(up8(p5), up4(p4), up2(p3), p2)) as the input of DBHead's first conv
 * (pytocr/modeling/heads/det_db_head.py:9-17).  Four planes of 64 channels in ONE allocation d_pyr of pyr_floats floats: plane j =
 * channels 64 j .. 64 j + 63 of the virtual f32[N,H,W,256] input, stored as f32[N, H >> shift[j], W >> shift[j], 64] at float offset
 * off[j] (off, shift: HOST arrays of four; shift 0..3 must divide H and W); the kernel reads pixel (y >> shift, x >> shift) of a plane
 * for pixel (y, x).  Result bit-identical to ptocr_conv3x3_wino4r_f32 on the materialised concat; no residual, no upsample. */
int ptocr_conv3x3_wino4r_pyramid_f32(const float *d_pyr, const long long *off, const int *shift, long long pyr_floats,
                                     const float *d_u, const float *d_bias, float *d_y, int N, int H, int W, int Cout,
                                     int cout_store, int relu, int out_ldc, int out_coff, void *stream);

/* Experiment, not the fp32 path (the host enables it with PTOCR_WINO_SPLIT=1; off by default): ptocr_conv3x3_wino4_f32 with
 * two-piece bf16 operands on the bf16 matrix pipe, fp32 accumulate -- x = h + m, h = bf16(x), m = bf16(x - h); a b becomes
 * (a_h + a_m)(b_h + b_m), 16 mantissa bits per operand.  d_u: the packing above with every fp32 U replaced by the dword
 * bf16(U) | bf16(U - bf16(U)) << 16.  Everything else as ptocr_conv3x3_wino4_f32. */
int ptocr_conv3x3_wino4_split_f32(const float *d_x, const float *d_u, const float *d_bias, const float *d_res, float *d_y,
                                  int N, int H, int W, int Cin, int Cout, int cout_store, int relu, int res_mode, int res_ldc,
                                  int out_ldc, int out_coff, int up, void *stream);
/* ResNet stem: conv 7x7 / stride 2 / pad 3 of an RGB image stored as f32[N,H,W,4] (4th channel ignored) -> f32[N,Ho,Wo,64],
 * Ho = (H-1)/2+1, Wo = (W-1)/2+1, + bias (folded BN) + optional ReLU (det_resnet.py:193-196).  d_w: f32[7][22][64],
 * w[ky][kx*3 + c][cout], row [ky][21] all zero (K runs over 7 x 22 = 154 instead of the generic kernel's 7*7*4 = 196). */
int ptocr_conv7x7s2_stem_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                             int relu, void *stream);
/* the same layer reading the model's input tensor f32[N,3,H,W] directly (saves the NCHW -> NHWC boundary pass) */
int ptocr_conv7x7s2_stem_nchw_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                  int relu, void *stream);
/* The stem WITH its max pool: conv 7x7 / s2 / p3 + folded BN + ReLU + MaxPool2d(3, 2, 1) (det_resnet.py:193-197,284-287) in one
 * kernel -> f32[N,Hp,Wp,64], Hp = (Ho-1)/2+1, Wp = (Wo-1)/2+1; the full-resolution stem output is never written.  Bit-identical to
 * ptocr_conv7x7s2_stem_f32(relu = 1) followed by ptocr_maxpool2d_f32(3, 3, 2, 2, 1, 1).  _nchw: input f32[N,3,H,W]. */
int ptocr_conv7x7s2_stem_relu_pool_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                       void *stream);
int ptocr_conv7x7s2_stem_relu_pool_nchw_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                            void *stream);
/* Pointwise convolution with 64 input channels (the FPN lateral in2, fpn.py:46-51): f32[N,H,W,64] x W[64][Cout] (k-major, BN
 * folded; Cout a multiple of 32, <= 256) + bias + optional ReLU; res_up2 = 1 adds d_res f32[N,H/2,W/2,Cout] nearest-upsampled
 * x2 AFTER the activation (fpn.py:133-134).  Output channels [out_coff, out_coff+Cout) of a tensor with channel stride out_ldc. */
int ptocr_conv1x1_k64_f32(const float *d_x, const float *d_w, const float *d_bias, const float *d_res, float *d_y,
                          int N, int H, int W, int Cout, int relu, int res_up2, int out_ldc, int out_coff, void *stream);
/* The same kernel for Cin = 32, 64 or 128 (128: Cout a multiple of 128, half of the output channels per workgroup -- the FPN lateral
 * in3) and act = 0 none / 1 ReLU / 2 Hardswish (MobileNetV3's 1x1 layers on the large maps,
 * det_mobilenet_v3.py:38-61; pad narrower layers with zero rows / columns). */
int ptocr_conv1x1_small_k_f32(const float *d_x, const float *d_w, const float *d_bias, const float *d_res, float *d_y,
                              int N, int H, int W, int Cin, int Cout, int act, int res_up2, int out_ldc, int out_coff, void *stream);
/* CRNN conv0 + relu0 + pooling0 fused (rec_vgg.py:78-88): y f32[N,H/2,W/2,64] = maxpool2x2(relu(conv3x3/s1/p1(x) + bias)) for an
 * input with 1..4 channels stored as f32[N,H,W,4]; d_w f32[Cin*9][64], row (ci*3 + ky)*3 + kx (BN folded). */
int ptocr_conv3x3_small_relu_pool_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W,
                                      int Cin, void *stream);
/* Measurement hook (no reference counterpart): d_buf = device u64[4 * workgroups] receives s_memtime samples (start, main
 * loop start, main loop end, end) from every Winograd workgroup launched afterwards; NULL switches the probe off. */
void ptocr_wino_set_timing_buffer(void *d_buf);
void ptocr_wino4_set_timing_buffer(void *d_buf);   /* the same for ptocr_conv3x3_wino4_f32 */

/* f32[N,C,H,W] -> f32[N,H,W,Cpad] (channels >= C zero-filled; Cpad % 4 == 0) */
int ptocr_nchw_to_nhwc_f32(const float *d_x, float *d_y, int N, int C, int H, int W, int Cpad, void *stream);
/* f32[N,H,W,C] -> f32[N,C,H,W] */
int ptocr_nhwc_to_nchw_f32(const float *d_x, float *d_y, int N, int C, int H, int W, void *stream);
/* NHWC max pool, -inf padding like torch.nn.MaxPool2d; C % 4 == 0 */
int ptocr_maxpool2d_f32(const float *d_x, float *d_y, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                        int ph, int pw, int Ho, int Wo, void *stream);
/* DB head tail: ConvTranspose2d(C->1, k=2, s=2, bias) + sigmoid.  d_x f32[N,H,W,C] (C % 4 == 0, C <= 256),
 * d_w f32[4][C] (index a*2+b), bias scalar -> d_maps f32[N,2H,2W]. */
int ptocr_convt2x2_sigmoid_f32(const float *d_x, const float *d_w, float bias, float *d_maps, int N, int H, int W,
                               int C, void *stream);

/* Fused DB head tail (det_db_head.py:13-17): ConvT(64,64,2,2)+BN+ReLU -> ConvT(64,1,2,2)+bias -> sigmoid.
 * d_x f32[N,H,W,64]; d_w1 f32[256][64] (row (a*2+b)*64+co, BN folded), d_b1 f32[256]; d_w2 f32[4][64], b2 -> d_maps f32[N,4H,4W]. */
int ptocr_db_head_tail_f32(const float *d_x, const float *d_w1, const float *d_b1, const float *d_w2, float b2, float *d_maps,
                           int N, int H, int W, int C, void *stream);
/* Depthwise conv (groups == C) + folded BN bias + activation (0 none / 1 ReLU / 2 Hardswish), NHWC, C % 4 == 0.
 * d_w f32[k*k][C] (tap-major), k in {2, 3, 5}, pad = (k-1)/2 (none for k = 2).  (MobileNetV3 InvertedResidual.conv2,
 * det_mobilenet_v3.py:123-126; the depthwise layers of the CRNN's VGG v2, rec_vgg.py:62-76) */
int ptocr_dwconv_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, int C, int k,
                     int stride, int act, void *stream);
/* the same with separate vertical / horizontal strides (the recognition-style MobileNetV3 of the direction classifier strides its
 * depthwise convs by (s, 1): pytocr/modeling/backbones/rec_mobilenet_v3.py:128-131) */
int ptocr_dwconv2_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int H, int W, int C, int k,
                      int stride_h, int stride_w, int act, void *stream);
/* Direction-classifier head behind that backbone's AvgPool2d(2, 2) (rec_mobilenet_v3.py:221,266 + heads/cls_head.py:16-29):
 * mean over the 2x2 blocks that fit, Linear(C -> K), softmax.  d_x f32[N,H,W,C], d_w f32[K][C], d_b f32[K], K <= 64 -> d_y f32[N][K]. */
int ptocr_cls_head_f32(const float *d_x, const float *d_w, const float *d_b, float *d_y, int N, int H, int W, int C, int K, void *stream);
/* Squeeze-Excitation (det_mobilenet_v3.py:67-85): x[n,:,:,c] *= hardsigmoid(fc2(relu(fc1(mean_hw(x))))) in place.
 * d_w1 f32[S][C], d_b1 f32[S], d_w2 f32[C][S], d_b2 f32[C]; d_work: N*(ceil(H*W/2048)+1)*C floats. C <= 1024, S <= 256. */
int ptocr_se_scale_f32(float *d_x, const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2, float *d_work,
                       int N, int H, int W, int C, int S, void *stream);

/* DB++ Adaptive Scale Fusion, attention_type "scale_channel_spatial" (asf.py:32-75,146-162), after its 3x3 conv (+bias) has
 * produced d_y f32[N,H,W,64] with ptocr_conv2d_f32.  Scales d_fuse f32[N,H,W,256] IN PLACE: channels [64i,64i+64) *= score_i.
 * d_w_cw1 f32[16][64], d_w_cw2 f32[64][16] (channel_wise.1/.3), d_w_sp3 f32[9], w_sp1 (spatial_wise.0/.2), d_w_att f32[4][64]
 * (attention_wise.0); d_work: ptocr_asf_work_floats(N,H,W) floats of scratch. */
int ptocr_asf_scale_channel_spatial_f32(const float *d_y, float *d_fuse, const float *d_w_cw1, const float *d_w_cw2,
                                        const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work,
                                        int N, int H, int W, void *stream);
long ptocr_asf_work_floats(int N, int H, int W);
/* attention_type "scale_spatial" (asf.py:78-107, 146-162): as above without the channel gate. */
int ptocr_asf_scale_spatial_f32(const float *d_y, float *d_fuse, const float *d_w_sp3, float w_sp1, const float *d_w_att, float *d_work,
                                int N, int H, int W, void *stream);
/* attention_type "scale_channel" (asf.py:9-29, 146-162): global average of y -> fc1 (d_w1 f32[32][64], d_b1 f32[32]: its BatchNorm
 * folded in) -> ReLU -> fc2 (d_w2 f32[4][32]) -> softmax over the four levels; fuse[n, :, :, 64 i .. 64 i + 63] *= score[n][i]
 * (the reference interpolates the 1x1 score map bilinearly to the map size: a constant). */
int ptocr_asf_scale_channel_f32(const float *d_y, float *d_fuse, const float *d_w1, const float *d_b1, const float *d_w2, float *d_work,
                                int N, int H, int W, void *stream);
/* The three ASF forms on a PYRAMID: the four 64-channel features are read from their own-resolution planes (d_pyr, off[4], shift[4],
 * pyr_floats exactly as ptocr_conv3x3_wino4r_pyramid_f32, which computes d_y from the same planes) and the re-weighted concat is
 * written to d_out f32[N,H,W,256] -- cat(up8(p5), up4(p4), up2(p3), p2) of pytocr/modeling/necks/fpn.py:118-131 is never
 * materialised; bit-identical to the in-place forms on the materialised concat (pytocr/modeling/necks/asf.py:146-162).
 * type 0 scale_channel_spatial: d_w_a = cw1, d_w_b = cw2, d_w_sp3, w_sp1, d_w_att; 1 scale_spatial: d_w_sp3, w_sp1, d_w_att
 * (d_w_a / d_w_b unused); 2 scale_channel: d_w_a = w1 (BN folded), d_w_b = b1, d_w_att = w2 (d_w_sp3 unused). */
int ptocr_asf_pyramid_f32(int type, const float *d_y, const float *d_pyr, const long long *off, const int *shift, long long pyr_floats,
                          float *d_out, const float *d_w_a, const float *d_w_b, const float *d_w_sp3, float w_sp1, const float *d_w_att,
                          float *d_work, int N, int H, int W, void *stream);

/* ---- bf16 inference path of the MobileNetV3 detector (BASELINE configs[3]) ---------------------------------------------
 * Activations are bf16 NHWC with the channel count padded to a multiple of 16 (padding channels hold zeros); weights bf16,
 * K-contiguous rows, BatchNorm folded; accumulation, bias, activation and every non-GEMM kernel in fp32.  Same reference
 * layers as the fp32 entry points above (det_mobilenet_v3.py:38-151, fpn.py:102-134, det_db_head.py:9-17).  `void *` = bf16. */
/* 1x1 conv: y[.., coff:coff+cstore] = act(x @ w^T + bias (+ res)) ; w bf16[Cout_pad][Cin] (Cout_pad % 32 == 0, zero rows beyond the
 * layer's channels), d_scale (optional) f32[N][Cin]: Squeeze-Excitation gate multiplied into the INPUT; res_mode 1: + d_res
 * (same geometry, channel stride res_ldc) before the activation; 2: + d_res[N,H/2,W/2] nearest-upsampled x2 after it (fpn.py:133). */
int ptocr_pwconv_bf16(const void *d_x, const void *d_w, const float *d_bias, const void *d_res, const float *d_scale, void *d_y,
                      int N, int H, int W, int Cin, int Cout_pad, int cstore, int act, int res_mode, int res_ldc, int out_ldc,
                      int out_coff, void *stream);
/* 3x3 / s1 / p1 conv with <= 32 outputs: w bf16[32][9 * Cin] (k = (ky*3 + kx) * Cin + ci); the result is stored to an out_up x
 * out_up block per pixel (nearest upsample) into channels [coff, coff + cstore) of a tensor with channel stride out_ldc. */
int ptocr_conv3x3_bf16(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int N, int H, int W, int Cin, int cstore,
                       int act, int out_up, int out_ldc, int out_coff, void *stream);

/* The DB head's first conv (reference det_db_head.py:10-12: Conv2d(in, in/4, 3, padding=1) + BN + ReLU) reading the FPN output
 * (reference fpn.py:96-100: torch.cat([out5 .. out2], 1)) as FOUR planes bf16[4][N,H,W,24] instead of one [N,H,W,96] concat, so that each
 * smoothing conv writes whole cache lines.  w bf16[32][9*96] / bias f32[32] as for ptocr_conv3x3_bf16 (channel order of the concat);
 * y bf16[N,H,W,out_ldc], cstore channels written.  Bit-identical to ptocr_conv3x3_bf16 on the concatenated tensor. */
int ptocr_conv3x3_planes_bf16(const void *d_x, const void *d_w, const float *d_bias, void *d_y, int N, int H, int W, int cstore, int act,
                              int out_ldc, void *stream);

/* The FPN lateral fused into the smoothing conv of the largest level (reference fpn.py:102-131: out2 = in2(c2) + upsample(out3), then
 * p2 = out2_conv(out2)): y = conv3x3(relu(bn(conv1x1(x2))) + nearest_x2(td)), the intermediate never written.
 * x2 bf16[N,H,W,16] (input channels padded to 16), wl bf16[96][16] + bl f32[96] (the lateral, BN folded), td bf16[N,H/2,W/2,td_ldc]
 * (>= 96 channels), w3 bf16[32][9*96] + b3 f32[32] (the smoothing conv); y, cstore, act, out_up, out_ldc, out_coff as in
 * ptocr_conv3x3_bf16.  Bit-identical to ptocr_pwconv_bf16 (res_mode 2) followed by ptocr_conv3x3_bf16. */
int ptocr_conv3x3_lat_bf16(const void *d_x2, const void *d_wl, const float *d_bl, const void *d_td, int td_ldc, const void *d_w3,
                           const float *d_b3, void *d_y, int N, int H, int W, int cstore, int act, int out_up, int out_ldc,
                           int out_coff, void *stream);
/* depthwise k x k conv + bias + activation; d_partial (optional) f32[N][nblk][C], nblk = ptocr_dwconv_bf16_nblk(...): per-chunk
 * channel sums of the activated output for the Squeeze-Excitation pool (fixed summation order: deterministic). */
int ptocr_dwconv_bf16(const void *d_x, const float *d_w, const float *d_bias, void *d_y, float *d_partial, int N, int H, int W,
                      int C, int k, int stride, int act, void *stream);
/* MobileNetV3 inverted residual, first two stages of the block in one launch (reference det_mobilenet_v3.py:106-151: conv1 1x1 + BN +
 * act, conv2 depthwise 3x3 / stride 2 + BN + act; no SE): the expanded tensor is never written.  x bf16[N,H,W,16] (input channels padded
 * to 16); we bf16[C32][16] + be f32[C32], C32 = C rounded up to 32 (expansion, BN folded, padding rows zero); wd f32[9][C] tap-major + bd f32[C] (depthwise, BN
 * folded); y bf16[N,(H-1)/2+1,(W-1)/2+1,C]; C a multiple of 16, <= 96; act_e == act_d in {1 ReLU, 2 Hardswish}.  Bit-identical to
 * ptocr_pwconv_bf16 followed by ptocr_dwconv_bf16. */
int ptocr_expand_dw3x3s2_bf16(const void *d_x, const void *d_we, const float *d_be, const float *d_wd, const float *d_bd, void *d_y,
                              int N, int H, int W, int C, int act_e, int act_d, void *stream);

int ptocr_dwconv_bf16_nblk(int N, int H, int W, int k, int stride);
/* SE gate from those sums: d_scale f32[N][C] = hardsigmoid(fc2(relu(fc1(sum / HW)))) (det_mobilenet_v3.py:76-85) */
int ptocr_se_fc_f32(const float *d_partial, const float *d_w1, const float *d_b1, const float *d_w2, const float *d_b2,
                    float *d_scale, int N, int HW, int C, int S, int nblk, void *stream);
/* the same gate from transposed weights d_w1t f32[C][S], d_w2t f32[S][C] (every load coalesced; what the bf16 path calls) */
int ptocr_se_fc_t_f32(const float *d_partial, const float *d_w1t, const float *d_b1, const float *d_w2t, const float *d_b2,
                      float *d_scale, int N, int HW, int C, int S, int nblk, void *stream);
/* the same gate for wide layers (what the bf16 path calls for C >= 256): eight blocks per image in two launches instead of one block
 * per image; d_hidden f32[N][S] is the caller's workspace.  Same result up to fp32 summation order. */
int ptocr_se_fc_split_f32(const float *d_partial, const float *d_w1t, const float *d_b1, const float *d_w2t, const float *d_b2,
                          float *d_hidden, float *d_scale, int N, int HW, int C, int S, int nblk, void *stream);
/* stem: conv 3x3 / s2 / p1 of the model input f32[N,3,H,W] -> bf16[N,Ho,Wo,16]; d_w f32[27][16] (row (c*3 + ky)*3 + kx), BN folded */
int ptocr_stem3x3s2_bf16(const float *d_x, const float *d_w, const float *d_bias, void *d_y, int N, int H, int W, int act, void *stream);
/* DB head tail for C = 24: ConvT(C,C,2,2)+BN+ReLU -> ConvT(C,1,2,2)+bias -> sigmoid; d_x bf16[N,H,W,ldc] -> d_maps f32[N,4H,4W];
 * d_w1 f32[4][C][C] ((a*2+b), ci, co), d_b1 f32[C], d_w2 f32[4][C] ((a*2+b), co). */
int ptocr_db_head_tail_bf16(const void *d_x, const float *d_w1, const float *d_b1, const float *d_w2, float b2, float *d_maps,
                            int N, int H, int W, int C, int ldc, void *stream);

/* ---- DB post-process ------------------------------------------------------------------------------------ */
typedef struct ptocr_dbpost *ptocr_dbpost_t;

/* Workspace for batches of up to max_n maps of up to max_h x max_w (device buffers are allocated here, once). */
int ptocr_dbpost_create(ptocr_dbpost_t *out, int max_n, int max_h, int max_w);
int ptocr_dbpost_destroy(ptocr_dbpost_t h);

/* d_maps f32[N,H,W] probability maps (device).  h_src_wh int[N][2] = (src_w, src_h) per image.
 * If d_bitmap != NULL it is used as the u8[N,H,W] segmentation (caller-made, e.g. dilated); otherwise the
 * segmentation is pred > thresh computed on the device.
 * Output (host): h_boxes int16[N][max_boxes][4][2] in the reference's order, h_counts int32[N] (boxes per
 * image), h_flags int32[N] (bit 0: unused since round 2 -- ClipperOffset::Execute's union is reproduced for every
 * candidate; bit 1: score within 1e-6 of box_thresh resolved by the exact raster-order re-summation; bit 2: internal capacity exceeded -> the call fails with an error).
 * Synchronises `stream` before returning (the boxes are host data, like the reference's return value). */
int ptocr_db_postprocess(ptocr_dbpost_t h, const float *d_maps, const uint8_t *d_bitmap, int N, int H, int W,
                         float thresh, float box_thresh, float unclip_ratio, const int *h_src_wh,
                         int use_padding_resize, int16_t *h_boxes, int max_boxes, int32_t *h_counts,
                         int32_t *h_flags, void *stream);

/* Same with the two remaining options of the reference: use_dilation = cv2.dilate(mask, 2x2 ones) before the extraction
 * (db_postprocess.py:52-55), use_padding_resize = map boxes back through get_affine_transform / transform_preds
 * (db_postprocess.cpp:111-145,289-301; needs H == W). */
int ptocr_db_postprocess_ex(ptocr_dbpost_t h, const float *d_maps, const uint8_t *d_bitmap, int N, int H, int W,
                            float thresh, float box_thresh, float unclip_ratio, const int *h_src_wh,
                            int use_padding_resize, int use_dilation, int16_t *h_boxes, int max_boxes, int32_t *h_counts,
                            int32_t *h_flags, void *stream);

/* Measurement hook (no reference counterpart): device time in ms, by HIP events on the call's own stream, from the first to
 * the last kernel of the LAST ptocr_db_postprocess call on this workspace (the call has synchronised that stream). */
int ptocr_dbpost_last_device_ms(ptocr_dbpost_t h, float *ms);

/* Inspection hook for the parity tests: per-candidate records of the LAST ptocr_db_postprocess call for image `img`
 * (synchronises the device).  h_total: number of border starts found -- exact below 1000; for an image with more, the
 * count of the bottom strip that already holds the 1000 used ones (>= 1000, the rest of the image is not labelled).  h_results: 1000 x {int status; int box[8]; float score; float rect[5]; int npix;
 * float distance;}  h_cands: 1000 x {int trigger_pixel; int is_hole;}  h_info: 1000 x {int npts; int off;
 * short xmin, xmax, ymin, ymax;}  status: 0 box, 1 <=2 points, 2 ssid<3, 3 score<box_thresh, 4 unclip<1.001, 5 ssid<5. */
int ptocr_dbpost_debug_results(ptocr_dbpost_t h, int img, int32_t *h_total, void *h_results, void *h_cands, void *h_info);

/* Second inspection hook: the border STATES of candidate k of image img of the last call (one word per state: x | y << 11 |
 * s_out << 26 | s_in << 29, directions 0=E 1=NE .. 7=SE); *h_n = their number (0 when the border has <= 2 contour points). */
int ptocr_dbpost_debug_states(ptocr_dbpost_t h, int img, int k, uint32_t *h_states, int cap, int32_t *h_n);

/* Third inspection hook: label image int32[H][W] (entries valid at run starts: component root = its raster-first pixel index,
 * -1 = background connected to the frame, -2 - k = root selected as candidate k) and per-word labels int32[H][ceil(W/32)] (root of
 * the run covering the word's first pixel) of image img of the last call, which must have been H x W. */
int ptocr_dbpost_debug_labels(ptocr_dbpost_t h, int img, int H, int W, int32_t *h_labels, int32_t *h_word_labels);

/* Timing experiments only (tools/dbg/post_stamps.py): with PTOCR_DBPOST_STAMPS=1 in the environment when the workspace is created, the
 * two per-border stage kernels record s_memtime stamps of their phases, 16 per record; copies the first n_records records.
 * h_stamps == NULL clears the buffer instead (the next call's records then stand alone). */
int ptocr_dbpost_debug_stamps(ptocr_dbpost_t h, int64_t *h_stamps, long n_records);

/* Test hook of the hull -> quad hand-off inside border_stage_kernel: plants, for every border slot, a ready word that claims the NEXT call's
 * epoch with a candidate count beyond the quad's table.  The next call must still return the same boxes (a quad that meets such a word
 * defers the border to the full-size pass instead of indexing by the count). */
int ptocr_dbpost_debug_plant(ptocr_dbpost_t h);

/* Labelling route of the following calls on this workspace: 0 = chosen from the workspace's last eight calls (default: a noise-like
 * batch keeps the noise route on for eight calls), 1 = text route (LDS slabs), 2 = noise route (global union-find, bottom strip first).
 * Boxes never depend on the route (tests/test_gpu_dbpost.py runs every case on both); only the time does. */
int ptocr_dbpost_set_route(ptocr_dbpost_t h, int route);

/* ---- pre-process next to the path --------------------------------------------------------------
 * Item descriptors live in device memory (arrays of the structs below, natural C layout). */
typedef struct { long src_off; int sh, sw; int rh, rw; long dst_off; int dh, dw; int flip; int pad_; } ptocr_pre_item;
/* flip = 1 reads the source rotated by 180 degrees (run_ocr.py:209-211: a line the direction classifier calls "180").
 * The classifier's own input (ClsResizeImg, rec_img_aug.py:29-37) is mode 0 with mean = std = 0.5. */
/* Batched u8 HxWx3 (BGR) -> cv2.resize(INTER_LINEAR) to rh x rw -> fp32 f32[dh][dw][cpad], zero beyond (rh, rw).
 * mode 0: 3 channels (RGB order if swap_rb) as (x/255 - mean)/std  (DetResizeForTest+ToTensor+Normalize, operators.py:41-112,155-252)
 * mode 1: BGR2GRAY then (x/255 - 0.5)/0.5 in channel 0                (resize_norm_img, rec_img_aug.py:108-134) */
int ptocr_preprocess_u8_f32(const uint8_t *d_src, float *d_dst, const void *d_items, int n_items, int max_dst_pixels,
                            int mode, int swap_rb, int cpad, const float *h_mean3, const float *h_std3, void *stream);
typedef struct { double minv[9]; int left, top; int cw, ch; int rot90; int img; long dst_off; } ptocr_warp_item;
/* Batched get_part_img (utils/utility.py:53-78): perspective warp (INTER_LINEAR, BORDER_REPLICATE) of n text boxes into packed
 * u8 crops; d_img is a stack of equally sized u8 HxWx3 images and item.img selects the source image (0 for a single image);
 * rot90 = np.rot90(crop, 1) when h >= 1.5 w (run_ocr.py:189-191). */
int ptocr_warp_crops_u8(const uint8_t *d_img, int H, int W, uint8_t *d_dst, const void *d_items, int n_items,
                        int max_crop_pixels, void *stream);

/* ---- recognition ----------------------------------------------------------------------------------------
 * Sequences are batch-major: row = b*T + t (no Im2Seq permute, no decode transpose). */
/* y[M,Nout] = x[M,K] @ w[Nout,K]^T + bias; K % 32 == 0, Nout % 64 == 0 (pad rows of w / bias with zeros), ldy >= Nout */
int ptocr_linear_f32(const float *d_x, const float *d_w, const float *d_bias, float *d_y, int M, int K, int Nout,
                     int ldy, void *stream);
/* One bidirectional LSTM layer, zero initial state, torch gate order i,f,g,o, hidden size H == 256.
 * d_xproj f32[B][T][2][4H]: x@W_ih^T + b_ih + b_hh for (forward, backward), one ptocr_linear_f32 with Nout = 8H;
 * d_whh f32[2][4H][H] (weight_hh_l0, weight_hh_l0_reverse); d_out f32[B][T][2H] (forward half | backward half). */
int ptocr_lstm_bidir_f32(const float *d_xproj, const float *d_whh, float *d_out, int T, int B, int H, void *stream);
/* When every (16-line group, direction) can own four CUs the recurrence runs split over four workgroups that exchange h each
 * time step through a buffer owned by the library PER (device, stream).  Co-residency of the four workgroups cannot be
 * guaranteed (another stream or process may hold CUs), so the exchange spins are bounded and every split call is followed,
 * on the same stream, by a repair launch of the exchange-free kernel that recomputes the layer when the exchange timed out:
 * d_out is correct either way, without a host round trip.  PTOCR_LSTM_SPLIT=0 selects the exchange-free form outright. */
/* split-form calls so far / how many of them were recomputed by the repair pass (synchronises) */
int ptocr_lstm_stats(int *split_calls, int *repaired);
/* split-form calls in which EVERY workgroup of EVERY (16 lines, direction) pair found its three partners on its own XCD (checked at run
 * time with HW_REG_XCC_ID) and exchanged h through that XCD's L2; a workgroup that did not uses the write-through exchange, with the
 * same results (counted per call by the repair launch that follows every split call on its stream) */
int ptocr_lstm_same_xcd_calls(int *calls);
/* the same per workgroup: workgroups of split calls (current device) that took the same-XCD exchange / workgroups those calls had */
int ptocr_lstm_fast_workgroups(long long *fast, long long *all);
/* test hook: placement of the split form's workgroups.  -1 = PTOCR_LSTM_COLOCATE (default: on); 0 = round-4 layout, write-through
 * exchange; 1 = the four parts of a pair on one XCD; 3 = as 1, but part 1 of every pair is forced onto the write-through stores, so a
 * pair mixes both kinds of store (what a workgroup that finds foreign XCC ids among its partners does) */
void ptocr_lstm_set_colocate(int mode);
/* test hook: polls of one exchange before a workgroup gives up (0 = default 65536); 1 = give up at the first miss AND one of
 * the four workgroups withholds its slice, which forces the time-out and so the repair path */
void ptocr_lstm_set_spin_limit(unsigned polls);
/* round-1 ABI, kept: always 0 now (a timed-out exchange is repaired on the stream, see above) */
int ptocr_lstm_check(void);
/* Per row of d_x f32[rows][ld] (first C columns valid, ld % 4 == 0): first arg-max and the max softmax probability.
 * is_prob = 0: d_x holds logits, prob = 1 / sum(exp(x - max));  is_prob = 1: d_x already holds probabilities, prob = max.
 * With rows = b*T + t this is preds.argmax(2), preds.max(2) of rec_postprocess.py:83-84 as int32[B][T], f32[B][T]. */
int ptocr_ctc_greedy_f32(const float *d_x, int rows, int C, int ld, int is_prob, int32_t *d_idx, float *d_prob, void *stream);
/* The CTC head's FC fused with those reductions (rec_ctc_head.py:17-36 + rec_postprocess.py:80-84): idx[r] / prob[r] of
 * logits = x[M,K] @ w[Nout,K]^T + bias over the first C columns, WITHOUT writing the logits (1.1 GB at B = 512).  Same values as
 * ptocr_linear_f32 followed by ptocr_ctc_greedy_f32(is_prob = 0): the indices are identical, prob within fp32 rounding.
 * K % 32 == 0, Nout % 128 == 0 (zero-padded rows of w / bias); d_work: M * (Nout / 64) * 16 bytes of device scratch. */
int ptocr_linear_ctc_greedy_f32(const float *d_x, const float *d_w, const float *d_bias, int M, int K, int Nout, int C,
                                void *d_work, int32_t *d_idx, float *d_prob, void *stream);
/* y[r][0:C] = softmax(x[r][0:C]) (rec_ctc_head.py:33) */
int ptocr_softmax_rows_f32(const float *d_x, int rows, int C, int ld, float *d_y, int ldy, void *stream);

#ifdef __cplusplus
}
#endif
#endif
