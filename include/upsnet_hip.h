/*
 * upsnet_hip.h -- C ABI of libupsnet_hip.so: the MI355X (gfx950 / CDNA4) replacement for the native
 * operators on UPSNet's per-image inference hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream; NULL = default stream);
 *   - the caller allocates every output and every workspace (the reference does the same:
 *     functions/deform_conv.py:44-45, functions/roialign.py:37); the library never allocates
 *     device memory, except upsnet_nms_host (the `_nms` drop-in, which the reference also lets malloc);
 *   - functions return 0 on success, non-zero on error; upsnet_last_error() then describes it
 *     (the reference returned 1/0 and printed, SURVEY.md section 8b "Errors");
 *   - layouts: "nchw" entry points take the reference's contiguous NCHW tensors and are drop-in
 *     replacements for the reference natives; "nhwc" entry points are the MI355X-native fast path
 *     (channels-last physical layout so that a bilinear tap is one contiguous C-vector).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root, uber-research/UPSNet).
 */
#ifndef UPSNET_HIP_H
#define UPSNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Last error message of the calling thread ("" if none). */
const char *upsnet_last_error(void);
/* Name of the kernel instance the calling thread's most recent convolution launch took where the choice is made inside the library
 * (bf16-mode convolutions: "conv3x3_wreg<8,2>", "conv1x1_wreg<4,1>", "conv3x3_halo", "conv_bf16<2>"; Winograd: "wino<0,64,128>").
 * Test introspection (tests/test_layerwise_gpu.py asserts the forms a benchmarked configuration runs on); "" before any launch. */
const char *upsnet_last_kernel_form(void);
/* Library ABI version (bumped on any signature change). */
int upsnet_abi_version(void);

/* Zero `bytes` bytes at `ptr` (any alignment) with a plain kernel on `stream`. The scratch of the selection ops is cleared this way
 * instead of hipMemsetAsync (a replacement for the reference's THCudaTensor zero_() / cudaMemset of its scratch, e.g.
 * upsnet/nms/nms_kernel.cu:112-113): a captured forward then holds kernel nodes only (memset nodes fault at graph replay). */
int upsnet_zero_fill(void *stream, void *ptr, size_t bytes);

/* ============================== ROIAlign ============================== */

/* Replaces roi_align_forward_gpu_kernel_launcher (upsnet/operators/src/roi_align_cuda.cpp:26-30,
 * bound to Python as roi_align_cuda.roi_align_forward, roi_align_cuda.cpp:39-75,114-118).
 * bottom_data [B,C,H,W] NCHW, bottom_rois [N,5] = (batch,x1,y1,x2,y2), top_data [N,C,PH,PW]. */
int upsnet_roi_align_forward(void *stream, const float *bottom_data, float spatial_scale, int num_rois,
                             int height, int width, int channels, int pooled_height, int pooled_width,
                             int sampling_ratio, const float *bottom_rois, float *top_data);

/* Same op on channels-last features: feat [B,H,W,C], out [N,PH,PW,C]. C % 4 == 0. */
int upsnet_roi_align_forward_nhwc(void *stream, const float *feat_nhwc, int batch, int height, int width,
                                  int channels, float spatial_scale, const float *rois, int num_rois,
                                  int pooled_height, int pooled_width, int sampling_ratio, float *out_nhwc);

/* Replaces FPNRoIAlign.forward (upsnet/operators/modules/fpn_roi_align.py:32-62): level assignment
 * floor(2+log2(sqrt(wh)/224+1e-6)) on device, all four levels in one launch, output in the
 * original ROI order. feat_nhwc[l] is [H_l,W_l,C]; out_nhwc [N,PH,PW,C]; levels_out (optional) [N].
 * num_rois_dev (optional): device int32 with the number of valid rois (<= num_rois); rows beyond it
 * are written as zeros. The pointer arrays themselves are HOST arrays of 4 entries. */
int upsnet_fpn_roi_align_forward(void *stream, const float *const feat_nhwc[4], const int feat_h[4],
                                 const int feat_w[4], const float spatial_scale[4], int channels,
                                 const float *rois, int num_rois, const int *num_rois_dev, int pooled_height,
                                 int pooled_width, int sampling_ratio, float *out_nhwc, int *levels_out);

/* r13: the same launch with a workgroup -> ROI table: workgroup b takes ROI order[b] (order == NULL: its own index). Output rows stay at the
 * ROI's own index: results are bit-identical whatever the table, only the workgroup -> XCD assignment changes (workgroup b runs on XCD
 * b % 8, each XCD has its own L2). upsnet_fpn_roi_order fills the table so that every XCD's workgroups take one contiguous range of the
 * ROIs bucketed by (pyramid level, 1/16 stripe of the image, 1/8 column cell; image_height x image_width = the extent the ROIs live in):
 * neighbouring ROIs share an L2 instead of every L2 fetching its own
 * copy of the shared pyramid cells (FPNRoIAlign.forward, upsnet/operators/modules/fpn_roi_align.py:32-62, processes level by level too).
 * num_rois <= 2048 for the table; caller-allocated int[num_rois]. */
int upsnet_fpn_roi_order(void *stream, const float *rois, int num_rois, const int *num_rois_dev, int image_height, int image_width,
                         int *order_out);
int upsnet_fpn_roi_align_forward_ordered(void *stream, const float *const feat_nhwc[4], const int feat_h[4], const int feat_w[4],
                                         const float spatial_scale[4], int channels, const float *rois, int num_rois, const int *num_rois_dev,
                                         int pooled_height, int pooled_width, int sampling_ratio, float *out_nhwc, int *levels_out,
                                         const int *order);

/* Development knob of upsnet_fpn_roi_align_forward: 0 = LDS tap-table kernel, one register set; 1 = two sets;
 * 2 = the r03-r07 kernel (per-bin tap setup in registers); 3 = the table kernel loading only the UNIQUE corner cells of a bin (r11);
 * 4 = 3 with packed fp32 blend arithmetic; < 0 = back to the default behaviour: the environment variable UPSNET_ROI_KERNEL (a variant number)
 * if set, else automatic -- 3 for launches with >= 100 bins per ROI, else 0. A call with variant >= 0 overrides the variable. All variants
 * return the same bits. upsnet_roi_geometry: the bins of a ROI are split over
 * workgroups until the launch has `target_workgroups` (default 1536), at least `min_bins` (default 8) bins each; 0 = default. */
void upsnet_roi_tuning(int variant);
void upsnet_roi_geometry(int target_workgroups, int min_bins);

/* ============================== Deformable convolution ============================== */

/* Replaces deformable_im2col_gpu_kernel_launcher (upsnet/operators/src/deform_conv_cuda.cpp:25-30,
 * bound as deform_conv_cuda.deform_im2col, deform_conv_cuda.cpp:49-68). NCHW, parallel_imgs images.
 * data_col is [C*kh*kw, parallel_imgs, Ho, Wo]. */
int upsnet_deform_im2col(void *stream, const float *data_im, const float *data_offset, int channels,
                         int height, int width, int ksize_h, int ksize_w, int pad_h, int pad_w, int stride_h,
                         int stride_w, int dilation_h, int dilation_w, int parallel_imgs, int deformable_group,
                         float *data_col);

/* Replaces modulated_deformable_im2col_gpu_kernel_launcher
 * (upsnet/operators/src/mod_deform_conv_cuda.cpp:24-31, bound as mod_deform_im2col :51-74). */
int upsnet_mod_deform_im2col(void *stream, const float *data_im, const float *data_offset,
                             const float *data_mask, int batch_size, int channels, int height_im, int width_im,
                             int height_col, int width_col, int kernel_h, int kernel_w, int pad_h, int pad_w,
                             int stride_h, int stride_w, int dilation_h, int dilation_w, int deformable_group,
                             float *data_col);

/* ---- backward natives of the same pybind modules (API parity for fine-tuning; not on the inference path).
 * fp32 NCHW, argument order of the reference launchers. The scatter kernels accumulate INTO their output with
 * fp32 atomics exactly like the reference (zero it first; summation order unspecified => parity 1e-5-level);
 * the col2im_coord gathers have a fixed order and are bit-exact against the oracle. */

/* Replaces roi_align_backward_gpu_kernel_launcher (upsnet/operators/src/roi_align_cuda.cpp:32-36, bound as
 * roi_align_cuda.roi_align_backward :77-112; kernel roi_align_kernel.cu:238-349). top_diff [N,C,PH,PW],
 * bottom_diff [B,C,H,W] accumulated. */
int upsnet_roi_align_backward(void *stream, const float *top_diff, float spatial_scale, int batch_size, int num_rois,
                              int height, int width, int channels, int pooled_height, int pooled_width,
                              int sampling_ratio, const float *bottom_rois, float *bottom_diff);

/* Replaces deformable_col2im_gpu_kernel_launcher (deform_conv_cuda.cpp:32-38, bound as deform_col2im :70-86;
 * kernel deform_conv_kernel.cu:293-383). data_col [C*kh*kw, parallel_imgs, Ho, Wo] -> grad_im accumulated. */
int upsnet_deform_col2im(void *stream, const float *data_col, const float *data_offset, int channels, int height,
                         int width, int ksize_h, int ksize_w, int pad_h, int pad_w, int stride_h, int stride_w,
                         int dilation_h, int dilation_w, int parallel_imgs, int deformable_group, float *grad_im);

/* Replaces deformable_col2im_coord_gpu_kernel_launcher (deform_conv_cuda.cpp:40-46, bound as deform_col2im_coord
 * :88-105; kernel deform_conv_kernel.cu:391-500). grad_offset is overwritten. */
int upsnet_deform_col2im_coord(void *stream, const float *data_col, const float *data_im, const float *data_offset,
                               int channels, int height, int width, int ksize_h, int ksize_w, int pad_h, int pad_w,
                               int stride_h, int stride_w, int dilation_h, int dilation_w, int parallel_imgs,
                               int deformable_group, float *grad_offset);

/* Replaces modulated_deformable_col2im_gpu_kernel_launcher (mod_deform_conv_cuda.cpp:33-40, bound as
 * mod_deform_col2im :76-92; kernel mod_deform_conv_kernel.cu:251-311, launcher :409-433). Like the reference
 * launcher (:423) the kernel receives pad_h for BOTH paddings; pad_w is accepted and ignored. */
int upsnet_mod_deform_col2im(void *stream, const float *data_col, const float *data_offset, const float *data_mask,
                             int batch_size, int channels, int height_im, int width_im, int height_col, int width_col,
                             int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h, int stride_w,
                             int dilation_h, int dilation_w, int deformable_group, float *grad_im);

/* Replaces modulated_deformable_col2im_coord_gpu_kernel_launcher (mod_deform_conv_cuda.cpp:42-50, bound as
 * mod_deform_col2im_coord :94-114; kernel mod_deform_conv_kernel.cu:313-381). grad_offset / grad_mask overwritten. */
int upsnet_mod_deform_col2im_coord(void *stream, const float *data_col, const float *data_im, const float *data_offset,
                                   const float *data_mask, int batch_size, int channels, int height_im, int width_im,
                                   int height_col, int width_col, int kernel_h, int kernel_w, int pad_h, int pad_w,
                                   int stride_h, int stride_w, int dilation_h, int dilation_w, int deformable_group,
                                   float *grad_offset, float *grad_mask);

/* Fused deformable convolution forward (v1 when mask == NULL, v2 otherwise), the MI355X-native
 * replacement for DeformConvFunction.forward's im2col + torch.mm (functions/deform_conv.py:43-57)
 * with no column buffer: the dense implicit-GEMM kernel below with a bilinear-gather A operand
 * (fp32 MFMA, v_mfma_f32_32x32x2_f32). Up to 4 feature maps that share the same weights (the FCN
 * head's four FPN levels) go in ONE launch.
 *   x[l]      [H_l, W_l, Cin]  NHWC (batch 1)
 *   offset[l] [Ho_l, Wo_l, 2*kh*kw] NHWC (channel order as the reference: 2*(i*kw+j) = dh, +1 = dw)
 *   mask[l]   [Ho_l, Wo_l, kh*kw]  NHWC, or mask == NULL
 *   wpack     [kh*kw*Cin, ldw] from upsnet_conv_pack_weight; bias [Cout] or NULL
 *   out[l]    [Ho_l, Wo_l, Cout] NHWC;  relu != 0 fuses max(.,0).
 * Pointer/shape arrays are HOST arrays of nlev entries. Cin % 32 == 0, deformable_group == 1. */
int upsnet_deform_conv_forward_nhwc(void *stream, int nlev, const float *const x[], const float *const offset[],
                                    const float *const mask[], float *const out[], const int height[],
                                    const int width[], int cin, int cout, int kh, int kw, int pad_h, int pad_w,
                                    int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                                    const float *wpack, int ldw, const float *bias, int relu);

/* Second-generation fused deformable convolution (csrc/deform_fused.hip): same operator and tensor layouts as
 * upsnet_deform_conv_forward_nhwc (replaces DeformConvFunction.forward, functions/deform_conv.py:43-57, and the modulated
 * v2 form, mod_deform_conv_kernel.cu:187-249), but the K walk is channel-slab outermost / tap innermost (neighbouring taps
 * re-use their corner lines out of L1 / L2), the per-(pixel, tap) sampling descriptors live in an LDS table built once per
 * workgroup, and the weights are read in MFMA fragment order straight from L2:
 *   wpack: upsnet_dcn_packed_weight_floats(cout, cin, kh, kw) floats, filled by upsnet_dcn_pack_weight from the
 *          nn.Conv2d-layout weight [Cout, Cin, kh, kw]; square pad / stride / dilation; at most 25 taps; Cin % 32 == 0.
 * upsnet_dcn_tuning: development knob (register-set / occupancy variants for A/B runs). */
size_t upsnet_dcn_packed_weight_floats(int cout, int cin, int kh, int kw);
int upsnet_dcn_pack_weight(void *stream, const float *weight, int cout, int cin, int kh, int kw, float *wpack);
int upsnet_deform_conv_fused_nhwc(void *stream, int nlev, const float *const x[], const float *const offset[],
                                  const float *const mask[], float *const out[], const int height[], const int width[],
                                  int cin, int cout, int kh, int kw, int pad, int stride, int dil, const float *wpack,
                                  const float *bias, int relu);
/* Split-K form for ONE small map (the DCN bottlenecks of a ResNet-101-DCN backbone have 96-273 tiles for 256 CUs): the (channel slab,
 * tap) walk is divided over ksplit (2..8) workgroups per tile, raw partial sums go to `workspace`
 * (upsnet_deform_conv_fused_splitk_workspace_bytes) and a fixed-order reduce kernel adds bias / ReLU (deterministic). Cout % 4 == 0. */
size_t upsnet_deform_conv_fused_splitk_workspace_bytes(int height, int width, int cout, int kh, int kw, int pad, int stride, int dil,
                                                       int ksplit);
int upsnet_deform_conv_fused_nhwc_splitk(void *stream, const float *x, const float *offset, const float *mask, float *out, int height,
                                         int width, int cin, int cout, int kh, int kw, int pad, int stride, int dil,
                                         const float *wpack, const float *bias, int relu, int ksplit, void *workspace);
void upsnet_dcn_tuning(int variant);

/* The same fused deformable convolution on the bf16 matrix cores (csrc/deform_fused_bf16.hip; BASELINE.json configs[2], opt-in with
 * the other bf16 kernels): blended samples and weights rounded to bf16 (nearest even), exact products, fp32 accumulation. Arguments
 * as upsnet_deform_conv_fused_nhwc; wpack: upsnet_dcn_pack_weight_bf16 into upsnet_dcn_packed_weight_bf16_elems 2-byte elements. */
size_t upsnet_dcn_packed_weight_bf16_elems(int cout, int cin, int kh, int kw);
int upsnet_dcn_pack_weight_bf16(void *stream, const float *weight, int cout, int cin, int kh, int kw, void *wpack);
int upsnet_deform_conv_fused_nhwc_bf16(void *stream, int nlev, const float *const x[], const float *const offset[],
                                       const float *const mask[], float *const out[], const int height[], const int width[], int cin,
                                       int cout, int kh, int kw, int pad, int stride, int dil, const void *wpack, const float *bias,
                                       int relu);

/* ============================== Input blob (the step before the path) ============================== */

/* BaseDataset.prep_im_for_blob + im_list_to_blob (upsnet/dataset/base_dataset.py:143-173,898-923) in one kernel:
 * image_hwc uint8 [height,width,3] (BGR) -> float32, minus pixel_means (in double, like numpy's float32 -= float64),
 * bilinear resize by im_scale (cv2.resize INTER_LINEAR semantics; identity for im_scale == 1), zero padded to
 * [padded_h, padded_w]. resized_h/w = cvRound(height*im_scale) etc. are computed by the caller (host mirror:
 * upsnet_amd/dataset/blob.py). nhwc4 == 0: blob is planar [3,padded_h,padded_w] (the reference's layout);
 * nhwc4 != 0: blob is [padded_h,padded_w,4] with a zero 4th channel (input of upsnet_conv2d_stem_nhwc4_f32). */
int upsnet_prep_image_u8(void *stream, const unsigned char *image_hwc, int height, int width, const double pixel_means[3],
                         double im_scale, int resized_h, int resized_w, int padded_h, int padded_w, int nhwc4, float *blob);

/* Layout plumbing: fp32 [N,C<=4,H,W] planar -> [N,H,W,4] (missing channels zero). */
int upsnet_image_to_nhwc4(void *stream, const float *nchw, int batch, int channels, int height, int width, float *nhwc4);

/* ============================== Dense convolution ============================== */

/* upsnet_conv2d_nhwc_f32 (below) with per-map weights: wpack[i] packs the weights of map i (same geometry, Cin, Cout, ldw); no bias,
 * no residual. Replaces the four per-level products W_l y_l of the FCN head's 1x1 score layer (upsnet/models/fcn.py:101-104, the
 * 512-channel concat + conv commuted below the bilinear upsamples) as one launch. */
int upsnet_conv2d_nhwc_f32_multiw(void *stream, int nseg, const float *const x[], float *const out[], const int batch[],
                                  const int height[], const int width[], int Cin, const float *const wpack[], int ldw, int Cout,
                                  int KH, int KW, int stride, int pad, int relu);

/* Replaces nn.Conv2d (+ folded frozen BatchNorm + bias + residual add + ReLU) of the backbone / FPN / RPN / heads
 * (upsnet/models/resnet.py:53-100, fpn.py:78-104, rpn.py:52-57, rcnn.py:79-87, fcn.py:88-108): NHWC fp32 implicit
 * GEMM on v_mfma_f32_32x32x2_f32 with a fused epilogue. Up to 5 feature maps sharing the weights per launch.
 *   x[i] [N_i,H_i,W_i,Cin] NHWC (Cin % 32 == 0), wpack [KH*KW*Cin, ldw] from upsnet_conv_pack_weight (ldw = Cout
 *   rounded up to a multiple of 32), bias [Cout] or NULL, residual (NULL or array; entries [N_i,Ho,Wo,Cout]),
 *   out[i] [N_i,Ho,Wo,Cout]; batch == NULL means N_i = 1. Pointer/shape arrays are HOST arrays of nseg entries.
 *   residual_up != 0: residual entries are [N_i,Ho/2,Wo/2,Cout] and are added through a nearest x2 upsampling
 *   (the FPN top-down add, F.interpolate(scale_factor=2, mode='nearest') + sum, fpn.py:34,90-96). */
int upsnet_conv2d_nhwc_f32(void *stream, int nseg, const float *const x[], const float *const residual[],
                           float *const out[], const int batch[], const int height[], const int width[], int Cin,
                           const float *wpack, int ldw, const float *bias, int Cout, int KH, int KW, int stride, int pad,
                           int relu, int residual_up);

/* upsnet_conv2d_nhwc_f32 for ONE small feature map (res4 / res5: 2048-8192 pixels = too few 64x64 tiles to fill 256 CUs): the
 * K walk is split over ksplit (2..8) workgroups per tile, raw partial sums go to `workspace`
 * (upsnet_conv2d_splitk_workspace_bytes) and a second kernel adds them in a fixed order with the bias / residual / ReLU
 * epilogue -- deterministic, differs from the unsplit kernel only by fp32 summation order. */
size_t upsnet_conv2d_splitk_workspace_bytes(int batch, int height, int width, int Cout, int KH, int KW, int stride, int pad, int ksplit);
int upsnet_conv2d_nhwc_f32_splitk(void *stream, const float *x, const float *residual, float *out, int batch, int height, int width,
                                  int Cin, const float *wpack, int ldw, const float *bias, int Cout, int KH, int KW, int stride,
                                  int pad, int relu, int ksplit, void *workspace);

/* 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on the fp32 MFMA (csrc/conv_wino.hip; 16/36 of the multiplies
 * of the direct form; all arithmetic fp32; differs from the direct kernel by fp32 rounding only, ~1e-5 absolute on unit-scale
 * data -- cuDNN, the reference's convolution backend, uses the same algorithm). Replaces the 3x3 nn.Conv2d layers of
 * upsnet/models/resnet.py:64-77, fpn.py:60-98, rpn.py:34-47 and the mask head rcnn.py:96-116 where the map has enough 2x2
 * tiles. A workgroup keeps all 16 transform-domain accumulators of its 64 tiles x 64 channels in registers: one walk over the
 * input channels, each patch pixel loaded and transformed once, lane-local output transform. Cin % 16 == 0 (conv_fill: % 32),
 * ldw = Cout rounded up to 64; wpack (16 * Cin * ldw floats, fragment order) from upsnet_conv_pack_weight_winograd
 * (weight [Cout,Cin,3,3]). Same calling convention as upsnet_conv2d_nhwc_f32 (multi-map launch, bias / residual / ReLU). */
int upsnet_conv2d_winograd_nhwc_f32(void *stream, int nseg, const float *const x[], const float *const residual[],
                                    float *const out[], const int batch[], const int height[], const int width[], int Cin,
                                    const float *wpack, int ldw, const float *bias, int Cout, int relu);

/* The same convolution of ONE map with the K walk split `ksplit` (2..8) ways -- maps with too few 2x2 tiles to fill 256 CUs
 * (res4 / res5 3x3, FPN P4; replaces the same nn.Conv2d call sites). workspace: upsnet_conv2d_splitk_workspace_bytes(batch,
 * height, width, Cout, 3, 3, 1, 1, ksplit) bytes. Fixed summation order: bit-repeatable. */
int upsnet_conv2d_winograd_nhwc_f32_splitk(void *stream, const float *x, const float *residual, float *out, int batch, int height,
                                           int width, int Cin, const float *wpack, int ldw, const float *bias, int Cout, int relu,
                                           int ksplit, void *workspace);
int upsnet_conv_pack_weight_winograd(void *stream, const float *weight, int cout, int cin, int ldw, float *wpack);

/* The Winograd convolution on 32-tile x 32-CHANNEL workgroups for any Cout (ldw = Cout rounded up to 32; weights packed by
 * upsnet_conv_pack_weight_winograd_tn32): twice the workgroups of the 32 x 64 form with half the work each, bit-identical results --
 * for the part of a launch that would otherwise run as a nearly empty last round (the tail ROIs of the mask head, rcnn.py:96-116). */
int upsnet_conv2d_winograd_nhwc_f32_tn32(void *stream, int nseg, const float *const x[], const float *const residual[],
                                         float *const out[], const int batch[], const int height[], const int width[], int Cin,
                                         const float *wpack, int ldw, const float *bias, int Cout, int relu);
int upsnet_conv_pack_weight_winograd_tn32(void *stream, const float *weight, int cout, int cin, int ldw, float *wpack);
/* One launch of both forms over a batch x [batch, H, W, Cin] (the mask head's 100 ROIs: 616 workgroups of the 32 x 64 form for 512 slots):
 * images [0, n_main) on 32 x 64 workgroups (wpack / ldw), images [n_main, batch) on 32 x 32 workgroups (wpack32 / ldw32), which the
 * dispatcher deals out as the main workgroups retire. out [batch, H, W, Cout]; bit-identical to upsnet_conv2d_winograd_nhwc_f32. */
int upsnet_conv2d_winograd_nhwc_f32_tail(void *stream, const float *x, float *out, int batch, int n_main, int height, int width, int Cin,
                                         const float *wpack, int ldw, const float *wpack32, int ldw32, const float *bias, int Cout, int relu);

/* Dense convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- BASELINE.json configs[2]
 * ("bf16 compute / fp32 accumulate") and its fp32-equivalent 3-term split. OPT-IN: the fp32 kernel above is the default.
 * Activations stay fp32 NHWC in HBM (same tensors as upsnet_conv2d_nhwc_f32) and are split to bf16 on the way into LDS.
 *   wpack_lo == NULL : "bf16"   -- operands rounded to bf16, one MFMA per product
 *   wpack_lo != NULL : "bf16x3" -- a = a_hi + a_lo, b = b_hi + b_lo; a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (error ~2^-16
 *                      relative per product, fp32 accumulation): inside the 1e-4 fp32-logit tolerance
 * wpack_hi / wpack_lo: bf16 [KH*KW*Cin/32][ldw][32] from upsnet_conv_pack_weight_bf16, ldw = Cout rounded up to 64.
 * Same calling convention as upsnet_conv2d_nhwc_f32 (without residual_up). relu: bit 0 = ReLU; bit 1 = `residual` is at half
 * resolution and is added through a nearest x2 upsampling (1x1 kernels only: the FPN top-down add). */
int upsnet_conv2d_nhwc_bf16(void *stream, int nseg, const float *const x[], const float *const residual[], float *const out[],
                            const int batch[], const int height[], const int width[], int Cin, const void *wpack_hi,
                            const void *wpack_lo, int ldw, const float *bias, int Cout, int KH, int KW, int stride, int pad,
                            int relu);
int upsnet_conv_pack_weight_bf16(void *stream, const float *weight, int cout, int cin, int kh, int kw, int ldw, void *wpack_hi,
                                 void *wpack_lo);
/* The FIRST bottleneck of a stage (projection shortcut; upsnet/models/resnet.py:84-100) as ONE launch on the bf16 matrix cores:
 *   out = relu(conv3(relu(conv2(relu(conv1_s(x))))) + proj_s(x)),  conv1_s / proj_s: 1x1 with stride s (1 or 2), conv2: 3x3 / 1 / 1.
 * x [N,Hin,Win,Cin] bf16, out [N,H,W,4 Cm] bf16, H = (Hin - 1) / s + 1; (Cm, Cin) in {(64, 64), (128, 256), (256, 512)}. w1: fragments
 * of [Cm, Cin]; w2: of [Cm, 9 Cm] (tap-major k); w3d: of [4 Cm, Cm + Cin] = [W3 | Wd] -- conv3 and the projection are one GEMM over
 * K = [t2 ; x], the shortcut is accumulated in fp32; b1, b2 [Cm]; b3d = b3 + bd [4 Cm]. Fragment order as upsnet_bottleneck_bf16. */
int upsnet_bottleneck_proj_bf16(void *stream, const void *x, void *out, int batch, int height_in, int width_in, int cin, int cmid, int stride,
                                const void *w1, const void *w2, const void *w3d, const float *b1, const float *b2, const float *b3d);
/* A/B switch of the 3x3 / stride 1 / 256 -> 256 layers in the plain bf16 mode (csrc/conv3x3_wreg_bf16.hip: weights fed to the MFMA
 * from L2, one barrier per 32-channel slab): enable 0 = the general haloed-patch kernel, 1 = default; tile_rows 0 = automatic,
 * 8 / 16 = forced tile height, 2 = 2-row tiles (256-channel workgroups), 1 = 2-row tiles with 128-channel workgroups. Same products
 * and K order either way. */
int upsnet_conv_bf16_tuning(int enable, int tile_rows);
/* A/B switch of the 1x1 layers with bf16 activations in the plain bf16 mode (csrc/conv1x1_wreg_bf16.hip: both MFMA operands loaded
 * from global memory in fragment order, no LDS, no barrier): enable 0 = conv_bf16_kernel, 1 = default (the layers it is faster on:
 * shortcut epilogue, Cin <= 256), 2 = every layer it can compute. Same products and K order. */
int upsnet_conv1x1_bf16_tuning(int enable);
/* The backbone stem in one launch on the fp32 matrix cores (upsnet/models/resnet.py:347-356; the headline path): convolution 7x7 /
 * stride 2 / pad 3 (Cin <= 3 -> 64, frozen BN folded: + bias) + ReLU + max-pool 3x3 / stride 2 / pad 1 -- the 134 MB convolution output
 * of a 1024x2048 image never reaches HBM. x4 [N,H,W,4] fp32 (RGB + a zero channel), wpack = upsnet_stem_pool_pack_weight_f32(weight
 * [64,Cin,7,7]) (2 x 75 x 64 floats), bias [64] or NULL, out [N,Hp,Wp,64] fp32, Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (likewise the
 * width). Exact fp32 products (v_mfma_f32_32x32x2_f32), fixed summation order. */
int upsnet_stem_pool_pack_weight_f32(void *stream, const float *weight, int cin, float *wpack);
int upsnet_stem_pool_f32(void *stream, const float *x4, int batch, int height, int width, const float *wpack, const float *bias, float *out);
/* The backbone stem in one launch on the bf16 matrix cores (upsnet/models/resnet.py:347-356 in the bf16 mode): convolution 7x7 /
 * stride 2 / pad 3 (Cin <= 4 -> 64, frozen BN folded: + bias) + ReLU + max-pool 3x3 / stride 2 / pad 1. x4 [N,H,W,4] fp32 (RGB + a zero
 * channel: upsnet_image_to_nhwc4 / upsnet_prep_image_u8), wpack = upsnet_stem_pool_pack_weight_bf16(weight [64,Cin,7,7]) (28 KiB),
 * bias [64] or NULL, out [N,Hp,Wp,64] bf16, Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (likewise the width). The convolution result
 * is rounded to bf16 before the pool; the image is rounded to bf16 on the way into LDS. */
int upsnet_stem_pool_pack_weight_bf16(void *stream, const float *weight, int cin, void *wpack);
int upsnet_stem_pool_bf16(void *stream, const float *x4, int batch, int height, int width, const void *wpack, const float *bias, void *out);
/* ConvTranspose2d(kernel 2, stride 2, pad 0) (+ bias, + ReLU) of a bf16 NHWC map on the bf16 matrix cores (the mask head's
 * upsampling layer, upsnet/models/rcnn.py:132-133, in the bf16 mode): one GEMM [N H W, Cin] x [Cin, 4 Cout] with a scatter epilogue.
 * x [N,H,W,Cin] bf16; wpack_hi = upsnet_conv_pack_weight_bf16 of the [4 Cout, Cin, 1, 1] matrix whose rows are ordered (ky, kx, c),
 * ldw = 4 Cout; bias [Cout] or NULL; out [N,2H,2W,Cout] fp32 (out_bf16 = 0) or bf16. Cin % 64 == 0, Cout % 32 == 0. */
int upsnet_deconv2x2_nhwc_bf16(void *stream, const void *x, int batch, int height, int width, int Cin, const void *wpack_hi, int ldw,
                               const float *bias, int Cout, int relu, void *out, int out_bf16);

/* One identity bottleneck of the backbone (upsnet/models/resnet.py:84-100: conv1 1x1 C -> Cm, conv2 3x3 Cm -> Cm, conv3 1x1 Cm -> C,
 * C = 4 Cm, stride 1, no projection; frozen BN folded; out = relu(conv3(relu(conv2(relu(conv1(x))))) + x)) as ONE launch on the bf16
 * matrix cores -- the bf16 mode of BASELINE.json configs[2]; the intermediates never leave the LDS. x, out: [N,H,W,C] bf16 NHWC;
 * cmid in {64, 128, 256, 512}; w1 / w2 / w3: bf16 in MFMA-fragment order [out channel / 32][k / 16][64 lanes][8]
 * (element (cb, ks, lane, e) = W[cb 32 + lane % 32][ks 16 + (lane / 32) 8 + e], k = input channel for the 1x1 layers and
 * (ky 3 + kx) Cm + c for the 3x3 layer); b1, b2 [Cm], b3 [C] fp32. Every intermediate is rounded to bf16 where the three separate
 * launches of upsnet_conv2d_nhwc_bf16 round it. */
int upsnet_bottleneck_bf16(void *stream, const void *x, void *out, int batch, int height, int width, int cmid, const void *w1,
                           const void *w2, const void *w3, const float *b1, const float *b2, const float *b3);

/* 7x7/2 stem (upsnet/models/resnet.py:347-356, conv1 + frozen BN + ReLU): Cin <= 4 input given as NHWC with 4 channels
 * (x [N,H,W,4], 4th channel ignored by zero weights; see upsnet_image_to_nhwc4 / upsnet_prep_image_u8). One K slab of the
 * implicit GEMM is one kernel row: 8 consecutive pixels x 4 channels. wpack [KH*32, ldw] from upsnet_conv_pack_weight_stem
 * (weight [Cout,Cin,KH,KW], KW <= 8); out [N,Ho,Wo,Cout] NHWC; bias/ReLU fused. */
int upsnet_conv2d_stem_nhwc4_f32(void *stream, const float *x, int batch, int height, int width, const float *wpack,
                                 int ldw, const float *bias, int Cout, int KH, int KW, int stride, int pad, int relu,
                                 float *out);
int upsnet_conv_pack_weight_stem(void *stream, const float *weight, int cout, int cin, int kh, int kw, int ldw, float *wpack);

/* nn.ConvTranspose2d(Cin, Cout, 2, 2, 0) (+ bias + ReLU) of the mask head (upsnet/models/rcnn.py:60,84) as ONE GEMM with
 * 4*Cout columns (dy, dx, co) whose epilogue stores column (dy,dx,co) of input pixel (h,w) at output pixel (2h+dy, 2w+dx):
 * x [N,H,W,Cin] NHWC -> out [N,2H,2W,Cout] NHWC. wpack [Cin, ldw >= 4*Cout] from upsnet_deconv2x2_pack_weight
 * (weight [Cin,Cout,2,2], the PyTorch layout). */
int upsnet_deconv2x2_nhwc_f32(void *stream, const float *x, int batch, int height, int width, int Cin, const float *wpack,
                              int ldw, const float *bias, int Cout, int relu, float *out);
int upsnet_deconv2x2_pack_weight(void *stream, const float *weight, int cin, int cout, int ldw, float *wpack);

/* The same transposed convolution on the lean 1x1 GEMM kernel (csrc/conv1x1.hip, scatter epilogue): x [N,H,W,Cin] NHWC ->
 * out [N,2H,2W,Cout] NHWC. wpack: upsnet_dcn_pack_weight(W', 4*Cout, Cin, 1, 1) of the [4*Cout, Cin, 1, 1] matrix W' whose rows are
 * (dy, dx, co) = weight[ci, co, dy, dx]; bias4 = the bias repeated for the four (dy, dx), [4*Cout], or NULL. Cin % 32 == 0,
 * Cout % 32 == 0. Same fp32 MFMA arithmetic and K order as upsnet_conv1x1_frag_nhwc_f32. Replaces the reference's
 * nn.ConvTranspose2d(dim, dim, 2, 2, 0) + ReLU of the mask head (upsnet/models/rcnn.py:132-133). */
int upsnet_deconv2x2_frag_nhwc_f32(void *stream, const float *x, float *out, int batch, int height, int width, int Cin,
                                   const float *wpack, const float *bias4, int Cout, int relu);

/* Development knob for A/B measurements: force_tile = 0 auto, 1: 128x128, 2: 128x64, 3: 128x32, 4: 64x128, 5: 64x64,
 * 6: 64x64 with 64-channel K slabs (pixels x output channels per workgroup). winograd_tiles = 0 auto, 32 / 64: 2x2 tiles per
 * workgroup of the Winograd kernel. Not needed in production. */
void upsnet_conv_tuning(int winograd_tiles, int force_tile);

/* weight [Cout, Cin, kh, kw] (nn.Conv2d layout) -> wpack [kh*kw*Cin, ldw] (tap-major rows, zero-padded columns). */
int upsnet_conv_pack_weight(void *stream, const float *weight, int cout, int cin, int kh, int kw, int ldw, float *wpack);

/* 1x1 convolution (stride 1 / 2) as a lean fp32 MFMA GEMM (csrc/conv1x1.hip): the 1x1 layers of the ResNet bottlenecks
 * (upsnet/models/resnet.py:53-100: conv1 / conv3 / downsample) and the FPN laterals with their top-down add (fpn.py:78-104).
 * out = relu?(conv1x1(x; stride) + bias + residual); x [N,H,W,Cin] NHWC, out [N,Ho,Wo,Cout] NHWC (Ho = (H-1)/stride + 1);
 * residual like out, or -- residual_up != 0 -- [N,Ho/2,Wo/2,Cout] read through a nearest x2 upsampling. Cin % 32 == 0.
 * wpack: upsnet_dcn_pack_weight(weight [Cout,Cin,1,1], cout, cin, 1, 1) (MFMA fragment order, read straight from L2).
 * Weights must be FINITE: when Cin / 32 (or a split-K slice) is odd the K walk runs one padding step that multiplies zero activations
 * by the last weight slab (0 x Inf would be NaN; an exact zero otherwise).
 * upsnet_conv1x1_tuning: development knob (0 auto, 64 / 128: output channels per workgroup). */
int upsnet_conv1x1_frag_nhwc_f32(void *stream, const float *x, const float *residual, float *out, int batch, int height, int width,
                                 int Cin, const float *wpack, const float *bias, int Cout, int stride, int relu, int residual_up);
void upsnet_conv1x1_tuning(int bn);
/* Two 1x1 convolutions of the same input in one launch -- conv1 and the projection shortcut (`downsample`) of a stage's first bottleneck
 * (upsnet/models/resnet.py:53-100 reads x twice): out_a [N,Ho,Wo,cout_a] from rows [0, cout_a) of the concatenated weight, out_b
 * [N,Ho,Wo,cout_b] from the rest; wpack = upsnet_dcn_pack_weight of the concatenated [cout_a + cout_b, Cin, 1, 1] weight, bias = both
 * biases concatenated (or NULL), cout_a % 32 == 0. Bit-identical to two upsnet_conv1x1_frag_nhwc_f32 launches. */
int upsnet_conv1x1_siblings_nhwc_f32(void *stream, const float *x, float *out_a, float *out_b, int batch, int height, int width, int Cin,
                                     const float *wpack, const float *bias, int cout_a, int cout_b, int stride, int relu_a, int relu_b);
/* development knob: waves per workgroup of the 32-pixel form of upsnet_conv1x1_pair_nhwc_f32 (8: two per SIMD, default; 4). */
void upsnet_conv1x1_pair32_tuning(int waves);

/* 3x3 / stride 1 / pad 1 convolution (+ bias, ReLU) of up to 5 NHWC maps sharing weights by Winograd F(4x4, 3x3) (csrc/conv_wino36.hip,
 * r11 / r12): 36 multiplies per 16 outputs, 0.5625 of the F(2x2) kernel's matrix work. Interpolation points {0, 3/4, -3/4, 3/2, -3/2, inf}:
 * within rtol = atol = 1e-4 of float64 on the model's layers, measured worst error 0.07-0.15 of that bound at the model's shapes (3-4x the
 * F(2x2) kernel's 0.014-0.032; tools/bench_winograd36.py), asserted <= 0.35 in tests/test_conv_gpu.py (small maps and the real 1024x2048
 * sizes). H, W of any size (maps with H < 4 take the bounds-flagged loads throughout). The
 * contract of upsnet_conv2d_winograd_nhwc_f32 without a residual; Cin % 32 == 0; wpack (36 * Cin * ldw floats, ldw = Cout rounded up to
 * 64) from upsnet_conv_pack_weight_winograd36 (weight [Cout, Cin, 3, 3]). */
int upsnet_conv2d_winograd36_nhwc_f32(void *stream, int nseg, const float *const x[], float *const out[], const int batch[],
                                      const int height[], const int width[], int Cin, const float *wpack, int ldw, const float *bias,
                                      int Cout, int relu);
int upsnet_conv_pack_weight_winograd36(void *stream, const float *weight, int cout, int cin, int ldw, float *wpack);
/* r13: upsnet_conv2d_winograd36_nhwc_f32 of ONE map with the channel walk of every tile split over `ksplit` (2..8, <= Cin / 16) workgroups: each
 * stores the output transform of its partial sums, the shared reduce / epilogue kernel adds them in a fixed order (+ bias, ReLU;
 * bit-repeatable). For maps with fewer 32-tile x 64-channel workgroups than CUs (res3 / res4 conv2, FPN P4 at 1024x2048). workspace:
 * upsnet_conv2d_winograd36_splitk_workspace_bytes(batch, height, width, Cout, ksplit) bytes, caller-allocated. */
size_t upsnet_conv2d_winograd36_splitk_workspace_bytes(int batch, int height, int width, int Cout, int ksplit);
int upsnet_conv2d_winograd36_nhwc_f32_splitk(void *stream, const float *x, float *out, int batch, int height, int width, int Cin,
                                             const float *wpack, int ldw, const float *bias, int Cout, int relu, int ksplit, void *workspace);

/* 1x1 convolution (stride 1 / 2; + bias, + residual, ReLU) on SMALL tiles (csrc/conv1x1_ksw.hip, r13): `v_mfma_f32_16x16x4_f32` fragments,
 * every wave of a workgroup holds the whole tile_pixels x tile_channels tile and walks a quarter of K straight from the NHWC map (no LDS /
 * barrier in the K loop); the four partial tiles are added in LDS in a fixed order ((w0 + w1) + (w2 + w3): bit-repeatable, independent
 * of the batch size for a given tile). For maps the 64-pixel tiles of upsnet_conv1x1_frag_nhwc_f32 do not spread evenly over the CUs
 * (UPSNet-101-DCN at 800x1333: 4200-pixel maps; the layers replaced are the bottlenecks' conv1 / conv3, upsnet/models/resnet.py:53-153).
 * Tiles (pixels x channels of a workgroup; anything else is an error): split_n == 0 -- 16 x 64, 32 x 32, 32 x 64, 64 x 64, the waves split K
 * as described; split_n == 1 -- 16 x 256, 32 x 128, 32 x 256: the four waves split the tile's CHANNELS instead and each walks all of K (no
 * reduction; for a short K walk into many channels, a bottleneck's conv3). x [N,H,W,Cin] NHWC, out [N,Ho,Wo,Cout] NHWC,
 * residual like out or NULL; Cin % 16 == 0, Cout % 4 == 0, all pointers 16-byte aligned. wpack: upsnet_conv1x1_ksw_packed_weight_floats
 * floats from upsnet_conv1x1_ksw_pack_weight (weight [Cout, Cin]). Finite weights assumed (a K step beyond a wave's share multiplies
 * zeros by the last valid weights). upsnet_conv1x1_ksw_tuning(1000 split_n + 100 RB + CB of a wave's tile): development knob forcing an instance (0: the caller's). */
size_t upsnet_conv1x1_ksw_packed_weight_floats(int cout, int cin);
int upsnet_conv1x1_ksw_pack_weight(void *stream, const float *weight, int cout, int cin, float *wpack);
int upsnet_conv1x1_ksw_nhwc_f32(void *stream, const float *x, const float *residual, float *out, int batch, int height, int width, int Cin,
                                const float *wpack, const float *bias, int Cout, int stride, int relu, int tile_pixels, int tile_channels,
                                int split_n);
void upsnet_conv1x1_ksw_tuning(int tile);

/* 3x3 / stride 1 / pad 1 convolution (+ bias, ReLU) into Cout <= 32 channels on the small-tile scheme of upsnet_conv1x1_ksw_nhwc_f32
 * (csrc/conv1x1_ksw.hip, r13): 16-pixel x 32-channel workgroups, the (tap, channel) walk split over the four waves, no LDS in the K loop.
 * Replaces the separate launch pair (split-K general kernel + reduce) of the 18-channel offset predictors of the deformable bottlenecks on
 * small maps (conv2_offset, upsnet/models/resnet.py:102-153). x [N,H,W,Cin] NHWC, out [N,H,W,Cout] NHWC, Cin % 16 == 0; wpack:
 * upsnet_conv3x3_ksw_packed_weight_floats(Cin) floats from upsnet_conv3x3_ksw_pack_weight (weight [Cout, Cin, 3, 3]). Finite weights
 * assumed. Fixed summation order for a given pixel count (bit-repeatable; the number of waves that share a tile's walk -- 4, 8 or 16 --
 * follows the tile count, so a pixel's bits may differ between launches of different sizes: not for ROI batches). */
size_t upsnet_conv3x3_ksw_packed_weight_floats(int cin);
int upsnet_conv3x3_ksw_pack_weight(void *stream, const float *weight, int cout, int cin, float *wpack);
int upsnet_conv3x3_ksw_nhwc_f32(void *stream, const float *x, float *out, int batch, int height, int width, int Cin, const float *wpack,
                                const float *bias, int Cout, int relu);

/* upsnet_conv1x1_frag_nhwc_f32 with the K walk of every tile split over `ksplit` (2..16) workgroups + the shared reduce / epilogue
 * kernel (bias, residual, ReLU; fixed summation order: bit-repeatable). For maps whose tile count does not spread evenly over the CUs:
 * a workgroup of this kernel keeps all four SIMDs of its CU at the MFMA rate, so a launch lasts (most workgroups on one CU) x (one K
 * walk) -- 264 tiles on 256 CUs take two walks, 264 x 4 quarter walks take five quarters (models/hipconv.py: UPSNET_CONV1X1_BALANCE).
 * workspace: upsnet_conv1x1_splitk_workspace_bytes(batch, Ho, Wo, Cout, ksplit) bytes, caller-allocated. No residual_up. */
size_t upsnet_conv1x1_splitk_workspace_bytes(int batch, int out_height, int out_width, int Cout, int ksplit);
int upsnet_conv1x1_frag_nhwc_f32_splitk(void *stream, const float *x, const float *residual, float *out, int batch, int height, int width,
                                        int Cin, const float *wpack, const float *bias, int Cout, int stride, int relu, int ksplit,
                                        void *workspace);

/* Two chained 1x1 convolutions of consecutive bottlenecks in ONE launch (csrc/conv1x1_pair.hip) -- the tail of block b and the
 * head of block b+1 of Bottleneck.forward (upsnet/models/resnet.py:53-100):
 *     out1 = relu(conv1x1(x; w3) + bias3 + residual)      x [pixels, C0], residual / out1 [pixels, C1]   (NHWC, pixels = N*H*W)
 *     out2 = relu(conv1x1(out1; w1) + bias1)              out2 [pixels, C2]
 * Both results are bit-identical to two upsnet_conv1x1_frag_nhwc_f32 launches; out1 is not read back from HBM.
 * w3pack / w1pack: upsnet_dcn_pack_weight(weight, cout, cin, 1, 1). Supported: (C0, C2) = (64, 64), (128, 128) or (256, 256), C1 % 128 == 0 (the res2, res3 and -- on 32-pixel tiles -- res4 stages). */
int upsnet_conv1x1_pair_nhwc_f32(void *stream, const float *x, const float *residual, float *out1, float *out2, long pixels, int C0,
                                 const float *w3pack, const float *bias3, int C1, const float *w1pack, const float *bias1, int C2);

/* ============================== NMS ============================== */

/* Drop-in for `_nms` (upsnet/nms/gpu_nms.hpp:15, nms_kernel.cu:97-150): HOST pointers in and out,
 * boxes already sorted by descending score, keep_out = indices into the sorted array. */
int upsnet_nms_host(int *keep_out_host, int *num_out_host, const float *boxes_host, int boxes_num,
                    int boxes_dim, float nms_overlap_thresh, int device_id);

/* Batched device-resident hard NMS: P independent problems in one set of launches, no host round trip.
 * Replaces gpu_nms (upsnet/nms/gpu_nms.pyx:23-38) including its score sort: visiting order is
 * (score desc, index desc), the pinned meaning of scores.argsort()[::-1].
 *   boxes [P,nmax,4], scores [P,nmax], counts [P] (device int32, counts[p] <= nmax <= 8192)
 *   pre_removed (optional) [P,nmax] uint8: boxes dropped before NMS (min-size filter)
 *   keep_idx [P,nmax] int32 out (original indices, visiting order), keep_cnt [P] int32 out
 *   workspace: upsnet_nms_workspace_bytes(P, nmax) bytes. */
size_t upsnet_nms_workspace_bytes(int num_problems, int nmax);
int upsnet_nms_batched(void *stream, const float *boxes, const float *scores, const int *counts,
                       const uint8_t *pre_removed, int num_problems, int nmax, float thresh, int *keep_idx,
                       int *keep_cnt, void *workspace);

/* Development knob of the greedy scan: 0 = default (row-layout scan for problems of <= 1024 boxes, general scan reading L2 above
 * that), 1 = general scan on an LDS copy of the mask (<= 1024 boxes), 2 = general scan reading L2 at every size. Same keep lists. */
void upsnet_nms_tuning(int lds_staging);

/* cpu_nms (upsnet/nms/cpu_nms.pyx:29-80, behind cpu_nms_wrapper, upsnet/nms/nms.py:37-40) on the device: same layout, visiting
 * order and IoU expression as upsnet_nms_batched, but suppression at `overlap >= thresh` (:77) with `thresh` a double (a Python
 * float in the reference's compiled module; the fp32 overlap is compared in double). */
int upsnet_cpu_nms_batched(void *stream, const float *boxes, const float *scores, const int *counts, int num_problems,
                           int nmax, double thresh, int *keep_idx, int *keep_cnt, void *workspace);

/* Batched device soft-NMS = cpu_soft_nms (upsnet/nms/cpu_nms.pyx:91-196), P independent problems (RPN levels / classes) per
 * launch, one workgroup each, state resident in LDS for nmax <= 4096. Bit-compatible with the compiled reference on the
 * returned prefix (pinned: tests/test_ref_cpu_nms.py):
 *   boxes [P,nmax,5] in/out (x1,y1,x2,y2,score; rows >= n_out[p] unspecified), inds [P,nmax] int64 out (original index of each
 *   surviving row), counts [P] device int32 (valid rows per problem; NULL = nmax everywhere), n_out [P] device int32.
 *   method: 0 hard, 1 linear, 2 gaussian. workspace: upsnet_soft_nms_batched_workspace_bytes(P, nmax) (may be NULL when
 *   nmax <= 4096). */
size_t upsnet_soft_nms_batched_workspace_bytes(int num_problems, int nmax);
int upsnet_soft_nms_batched(void *stream, float *boxes, int64_t *inds, const int *counts, int num_problems, int nmax,
                            float sigma, float Nt, float threshold, int method, int *n_out, void *workspace);

/* Single problem: upsnet_soft_nms_batched with P = 1 and all n rows valid. */
size_t upsnet_soft_nms_workspace_bytes(int n);
int upsnet_soft_nms(void *stream, float *boxes, int64_t *inds, int n, float sigma, float Nt, float threshold,
                    int method, int *n_out, void *workspace);

/* ============================== RPN proposals ============================== */

/* Replaces PyramidProposalFunction.forward + PyramidProposal.forward
 * (upsnet/operators/functions/pyramid_proposal.py:62-222, modules/pyramid_proposal.py:61-67) with
 * individual_proposals=True (every shipped yaml; the joint branch is upsnet_pyramid_proposals_joint_strided below), entirely on device.
 *   cls_prob[l] [A,H_l,W_l] NCHW (A anchors), bbox_pred[l] [4A,H_l,W_l] NCHW  (HOST arrays of pointers)
 *   anchors_host [nlev*A*4] base anchors (generate_anchors), strides_host [nlev]
 *   im_info (device) [3] = (H, W, scale)
 *   rois_out [post_nms_top_n,5], scores_out [post_nms_top_n], num_out device int32
 *   workspace: upsnet_proposal_workspace_bytes(...) bytes. */
size_t upsnet_proposal_workspace_bytes(int nlev, const int *heights_host, const int *widths_host, int num_anchors,
                                       int pre_nms_top_n, int post_nms_top_n);
int upsnet_pyramid_proposals(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                             const int *heights_host, const int *widths_host, const int *strides_host,
                             const float *anchors_host, int num_anchors, const float *im_info, int pre_nms_top_n,
                             int post_nms_top_n, float nms_thresh, float min_size, float *rois_out,
                             float *scores_out, int *num_out, void *workspace);

/* Same operator on strided inputs, so that channel slices of NHWC maps (the RPN head's [H,W,15] output: 3 scores + 12 deltas per
 * pixel) are consumed in place instead of being copied to NCHW first: element (channel c, pixel) of level l's score map is
 * cls_prob[l][c * cls_chan_stride[l] + pixel * cls_pix_stride[l]], likewise for the deltas. A NULL stride array means NCHW
 * (channel stride H*W, pixel stride 1). */
int upsnet_pyramid_proposals_strided(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                     const long *cls_chan_stride, const long *cls_pix_stride, const long *box_chan_stride,
                                     const long *box_pix_stride, const int *heights_host, const int *widths_host,
                                     const int *strides_host, const float *anchors_host, int num_anchors, const float *im_info,
                                     int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float *rois_out,
                                     float *scores_out, int *num_out, void *workspace);
/* r13: the same + roi_order_out (int[post_nms_top_n] or NULL): the workgroup -> ROI table of upsnet_fpn_roi_align_forward_ordered for these
 * rois (see upsnet_fpn_roi_order), written by the launch that ranks them -- no launch of its own. post_nms_top_n <= 2048 when given. */
int upsnet_pyramid_proposals_strided_ordered(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                     const long *cls_chan_stride, const long *cls_pix_stride, const long *box_chan_stride,
                                     const long *box_pix_stride, const int *heights_host, const int *widths_host,
                                     const int *strides_host, const float *anchors_host, int num_anchors, const float *im_info,
                                     int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float *rois_out,
                                     float *scores_out, int *num_out, void *workspace, int *roi_order_out);

/* individual_proposals=False -- the DEFAULT of the reference's constructors (functions/pyramid_proposal.py:26,
 * modules/pyramid_proposal.py:24) -- i.e. the joint branch functions/pyramid_proposal.py:181-208: every anchor of every level is decoded,
 * clipped and size-filtered (:132-141), the survivors are concatenated (:176-177) and ranked jointly (`scores.argsort()[::-1]`: equal
 * scores -> higher concatenation index first), the first pre_nms_top_n go through ONE NMS, the first post_nms_top_n kept boxes are
 * the result. Same arguments and workspace as upsnet_pyramid_proposals_strided; pre_nms_top_n <= 8192.
 *   rois_out [post_nms_top_n,5] / scores_out: the kept boxes in NMS visiting order, zero rows behind; num_out = their number.
 * NOT done here: the reference then pads the list back to post_nms_top_n rows with `np.random.choice(keep, ...)` (:205-207, numpy's
 * global generator -- host state); upsnet_amd/operators/functions/pyramid_proposal.py draws those indices on the host from the
 * same generator (identical stream) and gathers the rows. */
int upsnet_pyramid_proposals_joint_strided(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                           const long *cls_chan_stride, const long *cls_pix_stride, const long *box_chan_stride,
                                           const long *box_pix_stride, const int *heights_host, const int *widths_host,
                                           const int *strides_host, const float *anchors_host, int num_anchors, const float *im_info,
                                           int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float *rois_out,
                                           float *scores_out, int *num_out, void *workspace);

/* ============================== Detection selection (MaskROI) ============================== */

/* Replaces MaskROI.forward (upsnet/operators/modules/mask_roi.py:36-146): decode + clip, per-class (or
 * class-agnostic) score threshold, NMS, global top-max_det; dummy ROI when nothing survives.
 *   rois [N,5], bbox_delta [N,4*C], cls_prob [N,C], num_rois_dev optional device count
 *   boxes_out [cap,5], scores_out [cap], cls_out [cap] int64, src_out [cap] int32 (ROI row of each det),
 *   num_out device int32; cap = upsnet_mask_roi_capacity(N, C, class_agnostic). */
int upsnet_mask_roi_capacity(int num_rois, int num_classes, int class_agnostic);
size_t upsnet_mask_roi_workspace_bytes(int num_rois, int num_classes, int class_agnostic);
int upsnet_mask_roi(void *stream, const float *rois, const float *bbox_delta, const float *cls_prob, int num_rois,
                    const int *num_rois_dev, int num_classes, const float *im_info, int class_agnostic,
                    float score_thresh, float nms_thresh, int max_det, const float reg_weights_host[4],
                    float *boxes_out, float *scores_out, int64_t *cls_out, int *src_out, int *num_out,
                    void *workspace);
/* The same with MaskROI's `clip_boxes` constructor argument (mask_roi.py:25,53-54): clip_boxes = 0 skips clip_boxes(), the
 * decoded boxes reach the NMS and the output unclipped; upsnet_mask_roi is clip_boxes = 1. */
int upsnet_mask_roi_ex(void *stream, const float *rois, const float *bbox_delta, const float *cls_prob, int num_rois,
                       const int *num_rois_dev, int num_classes, const float *im_info, int class_agnostic, int clip_boxes,
                       float score_thresh, float nms_thresh, int max_det, const float reg_weights_host[4],
                       float *boxes_out, float *scores_out, int64_t *cls_out, int *src_out, int *num_out,
                       void *workspace);

/* The reference runs the mask head on the per-class detections AND on the class-agnostic "panoptic" detections
 * (upsnet/models/resnet_upsnet.py:190,215). Both are selections from the same (ROI row, class) table, and every ROI goes
 * through the mask head independently, so a panoptic detection that is also a per-class detection gets bit-identical
 * logits. Given the src/cls outputs of two upsnet_mask_roi calls (set A, set B) this entry writes, for every b in B,
 * map_out[b] = its row in the concatenation [A ; unmatched of B], compacts the unmatched boxes of B (in order) into
 * extra_boxes [cap_b,5] and their count into num_extra (device int). One mask-head pass over [A ; extra] then serves both. */
int upsnet_mask_roi_dedup(void *stream, const int *a_src, const int64_t *a_cls, const int *num_a, int cap_a, const int *b_src,
                          const int64_t *b_cls, const float *b_boxes, const int *num_b, int cap_b, int *map_out,
                          float *extra_boxes, int *num_extra);

/* out[k, :] = mask_logit[row[k], cls[k], :] for k < K: the class plane of each panoptic detection's mask logits
 * (upsnet/models/resnet_upsnet.py:215-221) as one gather. mask_logit is addressed by element strides (stride_n per ROI, stride_c
 * per class, stride_e per spatial element: NHWC or NCHW); row int32 [K], cls int64 [K] are clamped to the valid range; out [K, hw]. */
int upsnet_mask_logit_gather(void *stream, const float *mask_logit, int n_rows, int num_classes, int hw, long stride_n,
                             long stride_c, long stride_e, const int *row, const int64_t *cls, int K, float *out);

/* ============================== Panoptic head ============================== */

/* Replaces MaskRemoval.forward's selection (upsnet/operators/modules/mask_removal.py:50-93):
 *   mask_rois [m,4], cls_prob [m], mask_logit [m,ms,ms], cls_idx [m] int64 (1-based; 0 = dummy)
 *   keep_inds [m] int64 out (visiting order), num_keep device int32 (>=1: the reference's [0] fallback)
 *   real_keep device int32: 0 when the fallback fired (then the single mask plane is all zeros).
 *   m_dev (optional device int32): the arrays have capacity m and *m_dev (<= m) valid rows -- lets the call sit inside a
 *   HIP graph before the host knows the detection count; NULL: all m rows are valid. */
size_t upsnet_mask_removal_workspace_bytes(int m, int num_thing_classes, int height, int width);
int upsnet_mask_removal(void *stream, const float *mask_rois, const float *cls_prob, const float *mask_logit,
                        const int64_t *cls_idx, int m, const int *m_dev, int mask_size, int num_thing_classes, int height, int width,
                        double fraction_threshold, int64_t *keep_inds, int *num_keep, int *real_keep,
                        void *workspace);

/* Tail of the fixed-capacity forward (models/resnet_upsnet.py): kept_cls[i] = cls[keep[i]], kept_scores[i] = scores[keep[i]] for the
 * kept panoptic detections (panoptic_cls_inds / panoptic_cls_probs of upsnet/models/resnet_upsnet.py:242-247; rows past *num_keep read
 * row 0) and counters = {*det_num, *pan_num, *extra_num, *num_keep}: one launch instead of clamp + 2 index_select + 2 cat. */
int upsnet_panoptic_tail_pack(void *stream, const int64_t *keep, const int *num_keep, int K, const int64_t *cls, const float *scores,
                              const int *det_num, const int *pan_num, const int *extra_num, int64_t *kept_cls, float *kept_scores,
                              int *counters);

/* Materialise MaskRemoval's mask_energy [k,H,W] (mask_removal.py:86) for the kept instances. */
int upsnet_mask_paste(void *stream, const float *mask_rois, const float *mask_logit, const int64_t *keep_inds,
                      const int *num_keep, const int *real_keep, int kmax, int mask_size, int height, int width,
                      float *mask_energy);

/* Materialise SegTerm's seg_inst_energy [k,H,W] (upsnet/operators/modules/unary_logits.py:95-103).
 * boxes [k,4] image coords (already scaled), cls [k] int64, class_map [num_classes] int64. */
int upsnet_seg_term(void *stream, const float *fcn_output, int num_seg, int height, int width, const float *boxes,
                    const int64_t *cls, const int64_t *class_map, int k, float *seg_inst);

/* Fused parameter-free panoptic head (upsnet/models/resnet_upsnet.py:223-243 + SegTerm + mask paste):
 * one pass over fcn_output, nothing materialised.
 *   fcn_output [S,H,W] planar; mask_rois [m,5] (col 0 = batch), mask_logit [m,ms,ms], cls_idx [m] int64,
 *   keep_inds/num_keep/real_keep from upsnet_mask_removal; class_map [num_classes] int64
 *   pan_out [H,W] int64, sem_out [H,W] int64 or NULL (argmax over S, resnet_upsnet.py:213). */
int upsnet_panoptic_fuse(void *stream, const float *fcn_output, int num_seg, int height, int width, int num_stuff,
                         const float *mask_rois, const float *mask_logit, const int64_t *cls_idx,
                         const int64_t *keep_inds, const int *num_keep, const int *real_keep, int kmax,
                         int mask_size, const int64_t *class_map, int enable_void, int64_t *pan_out,
                         int64_t *sem_out);

/* Tail of the semantic head (upsnet/models/fcn.py:94-100: upsample x2/x4/x8, concat, 1x1 `score` conv), with the linear
 * 1x1 convolution commuted below the bilinear upsampling: part[l] = W[:, 128 l:128 (l+1)] . y_l at level l's own
 * resolution, NHWC [height >> l, width >> l, num_seg] (made with upsnet_conv2d_nhwc_f32); this entry computes
 *   score[y,x,s] = bias[s] + part[0][y,x,s] + sum_{l>=1} bilinear_up_{2^l}(part[l])[y,x,s]   (align_corners = False)
 * into score [height, width, num_seg] NHWC. bias may be NULL. 1 <= nlev <= 4. */
int upsnet_fcn_score_combine(void *stream, int nlev, const float *const part[], int num_seg, int height, int width,
                             const float *bias, float *score);

/* Same as upsnet_panoptic_fuse (enable_void branch) but ALSO fuses FCNHead's x4 bilinear upsampling
 * (F.interpolate(score, None, 4, 'bilinear', align_corners=False), upsnet/models/fcn.py:101): takes the low-resolution
 * fcn_score [S,Hs,Ws] (score_nhwc = 0) or [Hs,Ws,S] (score_nhwc = 1); label maps are [Hs*scale, Ws*scale]. */
int upsnet_panoptic_fuse_up(void *stream, const float *fcn_score, int score_nhwc, int num_seg, int score_h, int score_w,
                            int scale, int num_stuff, const float *mask_rois, const float *mask_logit,
                            const int64_t *cls_idx, const int64_t *keep_inds, const int *num_keep, const int *real_keep,
                            int kmax, int mask_size, const int64_t *class_map, int64_t *pan_out, int64_t *sem_out);

/* Reference-shaped fusion on materialised planes (resnet_upsnet.py:234-243):
 * seg_inst, mask_energy [k,H,W]. */
int upsnet_panoptic_argmax(void *stream, const float *fcn_output, int num_seg, int height, int width, int num_stuff,
                           const float *seg_inst, const float *mask_energy, int k, int enable_void,
                           int64_t *pan_out);

/* ============================== Unified panoptic result (the step after the path) ============================== */

/* BaseDataset.get_unified_pan_result (upsnet/dataset/base_dataset.py:332-371) for one image, on the device:
 *   pan [H,W] int64 panoptic label map of the network (stuff ids 0..id_last_stuff, instance j -> id_last_stuff+1+j, 255 void),
 *   seg [H,W] int64 semantic argmax map, cls_inds [num_inst] int64 (1-based thing class of instance j),
 *   id_last_stuff = num_seg_classes - num_classes, stuff_area_limit in pixels (reference default 4*64*64).
 * pan_2ch uint8 [H,W,3]: channel 0 = category, channel 1 = instance id (enumerate index + 1, 0 for stuff/void), channel 2 = 0.
 * workspace: upsnet_unified_pan_workspace_bytes() bytes of device memory (zeroed by the call). */
size_t upsnet_unified_pan_workspace_bytes(void);
int upsnet_unified_pan_result(void *stream, const int64_t *pan, const int64_t *seg, const int64_t *cls_inds, int num_inst,
                              int height, int width, int id_last_stuff, int num_seg_classes, int stuff_area_limit,
                              void *workspace, unsigned char *pan_2ch);

/* im_post (upsnet/upsnet_end2end_test.py:95-152) without the full-image pass per detection: for detection d the mask
 * probability of its class (mask_prob [n, C, M, M]; C == 1: class-agnostic) is zero-padded to (M+2)^2, resized (cv2
 * INTER_LINEAR) to the box expanded by (M+2)/M and truncated to int32 (expand_boxes, bbox_transform.py:365-381), thresholded
 * at 0.5 and pasted into an im_height x im_width mask -- of which only the run-length encoding is produced:
 * transitions[d, 0..transition_count[d]) are the column-major pixel indices (x * im_height + y) at which the mask value changes,
 * ascending (the first one starts the first run of ones). pycocotools' counts are the differences of [0, transitions..., H*W].
 * transition_count[d] > cap signals overflow (retry with a larger cap). pred_boxes [n,4] in image coordinates. */
int upsnet_im_post_rle(void *stream, const float *pred_boxes, const float *mask_prob, const int64_t *cls_inds, int num_det,
                       int num_mask_channels, int mask_size, int im_height, int im_width, int cap, unsigned *transitions,
                       int *transition_count);

#ifdef __cplusplus
}
#endif
#endif /* UPSNET_HIP_H */
