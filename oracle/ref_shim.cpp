// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY. extern "C" handles onto the REFERENCE's own launchers,
// which oracle/Makefile hipifies and compiles straight from /root/reference (never vendored) into
// oracle/_ref/. Prototypes as declared by the reference host wrappers:
//   upsnet/operators/src/roi_align_cuda.cpp:26-30, deform_conv_cuda.cpp:25-30,
//   mod_deform_conv_cuda.cpp:24-31, upsnet/nms/gpu_nms.hpp:15.
#include <hip/hip_runtime.h>

void roi_align_forward_gpu_kernel_launcher(hipStream_t stream, const float *bottom_data, const float spatial_scale,
                                           const int num_rois, const int height, const int width, const int channels,
                                           const int pooled_height, const int pooled_width, const int sampling_ratio,
                                           const float *bottom_rois, float *top_data);
void deformable_im2col_gpu_kernel_launcher(hipStream_t stream, const float *data_im, const float *data_offset,
                                           const int channels, const int height, const int width, const int ksize_h,
                                           const int ksize_w, const int pad_h, const int pad_w, const int stride_h,
                                           const int stride_w, const int dilation_h, const int dilation_w,
                                           const int parallel_imgs, const int deformable_group, float *data_col);
void modulated_deformable_im2col_gpu_kernel_launcher(hipStream_t stream, const float *data_im, const float *data_offset,
                                                     const float *data_mask, const int batch_size, const int channels,
                                                     const int height_im, const int width_im, const int height_col,
                                                     const int width_col, const int kernel_h, const int kenerl_w,
                                                     const int pad_h, const int pad_w, const int stride_h,
                                                     const int stride_w, const int dilation_h, const int dilation_w,
                                                     const int deformable_group, float *data_col);
int roi_align_backward_gpu_kernel_launcher(hipStream_t stream, const float *top_diff, const float spatial_scale, const int batch_size,
                                           const int num_rois, const int height, const int width, const int channels,
                                           const int pooled_height, const int pooled_width, const int sampling_ratio,
                                           const float *bottom_rois, float *bottom_diff);
void deformable_col2im_gpu_kernel_launcher(hipStream_t stream, const float *data_col, const float *data_offset, const int channels,
                                           const int height, const int width, const int ksize_h, const int ksize_w, const int pad_h,
                                           const int pad_w, const int stride_h, const int stride_w, const int dilation_h,
                                           const int dilation_w, const int parallel_imgs, const int deformable_group, float *grad_im);
void deformable_col2im_coord_gpu_kernel_launcher(hipStream_t stream, const float *data_col, const float *data_im,
                                                 const float *data_offset, const int channels, const int height, const int width,
                                                 const int ksize_h, const int ksize_w, const int pad_h, const int pad_w,
                                                 const int stride_h, const int stride_w, const int dilation_h, const int dilation_w,
                                                 const int parallel_imgs, const int deformable_group, float *grad_offset);
void modulated_deformable_col2im_gpu_kernel_launcher(hipStream_t stream, const float *data_col, const float *data_offset,
                                                     const float *data_mask, const int batch_size, const int channels,
                                                     const int height_im, const int width_im, const int height_col,
                                                     const int width_col, const int kernel_h, const int kernel_w, const int pad_h,
                                                     const int pad_w, const int stride_h, const int stride_w, const int dilation_h,
                                                     const int dilation_w, const int deformable_group, float *grad_im);
void modulated_deformable_col2im_coord_gpu_kernel_launcher(hipStream_t stream, const float *data_col, const float *data_im,
                                                           const float *data_offset, const float *data_mask, const int batch_size,
                                                           const int channels, const int height_im, const int width_im,
                                                           const int height_col, const int width_col, const int kernel_h,
                                                           const int kernel_w, const int pad_h, const int pad_w, const int stride_h,
                                                           const int stride_w, const int dilation_h, const int dilation_w,
                                                           const int deformable_group, float *grad_offset, float *grad_mask);
void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id);

extern "C" {
void ref_roi_align_forward(void *stream, const float *feat, float scale, int num_rois, int height, int width, int channels,
                           int ph, int pw, int sampling_ratio, const float *rois, float *out)
{
    roi_align_forward_gpu_kernel_launcher((hipStream_t)stream, feat, scale, num_rois, height, width, channels, ph, pw,
                                          sampling_ratio, rois, out);
}
void ref_deform_im2col(void *stream, const float *im, const float *off, int channels, int height, int width, int kh, int kw,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int parallel_imgs, int dg,
                       float *col)
{
    deformable_im2col_gpu_kernel_launcher((hipStream_t)stream, im, off, channels, height, width, kh, kw, pad_h, pad_w,
                                          stride_h, stride_w, dil_h, dil_w, parallel_imgs, dg, col);
}
void ref_mod_deform_im2col(void *stream, const float *im, const float *off, const float *mask, int batch, int channels,
                           int height, int width, int height_col, int width_col, int kh, int kw, int pad_h, int pad_w,
                           int stride_h, int stride_w, int dil_h, int dil_w, int dg, float *col)
{
    modulated_deformable_im2col_gpu_kernel_launcher((hipStream_t)stream, im, off, mask, batch, channels, height, width,
                                                    height_col, width_col, kh, kw, pad_h, pad_w, stride_h, stride_w,
                                                    dil_h, dil_w, dg, col);
}
// backward natives (roi_align_cuda.cpp:32-36, deform_conv_cuda.cpp:32-46, mod_deform_conv_cuda.cpp:33-50)
void ref_roi_align_backward(void *stream, const float *top_diff, float scale, int batch, int num_rois, int height, int width,
                            int channels, int ph, int pw, int sampling_ratio, const float *rois, float *bottom_diff)
{
    roi_align_backward_gpu_kernel_launcher((hipStream_t)stream, top_diff, scale, batch, num_rois, height, width, channels, ph, pw,
                                           sampling_ratio, rois, bottom_diff);
}
void ref_deform_col2im(void *stream, const float *col, const float *off, int channels, int height, int width, int kh, int kw,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int parallel_imgs, int dg,
                       float *grad_im)
{
    deformable_col2im_gpu_kernel_launcher((hipStream_t)stream, col, off, channels, height, width, kh, kw, pad_h, pad_w, stride_h,
                                          stride_w, dil_h, dil_w, parallel_imgs, dg, grad_im);
}
void ref_deform_col2im_coord(void *stream, const float *col, const float *im, const float *off, int channels, int height,
                             int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                             int parallel_imgs, int dg, float *grad_off)
{
    deformable_col2im_coord_gpu_kernel_launcher((hipStream_t)stream, col, im, off, channels, height, width, kh, kw, pad_h, pad_w,
                                                stride_h, stride_w, dil_h, dil_w, parallel_imgs, dg, grad_off);
}
void ref_mod_deform_col2im(void *stream, const float *col, const float *off, const float *mask, int batch, int channels,
                           int height, int width, int height_col, int width_col, int kh, int kw, int pad_h, int pad_w,
                           int stride_h, int stride_w, int dil_h, int dil_w, int dg, float *grad_im)
{
    modulated_deformable_col2im_gpu_kernel_launcher((hipStream_t)stream, col, off, mask, batch, channels, height, width,
                                                    height_col, width_col, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h,
                                                    dil_w, dg, grad_im);
}
void ref_mod_deform_col2im_coord(void *stream, const float *col, const float *im, const float *off, const float *mask, int batch,
                                 int channels, int height, int width, int height_col, int width_col, int kh, int kw, int pad_h,
                                 int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float *grad_off,
                                 float *grad_mask)
{
    modulated_deformable_col2im_coord_gpu_kernel_launcher((hipStream_t)stream, col, im, off, mask, batch, channels, height, width,
                                                          height_col, width_col, kh, kw, pad_h, pad_w, stride_h, stride_w,
                                                          dil_h, dil_w, dg, grad_off, grad_mask);
}
void ref_nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim, float thresh, int device_id)
{
    _nms(keep_out, num_out, boxes_host, boxes_num, boxes_dim, thresh, device_id);
}
}
