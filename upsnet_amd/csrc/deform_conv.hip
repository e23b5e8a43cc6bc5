// deform_conv.hip -- deformable convolution v1 / v2 forward for gfx950.
//
//  * upsnet_deform_im2col / upsnet_mod_deform_im2col : NCHW drop-ins for the reference launchers
//    (deform_conv_kernel.cu:194-285, mod_deform_conv_kernel.cu:187-249,383-407): same column-buffer
//    layout, one thread per (c, b, h_col, w_col).
//  * the MI355X-native fused operator (upsnet_deform_conv_forward_nhwc: sampling + fp32 MFMA GEMM, no column
//    buffer) lives in conv.hip: it is the dense implicit-GEMM kernel with a bilinear-gather A-operand.
#include "common.h"
#include "upsnet_hip.h"

// ---------------------------------------------------------------------------------------------
// exact bilinear sample (deform_conv_kernel.cu:88-118)
__device__ static inline float dcn_bilinear(const float *__restrict__ plane, const int height, const int width,
                                            const float h, const float w)
{
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h - (float)h_low, lw = w - (float)w_low;
    const float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * width + w_low];
    if (h_low >= 0 && w_high <= width - 1) v2 = plane[h_low * width + w_high];
    if (h_high <= height - 1 && w_low >= 0) v3 = plane[h_high * width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = plane[h_high * width + w_high];
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    float val = w1 * v1;
    val = val + w2 * v2;
    val = val + w3 * v3;
    val = val + w4 * v4;
    return val;
}

template <bool MOD>
__global__ void __launch_bounds__(256)
deform_im2col_nchw_kernel(const long n, const float *__restrict__ data_im, const float *__restrict__ data_offset,
                          const float *__restrict__ data_mask, const int height, const int width, const int kh,
                          const int kw, const int pad_h, const int pad_w, const int stride_h, const int stride_w,
                          const int dil_h, const int dil_w, const int cpg, const int batch_size, const int num_channels,
                          const int deformable_group, const int height_col, const int width_col,
                          float *__restrict__ data_col)
{
    for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)blockDim.x * gridDim.x) {
        const int w_col = index % width_col;
        const int h_col = (index / width_col) % height_col;
        const int b_col = (index / width_col / height_col) % batch_size;
        const int c_im = (index / width_col / height_col) / batch_size;
        const int c_col = c_im * kh * kw;
        const int g = c_im / cpg;
        const int h_in = h_col * stride_h - pad_h, w_in = w_col * stride_w - pad_w;
        const long plane_col = (long)height_col * width_col;
        float *col_ptr = data_col + (((long)c_col * batch_size + b_col) * height_col + h_col) * width_col + w_col;
        const float *im_ptr = data_im + ((long)b_col * num_channels + c_im) * height * width;
        const float *off_ptr = data_offset + ((long)b_col * deformable_group + g) * 2 * kh * kw * plane_col;
        const float *mask_ptr = MOD ? data_mask + ((long)b_col * deformable_group + g) * kh * kw * plane_col : nullptr;
        const long pix = (long)h_col * width_col + w_col;
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                const float off_h = off_ptr[(long)(2 * (i * kw + j)) * plane_col + pix];
                const float off_w = off_ptr[(long)(2 * (i * kw + j) + 1) * plane_col + pix];
                const float h_im = (float)(h_in + i * dil_h) + off_h;
                const float w_im = (float)(w_in + j * dil_w) + off_w;
                float val = 0.f;
                if (h_im > -1 && w_im > -1 && h_im < (float)height && w_im < (float)width)
                    val = dcn_bilinear(im_ptr, height, width, h_im, w_im);
                if (MOD) val = val * mask_ptr[(long)(i * kw + j) * plane_col + pix];
                *col_ptr = val;
                col_ptr += (long)batch_size * plane_col;
            }
    }
}

extern "C" int upsnet_deform_im2col(void *stream, const float *data_im, const float *data_offset, int channels,
                                    int height, int width, int ksize_h, int ksize_w, int pad_h, int pad_w,
                                    int stride_h, int stride_w, int dilation_h, int dilation_w, int parallel_imgs,
                                    int deformable_group, float *data_col)
{
    UPS_REQUIRE(data_im && data_offset && data_col, "deform_im2col: null pointer");
    UPS_REQUIRE(channels > 0 && deformable_group > 0 && channels % deformable_group == 0 && parallel_imgs > 0,
                "deform_im2col: bad channels/deformable_group/parallel_imgs");
    const int height_col = (height + 2 * pad_h - (dilation_h * (ksize_h - 1) + 1)) / stride_h + 1;
    const int width_col = (width + 2 * pad_w - (dilation_w * (ksize_w - 1) + 1)) / stride_w + 1;
    UPS_REQUIRE(height_col > 0 && width_col > 0, "deform_im2col: empty output");
    const long n = (long)channels * height_col * width_col * parallel_imgs;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(deform_im2col_nchw_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, data_im,
                       data_offset, (const float *)nullptr, height, width, ksize_h, ksize_w, pad_h, pad_w, stride_h,
                       stride_w, dilation_h, dilation_w, channels / deformable_group, parallel_imgs, channels,
                       deformable_group, height_col, width_col, data_col);
    UPS_CHECK_LAUNCH("deform_im2col_nchw_kernel");
    return 0;
}

extern "C" int upsnet_mod_deform_im2col(void *stream, const float *data_im, const float *data_offset,
                                        const float *data_mask, int batch_size, int channels, int height_im,
                                        int width_im, int height_col, int width_col, int kernel_h, int kernel_w,
                                        int pad_h, int pad_w, int stride_h, int stride_w, int dilation_h, int dilation_w,
                                        int deformable_group, float *data_col)
{
    UPS_REQUIRE(data_im && data_offset && data_mask && data_col, "mod_deform_im2col: null pointer");
    UPS_REQUIRE(channels > 0 && deformable_group > 0 && channels % deformable_group == 0 && batch_size > 0,
                "mod_deform_im2col: bad channels/deformable_group/batch");
    const long n = (long)channels * batch_size * height_col * width_col;
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(deform_im2col_nchw_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, data_im,
                       data_offset, data_mask, height_im, width_im, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                       dilation_h, dilation_w, channels / deformable_group, batch_size, channels, deformable_group,
                       height_col, width_col, data_col);
    UPS_CHECK_LAUNCH("mod_deform_im2col_nchw_kernel");
    return 0;
}

