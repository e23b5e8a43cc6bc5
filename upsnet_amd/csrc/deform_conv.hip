// deform_conv.hip -- NCHW drop-ins of the reference's deformable im2col launchers (v1 / v2) for gfx950.
//
//  * upsnet_deform_im2col / upsnet_mod_deform_im2col keep the reference launchers' signatures and the column-buffer layout
//    [C * kh * kw][batch][Ho][Wo] (deform_conv_cuda.cpp:25-47, mod_deform_conv_cuda.cpp:28-49); the arithmetic of one column
//    element is the reference's (deform_conv_kernel.cu:88-118,227-240; mod_deform_conv_kernel.cu:87-116,187-249) -- bit-exact.
//  * the decomposition is this file's own: the sampling position of a (pixel, tap) pair does not depend on the channel, so a
//    workgroup owns 64 consecutive output pixels of one image and one deformable group, tabulates the 64 x kh*kw samples ONCE in
//    LDS (four corner offsets inside a plane, -1 for a corner outside the image; four bilinear weights, all 0 for a sample outside
//    the image; the v2 modulation), and its four waves then walk the channels of the group: lane = pixel, so every column write is
//    a 256-byte run. (The reference recomputes offsets, floors, weights and validity for every channel.)
//  * the MI355X-native operator -- sampling + MFMA GEMM in one kernel, no column buffer -- is deform_fused.hip; this file exists for
//    callers of the reference's native API and for geometries the fused kernel does not take (deformable_groups > 1, Cin % 32 != 0).
#include "common.h"
#include "upsnet_hip.h"

struct DcnSample {
    int o[4];       // element offsets of the four corners inside a channel plane (-1: outside the image, contributes 0)
    float w[4];     // hh*hw, hh*lw, lh*hw, lh*lw (deform_conv_kernel.cu:111); all 0 when the sample itself is outside
    float m;        // v2 modulation (1 for v1)
};

#define DCN_TP 64   // output pixels per workgroup

template <bool MOD>
__global__ void __launch_bounds__(256)
deform_im2col_nchw_kernel(const float *__restrict__ data_im, const float *__restrict__ data_offset, const float *__restrict__ data_mask,
                          const int height, const int width, const int kh, const int kw, const int pad_h, const int pad_w,
                          const int stride_h, const int stride_w, const int dil_h, const int dil_w, const int cpg, const int batch_size,
                          const int num_channels, const int deformable_group, const int height_col, const int width_col,
                          float *__restrict__ data_col)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DcnSample *tab = reinterpret_cast<DcnSample *>(smem_raw);          // [tap][pixel]
    const int taps = kh * kw;
    const long plane_col = (long)height_col * width_col;
    const long pix0 = (long)blockIdx.x * DCN_TP;
    const int b = blockIdx.y / deformable_group, g = blockIdx.y - b * deformable_group;
    const float *off_ptr = data_offset + ((long)b * deformable_group + g) * 2 * taps * plane_col;
    const float *mask_ptr = MOD ? data_mask + ((long)b * deformable_group + g) * taps * plane_col : nullptr;
    for (int idx = threadIdx.x; idx < taps * DCN_TP; idx += blockDim.x) {
        const int t = idx / DCN_TP, px = idx - t * DCN_TP;
        const long pix = pix0 + px;
        DcnSample sm;
        sm.o[0] = sm.o[1] = sm.o[2] = sm.o[3] = -1;
        sm.w[0] = sm.w[1] = sm.w[2] = sm.w[3] = 0.f;
        sm.m = 1.f;
        if (pix < plane_col) {
            const int h_col = (int)(pix / width_col), w_col = (int)(pix - (long)h_col * width_col);
            const int i = t / kw, j = t - i * kw;
            const float h_im = (float)(h_col * stride_h - pad_h + i * dil_h) + off_ptr[(long)(2 * t) * plane_col + pix];
            const float w_im = (float)(w_col * stride_w - pad_w + j * dil_w) + off_ptr[(long)(2 * t + 1) * plane_col + pix];
            if (h_im > -1 && w_im > -1 && h_im < (float)height && w_im < (float)width) {
                const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                const int h_high = h_low + 1, w_high = w_low + 1;
                const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                const float hh = 1.0f - lh, hw = 1.0f - lw;
                if (h_low >= 0 && w_low >= 0) sm.o[0] = h_low * width + w_low;
                if (h_low >= 0 && w_high <= width - 1) sm.o[1] = h_low * width + w_high;
                if (h_high <= height - 1 && w_low >= 0) sm.o[2] = h_high * width + w_low;
                if (h_high <= height - 1 && w_high <= width - 1) sm.o[3] = h_high * width + w_high;
                sm.w[0] = hh * hw; sm.w[1] = hh * lw; sm.w[2] = lh * hw; sm.w[3] = lh * lw;
            }
            if (MOD) sm.m = mask_ptr[(long)t * plane_col + pix];
        }
        tab[idx] = sm;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const long pix = pix0 + lane;
    if (pix >= plane_col) return;
    for (int cg = wave; cg < cpg; cg += nwaves) {
        const int c_im = g * cpg + cg;
        const float *plane = data_im + ((long)b * num_channels + c_im) * height * width;
        float *col = data_col + (((long)c_im * taps) * batch_size + b) * plane_col + pix;
        for (int t = 0; t < taps; ++t) {
            const DcnSample sm = tab[t * DCN_TP + lane];
            const float v1 = sm.o[0] >= 0 ? plane[sm.o[0]] : 0.f, v2 = sm.o[1] >= 0 ? plane[sm.o[1]] : 0.f;
            const float v3 = sm.o[2] >= 0 ? plane[sm.o[2]] : 0.f, v4 = sm.o[3] >= 0 ? plane[sm.o[3]] : 0.f;
            float val = sm.w[0] * v1;
            val = val + sm.w[1] * v2;
            val = val + sm.w[2] * v3;
            val = val + sm.w[3] * v4;
            if (MOD) val = val * sm.m;
            col[(long)t * batch_size * plane_col] = val;
        }
    }
}

static int deform_im2col_launch(bool mod, void *stream, const float *data_im, const float *data_offset, const float *data_mask, int batch,
                                int channels, int height, int width, int height_col, int width_col, int kh, int kw, int pad_h, int pad_w,
                                int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group, float *data_col)
{
    const long plane_col = (long)height_col * width_col;
    if (plane_col == 0 || batch == 0) return 0;
    const size_t smem = (size_t)kh * kw * DCN_TP * sizeof(DcnSample);
    UPS_REQUIRE(smem <= 160 * 1024 - 1024, "deform_im2col: kernel of %dx%d taps exceeds the sampling table", kh, kw);
    if (smem > 64 * 1024) {
        const void *fn = mod ? reinterpret_cast<const void *>(&deform_im2col_nchw_kernel<true>) : reinterpret_cast<const void *>(&deform_im2col_nchw_kernel<false>);
        UPS_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    const dim3 grid((unsigned)((plane_col + DCN_TP - 1) / DCN_TP), (unsigned)(batch * deformable_group));
    if (mod)
        hipLaunchKernelGGL(deform_im2col_nchw_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, data_im, data_offset, data_mask, height,
                           width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, channels / deformable_group, batch, channels,
                           deformable_group, height_col, width_col, data_col);
    else
        hipLaunchKernelGGL(deform_im2col_nchw_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, data_im, data_offset, data_mask, height,
                           width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, channels / deformable_group, batch, channels,
                           deformable_group, height_col, width_col, data_col);
    UPS_CHECK_LAUNCH("deform_im2col_nchw_kernel");
    return 0;
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
    return deform_im2col_launch(false, stream, data_im, data_offset, nullptr, parallel_imgs, channels, height, width, height_col, width_col,
                                ksize_h, ksize_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w, deformable_group, data_col);
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
    return deform_im2col_launch(true, stream, data_im, data_offset, data_mask, batch_size, channels, height_im, width_im, height_col, width_col,
                                kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w, deformable_group, data_col);
}
