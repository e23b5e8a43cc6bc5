// backward.hip -- the backward natives of the reference's operator extensions: API parity for
// fine-tuning, not part of the inference path. NCHW fp32 layouts and argument order of the reference launchers:
//   roi_align_backward_gpu_kernel_launcher          upsnet/operators/src/roi_align_kernel.cu:238-349,375-398
//   deformable_col2im / col2im_coord launchers       deform_conv_kernel.cu:293-383, 391-500
//   modulated_deformable_col2im / col2im_coord       mod_deform_conv_kernel.cu:251-381, 409-460
// One thread per top element (col2im, ROIAlign: scatter with fp32 atomics, like the reference -- the summation order of
// colliding contributions is therefore not fixed and parity is 1e-5-level, not bit-exact) or per offset element (col2im_coord:
// gather, fixed order, bit-exact against the oracle). Arithmetic per contribution follows the reference expression by expression.
#include "common.h"
#include "upsnet_hip.h"

// ---- ROIAlign: gradient w.r.t. the features
struct RabTap { int y_low, y_high, x_low, x_high; float w1, w2, w3, w4; bool ok; };

__device__ static inline RabTap rab_tap(const int height, const int width, float y, float x)   // bilinear_interpolate_gradient (:97-160)
{
    RabTap t;
    t.ok = !(y < -1.0f || y > (float)height || x < -1.0f || x > (float)width);
    t.y_low = t.y_high = t.x_low = t.x_high = -1;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    if (!t.ok) return t;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    t.y_low = (int)y; t.x_low = (int)x;
    if (t.y_low >= height - 1) { t.y_high = t.y_low = height - 1; y = (float)t.y_low; } else t.y_high = t.y_low + 1;
    if (t.x_low >= width - 1) { t.x_high = t.x_low = width - 1; x = (float)t.x_low; } else t.x_high = t.x_low + 1;
    const float ly = y - (float)t.y_low, lx = x - (float)t.x_low;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    return t;
}

__global__ void __launch_bounds__(256)
roi_align_backward_kernel(const long n, const float *__restrict__ top_diff, const float spatial_scale, const int channels, const int height,
                          const int width, const int ph_n, const int pw_n, const int sampling_ratio, const float *__restrict__ rois,
                          float *__restrict__ bottom_diff)
{
    for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)blockDim.x * gridDim.x) {
        const int pw = index % pw_n, ph = (index / pw_n) % ph_n;
        const int c = (index / pw_n / ph_n) % channels;
        const int r = index / pw_n / ph_n / channels;
        const float *roi = rois + (long)r * 5;
        const int b = (int)roundf(roi[0]);
        const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
        const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
        const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
        const float bin_h = roi_height / (float)ph_n, bin_w = roi_width / (float)pw_n;
        float *plane = bottom_diff + ((long)b * channels + c) * height * width;
        const float top = top_diff[index];
        const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)ph_n);
        const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pw_n);
        const float count = (float)(gh * gw);
        for (int iy = 0; iy < gh; ++iy) {
            const float y = roi_start_h + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = roi_start_w + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
                const RabTap t = rab_tap(height, width, y, x);
                if (t.x_low >= 0 && t.x_high >= 0 && t.y_low >= 0 && t.y_high >= 0) {
                    atomicAdd(plane + t.y_low * width + t.x_low, top * t.w1 / count);
                    atomicAdd(plane + t.y_low * width + t.x_high, top * t.w2 / count);
                    atomicAdd(plane + t.y_high * width + t.x_low, top * t.w3 / count);
                    atomicAdd(plane + t.y_high * width + t.x_high, top * t.w4 / count);
                }
            }
        }
    }
}

extern "C" int upsnet_roi_align_backward(void *stream, const float *top_diff, float spatial_scale, int batch_size, int num_rois, int height,
                                         int width, int channels, int pooled_height, int pooled_width, int sampling_ratio,
                                         const float *rois, float *bottom_diff)
{
    (void)batch_size;
    UPS_REQUIRE(top_diff && rois && bottom_diff, "roi_align_backward: null pointer");
    const long n = (long)num_rois * channels * pooled_height * pooled_width;
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(roi_align_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, top_diff, spatial_scale, channels, height,
                       width, pooled_height, pooled_width, sampling_ratio, rois, bottom_diff);
    UPS_CHECK_LAUNCH("roi_align_backward_kernel");
    return 0;
}

// ---- deformable convolution: column gradient -> input gradient (scatter)
// weight of integer pixel (h, w) for the sample at (ph, pw): get_gradient_weight (deform_conv_kernel.cu:120-143)
__device__ static inline float dcb_grad_weight(const float ph, const float pw, const int h, const int w, const int height, const int width)
{
    if (ph <= -1 || ph >= (float)height || pw <= -1 || pw >= (float)width) return 0.f;
    const int h_low = (int)floorf(ph), w_low = (int)floorf(pw);
    const int h_high = h_low + 1, w_high = w_low + 1;
    float weight = 0.f;
    if (h == h_low && w == w_low) weight = ((float)(h + 1) - ph) * ((float)(w + 1) - pw);
    if (h == h_low && w == w_high) weight = ((float)(h + 1) - ph) * (pw + 1.0f - (float)w);
    if (h == h_high && w == w_low) weight = (ph + 1.0f - (float)h) * ((float)(w + 1) - pw);
    if (h == h_high && w == w_high) weight = (ph + 1.0f - (float)h) * (pw + 1.0f - (float)w);
    return weight;
}

template <bool MOD>
__global__ void __launch_bounds__(256)
deform_col2im_kernel(const long n, const float *__restrict__ data_col, const float *__restrict__ data_offset,
                     const float *__restrict__ data_mask, const int channels, const int height, const int width, const int kh, const int kw,
                     const int pad_h, const int pad_w, const int stride_h, const int stride_w, const int dil_h, const int dil_w,
                     const int cpg, const int batch_size, const int deformable_group, const int height_col, const int width_col,
                     float *__restrict__ grad_im)
{
    const long plane_col = (long)height_col * width_col;
    for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)blockDim.x * gridDim.x) {
        // data_col is [C, kh, kw, B, Hc, Wc]
        const int w_out = index % width_col, h_out = (index / width_col) % height_col;
        const int b = (index / plane_col) % batch_size;
        const long t = index / plane_col / batch_size;
        const int j = t % kw, i = (t / kw) % kh;
        const int c = (int)(t / kw / kh);
        const int g = c / cpg;
        const float *off = data_offset + ((long)b * deformable_group + g) * 2 * kh * kw * plane_col;
        const long pix = (long)h_out * width_col + w_out;
        const float off_h = off[(long)(2 * (i * kw + j)) * plane_col + pix], off_w = off[(long)(2 * (i * kw + j) + 1) * plane_col + pix];
        const float ph = (float)(h_out * stride_h - pad_h + i * dil_h) + off_h;
        const float pw = (float)(w_out * stride_w - pad_w + j * dil_w) + off_w;
        float top = data_col[index];
        if (MOD) top = top * data_mask[(((long)b * deformable_group + g) * kh * kw + (i * kw + j)) * plane_col + pix];
        // the integer pixels within distance < 1 of the sample (the reference scans a 5x5 window around the truncated position)
        const int hb = (int)floorf(ph), wb = (int)floorf(pw);
        for (int dy = 0; dy <= 1; ++dy)
            for (int dx = 0; dx <= 1; ++dx) {
                const int h = hb + dy, w = wb + dx;
                if (h >= 0 && h < height && w >= 0 && w < width && fabsf(ph - (float)h) < 1 && fabsf(pw - (float)w) < 1) {
                    const float weight = dcb_grad_weight(ph, pw, h, w, height, width);
                    atomicAdd(grad_im + (((long)b * channels + c) * height + h) * width + w, weight * top);
                }
            }
    }
}

// ---- deformable convolution: column gradient -> offset (and mask) gradient (gather)
// d(sample)/d(ph) (bp_dir 0) or d(sample)/d(pw) (bp_dir 1): get_coordinate_weight (deform_conv_kernel.cu:146-184)
__device__ static inline float dcb_coord_weight(const float ph, const float pw, const int height, const int width,
                                                const float *__restrict__ im, const int bp_dir)
{
    if (ph <= -1 || ph >= (float)height || pw <= -1 || pw >= (float)width) return 0.f;
    const int h_low = (int)floorf(ph), w_low = (int)floorf(pw);
    const int h_high = h_low + 1, w_high = w_low + 1;
    float weight = 0.f;
    if (bp_dir == 0) {
        if (h_low >= 0 && w_low >= 0) weight += -1.0f * ((float)(w_low + 1) - pw) * im[h_low * width + w_low];
        if (h_low >= 0 && w_high <= width - 1) weight += -1.0f * (pw - (float)w_low) * im[h_low * width + w_high];
        if (h_high <= height - 1 && w_low >= 0) weight += ((float)(w_low + 1) - pw) * im[h_high * width + w_low];
        if (h_high <= height - 1 && w_high <= width - 1) weight += (pw - (float)w_low) * im[h_high * width + w_high];
    } else {
        if (h_low >= 0 && w_low >= 0) weight += -1.0f * ((float)(h_low + 1) - ph) * im[h_low * width + w_low];
        if (h_low >= 0 && w_high <= width - 1) weight += ((float)(h_low + 1) - ph) * im[h_low * width + w_high];
        if (h_high <= height - 1 && w_low >= 0) weight += -1.0f * (ph - (float)h_low) * im[h_high * width + w_low];
        if (h_high <= height - 1 && w_high <= width - 1) weight += (ph - (float)h_low) * im[h_high * width + w_high];
    }
    return weight;
}

__device__ static inline float dcb_bilinear(const float *__restrict__ plane, const int height, const int width, const float h, const float w)
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
    float val = (hh * hw) * v1;
    val = val + (hh * lw) * v2;
    val = val + (lh * hw) * v3;
    val = val + (lh * lw) * v4;
    return val;
}

template <bool MOD>
__global__ void __launch_bounds__(256)
deform_col2im_coord_kernel(const long n, const float *__restrict__ data_col, const float *__restrict__ data_im,
                           const float *__restrict__ data_offset, const float *__restrict__ data_mask, const int channels, const int height,
                           const int width, const int kh, const int kw, const int pad_h, const int pad_w, const int stride_h,
                           const int stride_w, const int dil_h, const int dil_w, const int batch_size, const int deformable_group,
                           const int height_col, const int width_col, float *__restrict__ grad_offset, float *__restrict__ grad_mask)
{
    const long plane_col = (long)height_col * width_col;
    const int taps = kh * kw, offset_channels = 2 * taps * deformable_group;
    const int cpg_im = channels / deformable_group;   // image channels per deformable group
    for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)blockDim.x * gridDim.x) {
        const int w = index % width_col, h = (index / width_col) % height_col;
        const int c = (index / plane_col) % offset_channels;
        const int b = (int)(index / plane_col / offset_channels);
        const int g = c / (2 * taps);
        const int oc = c - g * 2 * taps;             // offset channel inside the group: 2*tap + (0: h, 1: w)
        const int tap = oc >> 1, bp_dir = oc & 1;
        const int i = tap / kw, j = tap - i * kw;
        const long pix = (long)h * width_col + w;
        const float *off = data_offset + ((long)b * deformable_group + g) * 2 * taps * plane_col;
        const float off_h = off[(long)(2 * tap) * plane_col + pix], off_w = off[(long)(2 * tap + 1) * plane_col + pix];
        float ph = (float)(h * stride_h - pad_h + i * dil_h) + off_h;
        float pw = (float)(w * stride_w - pad_w + j * dil_w) + off_w;
        const bool outside = ph <= -1 || pw <= -1 || ph >= (float)height || pw >= (float)width;
        if (outside) ph = pw = -2.f;
        const float m = MOD ? data_mask[(((long)b * deformable_group + g) * taps + tap) * plane_col + pix] : 1.f;
        float val = 0.f, mval = 0.f;
        // every image channel of the group contributes through its column row (channel, tap)
        for (int cc = 0; cc < cpg_im; ++cc) {
            const int c_im = g * cpg_im + cc;
            const float col = data_col[(((long)(c_im * taps + tap)) * batch_size + b) * plane_col + pix];
            const float *plane = data_im + ((long)b * channels + c_im) * height * width;
            if (MOD && !outside) mval += col * dcb_bilinear(plane, height, width, ph, pw);
            const float weight = dcb_coord_weight(ph, pw, height, width, plane, bp_dir);
            if (MOD) val += weight * col * m;
            else val += weight * col;
        }
        grad_offset[index] = val;
        if (MOD && bp_dir == 0) grad_mask[(((long)b * deformable_group + g) * taps + tap) * plane_col + pix] = mval;
    }
}

static int dcb_geometry(int height, int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                        int *hc, int *wc)
{
    *hc = (height + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    *wc = (width + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    return *hc > 0 && *wc > 0;
}

static inline int dcb_blocks(long n) { long b = (n + 255) / 256; return (int)(b > 65535 ? 65535 : b); }

extern "C" int upsnet_deform_col2im(void *stream, const float *data_col, const float *data_offset, int channels, int height, int width,
                                    int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                    int parallel_imgs, int deformable_group, float *grad_im)
{
    UPS_REQUIRE(data_col && data_offset && grad_im, "deform_col2im: null pointer");
    UPS_REQUIRE(deformable_group >= 1 && channels % deformable_group == 0, "deform_col2im: channels %% deformable_group != 0");
    int hc, wc;
    UPS_REQUIRE(dcb_geometry(height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, &hc, &wc), "deform_col2im: empty output");
    const long n = (long)channels * kh * kw * hc * wc * parallel_imgs;
    hipLaunchKernelGGL(deform_col2im_kernel<false>, dim3(dcb_blocks(n)), dim3(256), 0, (hipStream_t)stream, n, data_col, data_offset, nullptr,
                       channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, channels / deformable_group,
                       parallel_imgs, deformable_group, hc, wc, grad_im);
    UPS_CHECK_LAUNCH("deform_col2im_kernel");
    return 0;
}

extern "C" int upsnet_deform_col2im_coord(void *stream, const float *data_col, const float *data_im, const float *data_offset, int channels,
                                          int height, int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                          int dil_h, int dil_w, int parallel_imgs, int deformable_group, float *grad_offset)
{
    UPS_REQUIRE(data_col && data_im && data_offset && grad_offset, "deform_col2im_coord: null pointer");
    UPS_REQUIRE(deformable_group >= 1 && channels % deformable_group == 0, "deform_col2im_coord: channels %% deformable_group != 0");
    int hc, wc;
    UPS_REQUIRE(dcb_geometry(height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, &hc, &wc), "deform_col2im_coord: empty output");
    const long n = (long)hc * wc * 2 * kh * kw * deformable_group * parallel_imgs;
    hipLaunchKernelGGL(deform_col2im_coord_kernel<false>, dim3(dcb_blocks(n)), dim3(256), 0, (hipStream_t)stream, n, data_col, data_im,
                       data_offset, nullptr, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, parallel_imgs,
                       deformable_group, hc, wc, grad_offset, nullptr);
    UPS_CHECK_LAUNCH("deform_col2im_coord_kernel");
    return 0;
}

extern "C" int upsnet_mod_deform_col2im(void *stream, const float *data_col, const float *data_offset, const float *data_mask, int batch_size,
                                        int channels, int height_im, int width_im, int height_col, int width_col, int kh, int kw, int pad_h,
                                        int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group, float *grad_im)
{
    UPS_REQUIRE(data_col && data_offset && data_mask && grad_im, "mod_deform_col2im: null pointer");
    UPS_REQUIRE(deformable_group >= 1 && channels % deformable_group == 0, "mod_deform_col2im: channels %% deformable_group != 0");
    const long n = (long)channels * kh * kw * batch_size * height_col * width_col;
    // the reference launcher hands pad_h to the kernel for BOTH paddings (mod_deform_conv_kernel.cu:423); mirrored, so a
    // caller with pad_h != pad_w gets the reference's result, not the mathematically intended one
    (void)pad_w;
    hipLaunchKernelGGL(deform_col2im_kernel<true>, dim3(dcb_blocks(n)), dim3(256), 0, (hipStream_t)stream, n, data_col, data_offset, data_mask,
                       channels, height_im, width_im, kh, kw, pad_h, pad_h, stride_h, stride_w, dil_h, dil_w, channels / deformable_group,
                       batch_size, deformable_group, height_col, width_col, grad_im);
    UPS_CHECK_LAUNCH("deform_col2im_kernel<mod>");
    return 0;
}

extern "C" int upsnet_mod_deform_col2im_coord(void *stream, const float *data_col, const float *data_im, const float *data_offset,
                                              const float *data_mask, int batch_size, int channels, int height_im, int width_im,
                                              int height_col, int width_col, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                              int stride_w, int dil_h, int dil_w, int deformable_group, float *grad_offset, float *grad_mask)
{
    UPS_REQUIRE(data_col && data_im && data_offset && data_mask && grad_offset && grad_mask, "mod_deform_col2im_coord: null pointer");
    UPS_REQUIRE(deformable_group >= 1 && channels % deformable_group == 0, "mod_deform_col2im_coord: channels %% deformable_group != 0");
    const long n = (long)batch_size * height_col * width_col * 2 * kh * kw * deformable_group;
    hipLaunchKernelGGL(deform_col2im_coord_kernel<true>, dim3(dcb_blocks(n)), dim3(256), 0, (hipStream_t)stream, n, data_col, data_im,
                       data_offset, data_mask, channels, height_im, width_im, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                       batch_size, deformable_group, height_col, width_col, grad_offset, grad_mask);
    UPS_CHECK_LAUNCH("deform_col2im_coord_kernel<mod>");
    return 0;
}
