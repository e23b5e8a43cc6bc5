// preprocess.hip -- the step before the hot path: image -> network input blob.
//
// Reference: BaseDataset.prep_im_for_blob + im_list_to_blob (upsnet/dataset/base_dataset.py:143-173, 898-923), numpy + cv2 on
// the host: uint8 BGR HWC -> float32, subtract pixel_means (float64 array, so the subtraction happens in double and is rounded
// to float), cv2.resize(fx = fy = im_scale, INTER_LINEAR) of the float image, HWC -> CHW, zero-pad bottom/right to a multiple
// of the coarsest FPN stride; then a 25 MB fp32 host-to-device copy.
// Here: ONE kernel from the uint8 image (6 MB on the wire) to the padded fp32 blob, either planar NCHW (the reference's blob
// layout) or NHWC with 4 channels (RGB + a zero channel), the layout the stem convolution (conv.hip, loader mode 3) consumes.
// cv2 is not available for pinning: the INTER_LINEAR formula is OpenCV's published one (resize.cpp: scale = 1/fx,
// fx_d = (float)((d + 0.5) * scale - 0.5), floor, clamp, fp32 weights (1-f, f), horizontal pass then vertical pass), the same
// restatement as the mask paste of panoptic.hip. For im_scale == 1 (Cityscapes) the formula degenerates to an exact copy.
#include "common.h"
#include "upsnet_hip.h"

struct LinCoef {
    int s0, s1;
    float f;
};

__device__ static inline LinCoef prep_lin_coef(int d, double scale, int ssize)
{
    LinCoef c;
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    c.s0 = sx;
    c.s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
    c.f = fx;
    return c;
}

__global__ void __launch_bounds__(256)
prep_image_u8_kernel(const unsigned char *__restrict__ im, const int H, const int W, const double m0, const double m1,
                     const double m2, const double inv_scale, const int Hr, const int Wr, const int Hp, const int Wp,
                     const int nhwc4, float *__restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Hp * Wp) return;
    const int x = (int)(idx % Wp), y = (int)(idx / Wp);
    float v[3] = {0.f, 0.f, 0.f};
    if (y < Hr && x < Wr) {
        const LinCoef cx = prep_lin_coef(x, inv_scale, W), cy = prep_lin_coef(y, inv_scale, H);
        const float a0 = 1.0f - cx.f, a1 = cx.f, b0 = 1.0f - cy.f, b1 = cy.f;
        const double mean[3] = {m0, m1, m2};
        const unsigned char *p00 = im + ((long)cy.s0 * W + cx.s0) * 3, *p01 = im + ((long)cy.s0 * W + cx.s1) * 3;
        const unsigned char *p10 = im + ((long)cy.s1 * W + cx.s0) * 3, *p11 = im + ((long)cy.s1 * W + cx.s1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s00 = (float)((double)p00[c] - mean[c]), s01 = (float)((double)p01[c] - mean[c]);
            const float s10 = (float)((double)p10[c] - mean[c]), s11 = (float)((double)p11[c] - mean[c]);
            const float r0 = s00 * a0 + s01 * a1;
            const float r1 = s10 * a0 + s11 * a1;
            v[c] = r0 * b0 + r1 * b1;
        }
    }
    if (nhwc4) {
        reinterpret_cast<float4 *>(out)[idx] = make_float4(v[0], v[1], v[2], 0.f);
    } else {
        const long plane = (long)Hp * Wp;
        out[idx] = v[0]; out[plane + idx] = v[1]; out[2 * plane + idx] = v[2];
    }
}

extern "C" int upsnet_prep_image_u8(void *stream, const unsigned char *image_hwc, int height, int width, const double pixel_means[3],
                                    double im_scale, int resized_h, int resized_w, int padded_h, int padded_w, int nhwc4, float *blob)
{
    UPS_REQUIRE(image_hwc && pixel_means && blob, "prep_image_u8: null pointer");
    UPS_REQUIRE(height > 0 && width > 0 && im_scale > 0, "prep_image_u8: bad image shape / scale");
    UPS_REQUIRE(resized_h > 0 && resized_w > 0 && padded_h >= resized_h && padded_w >= resized_w, "prep_image_u8: bad output shape");
    const long n = (long)padded_h * padded_w;
    hipLaunchKernelGGL(prep_image_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, image_hwc, height,
                       width, pixel_means[0], pixel_means[1], pixel_means[2], 1.0 / im_scale, resized_h, resized_w, padded_h, padded_w,
                       nhwc4, blob);
    UPS_CHECK_LAUNCH("prep_image_u8_kernel");
    return 0;
}

// layout plumbing for callers that hand over the reference's fp32 NCHW blob: [N,3,H,W] -> [N,H,W,4] (4th channel zero)
__global__ void __launch_bounds__(256)
image_to_nhwc4_kernel(const float *__restrict__ x, const int C, const long plane, const long total, float *__restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long n = idx / plane, pix = idx - n * plane;
    const float *b = x + n * C * plane + pix;
    reinterpret_cast<float4 *>(out)[idx] = make_float4(b[0], C > 1 ? b[plane] : 0.f, C > 2 ? b[2 * plane] : 0.f, C > 3 ? b[3 * plane] : 0.f);
}

extern "C" int upsnet_image_to_nhwc4(void *stream, const float *nchw, int batch, int channels, int height, int width, float *nhwc4)
{
    UPS_REQUIRE(nchw && nhwc4 && batch > 0 && channels >= 1 && channels <= 4 && height > 0 && width > 0, "image_to_nhwc4: bad arguments");
    const long plane = (long)height * width, total = plane * batch;
    hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nchw, channels,
                       plane, total, nhwc4);
    UPS_CHECK_LAUNCH("image_to_nhwc4_kernel");
    return 0;
}
