// fcn_head.hip -- tail of the semantic head (upsnet/models/fcn.py:94-100).
//
// Reference: P3/P4/P5 subnet outputs (128 ch each) are bilinearly upsampled x2/x4/x8 to P2 size, concatenated with P2's
// (512 ch, 268 MB at 256x512) and pushed through the 1x1 `score` convolution.
// Here: the 1x1 convolution and the bilinear upsampling are both linear and the interpolation weights sum to one, so
//     score = b + sum_l up_{2^l}( W[:, 128 l : 128 (l+1)] . y_l )
// The four small products run at each level's own resolution on the MFMA convolution kernel (csrc/conv.hip); this
// kernel adds their upsampled values: 13 MB read + 10 MB written instead of three upsample passes, the concat and a
// 512-channel convolution. Same value up to fp32 summation order (the reference's own sgemm order is unpinned,
// SURVEY.md 8c-iv); the interpolation follows upsample_bilinear2d (align_corners=False) term by term.
#include "common.h"
#include "upsnet_hip.h"

#define FSC_MAXLEV 4

struct FscParams {
    const float *part[FSC_MAXLEV];
    const float *bias;
    float *out;
    int nlev, S, H, W;
};

__global__ void __launch_bounds__(256)
fcn_score_combine_kernel(const FscParams p)
{
    const long total = (long)p.H * p.W * p.S;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int s = (int)(idx % p.S);
    const long pix = idx / p.S;
    const int x = (int)(pix % p.W), y = (int)(pix / p.W);
    float acc = p.part[0][idx];
    if (p.bias) acc = acc + p.bias[s];
#pragma unroll
    for (int l = 1; l < FSC_MAXLEV; ++l) {
        if (l >= p.nlev) break;
        const int Hs = p.H >> l, Ws = p.W >> l;
        const float r = 1.0f / (float)(1 << l);
        float h1r = r * ((float)y + 0.5f) - 0.5f;
        if (h1r < 0) h1r = 0;
        const int h1 = (int)h1r, h1p = h1 < Hs - 1 ? 1 : 0;
        const float h1l = h1r - (float)h1, h0l = 1.0f - h1l;
        float w1r = r * ((float)x + 0.5f) - 0.5f;
        if (w1r < 0) w1r = 0;
        const int w1 = (int)w1r, w1p = w1 < Ws - 1 ? 1 : 0;
        const float w1l = w1r - (float)w1, w0l = 1.0f - w1l;
        const float *q = p.part[l] + s;
        const long S = p.S;
        const float top = w0l * q[((long)h1 * Ws + w1) * S] + w1l * q[((long)h1 * Ws + w1 + w1p) * S];
        const float bot = w0l * q[((long)(h1 + h1p) * Ws + w1) * S] + w1l * q[((long)(h1 + h1p) * Ws + w1 + w1p) * S];
        acc = acc + (h0l * top + h1l * bot);
    }
    p.out[idx] = acc;
}

extern "C" int upsnet_fcn_score_combine(void *stream, int nlev, const float *const part[], int num_seg, int height, int width,
                                        const float *bias, float *score)
{
    UPS_REQUIRE(nlev >= 1 && nlev <= FSC_MAXLEV && part && score, "fcn_score_combine: 1..%d levels (got %d)", FSC_MAXLEV, nlev);
    UPS_REQUIRE(num_seg > 0 && height > 0 && width > 0, "fcn_score_combine: bad shape");
    UPS_REQUIRE(height % (1 << (nlev - 1)) == 0 && width % (1 << (nlev - 1)) == 0,
                "fcn_score_combine: %dx%d is not divisible by 2^%d", height, width, nlev - 1);
    FscParams p;
    for (int l = 0; l < FSC_MAXLEV; ++l) {
        p.part[l] = l < nlev ? part[l] : nullptr;
        UPS_REQUIRE(l >= nlev || part[l], "fcn_score_combine: null level %d", l);
    }
    p.bias = bias; p.out = score; p.nlev = nlev; p.S = num_seg; p.H = height; p.W = width;
    const long total = (long)height * width * num_seg;
    hipLaunchKernelGGL(fcn_score_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    UPS_CHECK_LAUNCH("fcn_score_combine_kernel");
    return 0;
}
