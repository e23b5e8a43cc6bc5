// stem_pool.hip -- the ResNet stem as ONE launch on the fp32 matrix cores: 7x7 / stride 2 / pad 3 convolution (3 -> 64, frozen BN
// folded) + ReLU + 3x3 / stride 2 / pad 1 max-pool (gfx950). Reference: resnet.conv1 (upsnet/models/resnet.py:347-356); the headline
// (fp32) form of stem_pool_bf16.hip.
//
// As two launches (fp32 MFMA stem 148 us + library max-pool 58 us at 1024x2048, profiles/r09) the layer writes its 134 MB convolution
// output to HBM and reads it back (fused: 163 us, tools/microbench_stem.py). Here a workgroup owns 4 x 16 POOLED pixels:
//   * the 23 x 72 input pixels under them (NHWC4 fp32: RGB + a zero channel, 16 bytes per pixel) are loaded once into LDS, even and
//     odd columns in separate half rows (the stride-2 convolution reads every other column: 16 consecutive lanes then read 16
//     consecutive 16-byte slots);
//   * the 9 x 33 convolution outputs the pool windows touch are computed as 10 blocks of 32 pixels with v_mfma_f32_32x32x2_f32, operands
//     swapped (A = weights, rows = 32 output channels; B = activations, columns = pixels): a k-pair = two CONSECUTIVE TAPS (2t, 2t + 1) of
//     the 49 in raster order (the 50th is a zero weight) for ONE input channel -- lane half 0 reads the pixel of tap 2t, half 1 the pixel
//     of tap 2t + 1 --, so a lane's ONE 16-byte LDS read (its pixel, 4 channels) feeds three MFMAs (the fourth channel is zero and
//     skipped): K = 25 tap pairs x 3 channels = 75 MFMAs per block and 32-channel half. The 2 x 75 weight values of a lane stay in
//     registers for the whole kernel;
//   * per 32-channel half: bias + ReLU into an LDS tile [pixel][32 channels] fp32 (outside the map: 0 -- the pool pads with -inf and every
//     window holds a real, non-negative value, so 0 never changes a maximum), then the pool: nine 16-byte reads per (pooled pixel,
//     4 channels), one 16-byte store.
// Exact fp32 products, fp32 sums in a fixed order (tap pair in raster order, channel).
#include <stdlib.h>

#include "common.h"
#include "upsnet_hip.h"

typedef float spf_floatx16 __attribute__((ext_vector_type(16)));

#define SPF_TPW 16                          // pooled columns per workgroup (pooled rows: template parameter TPH)
#define SPF_CTW (2 * SPF_TPW + 1)           // convolution outputs under them: (2 TPH + 1) x 33
#define SPF_PW (2 * (SPF_CTW - 1) + 8)      // input patch: (4 TPH + 7) rows x 72 columns (one column beyond the 7 taps: the zero eighth tap)
#define SPF_PWH (SPF_PW / 2)                // columns per parity
#define SPF_CTP 144                         // bytes per convolution pixel in LDS: 32 fp32 + 16
#define SPF_NW 75                           // weight values per lane and 32-channel half: 25 tap pairs x 3 channels

// weight [64, Cin <= 3, 7, 7] fp32 -> [half 2][tap pair 25][c 3][lane 64]: lane (row l, k = lane / 32) holds W[32 half + l][c][tap 2 t + k]
// with tap = ky 7 + kx (0 for tap 49 and c >= Cin)
__global__ void stem_pool_pack_weight_f32_kernel(const float *__restrict__ w, int cin, float *__restrict__ wp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * SPF_NW * 64) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int c = r % 3; r /= 3;
    const int t = r % 25, cb = r / 25;
    const int co = 32 * cb + (lane & 31), tap = 2 * t + (lane >> 5);
    wp[idx] = (tap < 49 && c < cin) ? w[(co * cin + c) * 49 + tap] : 0.f;
}

extern "C" int upsnet_stem_pool_pack_weight_f32(void *stream, const float *weight, int cin, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cin >= 1 && cin <= 3, "stem_pool_pack_weight_f32: weight [64, Cin <= 3, 7, 7]");
    hipLaunchKernelGGL(stem_pool_pack_weight_f32_kernel, dim3((2 * SPF_NW * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, cin, wpack);
    UPS_CHECK_LAUNCH("stem_pool_pack_weight_f32_kernel");
    return 0;
}

template <int SPF_TPH>
__global__ void __launch_bounds__(256, SPF_TPH <= 4 ? 2 : 1)
stem_pool_f32_kernel(const float *__restrict__ x4, const int N, const int H, const int W, const float *__restrict__ wpk, const float *__restrict__ bias,
                     const int Hc, const int Wc, const int Hp, const int Wp, const int tiles_x, const int tiles_y, float *__restrict__ out)
{
    constexpr int SPF_CTH = 2 * SPF_TPH + 1, SPF_NCT = SPF_CTH * SPF_CTW, SPF_NBLK = (SPF_NCT + 31) / 32, SPF_PH = 2 * (SPF_CTH - 1) + 7;
    __shared__ __attribute__((aligned(16))) unsigned char PT[SPF_PH * SPF_PW * 16];
    __shared__ __attribute__((aligned(16))) unsigned char CT[SPF_NBLK * 32 * SPF_CTP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, lhalf = lane >> 5;
    const int t = blockIdx.x, per = tiles_x * tiles_y;
    const int t_n = t / per, t_r = t - t_n * per;
    const int t_y = t_r / tiles_x, t_x = t_r - t_y * tiles_x;
    const int py0 = t_y * SPF_TPH, px0 = t_x * SPF_TPW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;          // convolution pixel of tile position (0, 0)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // input pixel of patch position (0, 0)

    // ---- the weights: 2 x 75 values per lane, resident for the whole kernel (issued first: in flight while the patch is staged)
    float wf[2][SPF_NW];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int k = 0; k < SPF_NW; ++k) wf[cb][k] = wpk[(cb * SPF_NW + k) * 64 + lane];
    // ---- input patch, zero outside the image; column c of row r at [(r, c & 1)][c >> 1]
    const float4 *xin = reinterpret_cast<const float4 *>(x4) + (size_t)t_n * H * W;
    for (int i = tid; i < SPF_PH * SPF_PW; i += 256) {
        const int r = i / SPF_PW, c = i - r * SPF_PW;
        const int iy = iy0 + r, ix = ix0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xin[(size_t)iy * W + ix];
        *reinterpret_cast<float4 *>(PT + ((r * 2 + (c & 1)) * SPF_PWH + (c >> 1)) * 16) = v;
    }
    __syncthreads();

#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = bias ? *reinterpret_cast<const float4 *>(bias + cb * 32 + 8 * g + 4 * lhalf) : make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- convolution: blocks of 32 tile pixels, wave w takes blocks w, w + 4, w + 8
        for (int blk = wave; blk < SPF_NBLK; blk += 4) {
            const int q = blk * 32 + l32;
            const int qq = q < SPF_NCT ? q : 0;
            const int cyl = qq / SPF_CTW, cxl = qq - cyl * SPF_CTW;
            // tap (ky, kx) of this pixel: patch row 2 cyl + ky, column 2 cxl + kx = parity kx & 1, slot cxl + (kx >> 1)
            const unsigned char *base = PT + ((2 * cyl) * 2 * SPF_PWH + cxl) * 16;
            spf_floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                // lane half 0: tap 2 t, half 1: tap 2 t + 1 (tap 49 has zero weights: any valid address)
                const int ta = 2 * t, tb = 2 * t + 1 < 49 ? 2 * t + 1 : 48;
                const int offa = (((ta / 7) * 2 + ((ta % 7) & 1)) * SPF_PWH + ((ta % 7) >> 1)) * 16;
                const int offb = (((tb / 7) * 2 + ((tb % 7) & 1)) * SPF_PWH + ((tb % 7) >> 1)) * 16;
                const float4 xv = *reinterpret_cast<const float4 *>(base + (lhalf ? offb : offa));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][t * 3 + 0], xv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][t * 3 + 1], xv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[cb][t * 3 + 2], xv.z, acc, 0, 0, 0);
            }
            const int cy = cy0 + cyl, cx = cx0 + cxl;
            const bool real = q < SPF_NCT && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = real ? fmaxf(acc[4 * g + 0] + bv[g].x, 0.f) : 0.f; v.y = real ? fmaxf(acc[4 * g + 1] + bv[g].y, 0.f) : 0.f;
                v.z = real ? fmaxf(acc[4 * g + 2] + bv[g].z, 0.f) : 0.f; v.w = real ? fmaxf(acc[4 * g + 3] + bv[g].w, 0.f) : 0.f;
                *reinterpret_cast<float4 *>(CT + q * SPF_CTP + (8 * g + 4 * lhalf) * 4) = v;
            }
        }
        __syncthreads();
        // ---- 3x3 / 2 max-pool of this half: item = (pooled pixel, 4 channels)
        for (int item = tid; item < SPF_TPH * SPF_TPW * 8; item += 256) {
            const int pp = item >> 3, c4 = item & 7;
            const int ppy = pp / SPF_TPW, ppx = pp - ppy * SPF_TPW;
            const int py = py0 + ppy, px = px0 + ppx;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float4 v = *reinterpret_cast<const float4 *>(CT + ((2 * ppy + dy) * SPF_CTW + 2 * ppx + dx) * SPF_CTP + c4 * 16);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            if (py < Hp && px < Wp) *reinterpret_cast<float4 *>(out + (((size_t)t_n * Hp + py) * Wp + px) * 64 + cb * 32 + c4 * 4) = m;
        }
        __syncthreads();      // the tile is rewritten by the next half
    }
}

/* conv 7x7 / 2 / 3 (Cin <= 3 -> 64, + bias) + ReLU + max-pool 3x3 / 2 / 1 in one launch on the fp32 matrix cores (backbone stem,
 * upsnet/models/resnet.py:347-356). x4: [N,H,W,4] fp32 (RGB + zero channel, upsnet_image_to_nhwc4 / upsnet_prep_image_u8); wpack:
 * upsnet_stem_pool_pack_weight_f32 (2 x 75 x 64 floats); bias [64] or NULL; out [N,Hp,Wp,64] fp32 with Hc = (H - 1) / 2 + 1,
 * Hp = (Hc - 1) / 2 + 1 (likewise for the width). */
extern "C" int upsnet_stem_pool_f32(void *stream, const float *x4, int batch, int height, int width, const float *wpack, const float *bias,
                                    float *out)
{
    UPS_REQUIRE(x4 && wpack && out && batch > 0 && height > 0 && width > 0, "stem_pool_f32: bad arguments");
    UPS_REQUIRE((long)batch * height * width < (1L << 28), "stem_pool_f32: image batch too large");
    const int Hc = (height - 1) / 2 + 1, Wc = (width - 1) / 2 + 1;
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    // (pooled tile rows 3 / 4 / 7 = 8 / 10 / 16 pixel blocks per workgroup measured 164 / 163 / 159 us at 1024x2048 and 88 / 97 / 94 us at
    // 800x1344, tools/microbench_stem.py: the kernel is not bound by the balance of its four waves; 4 keeps two workgroups per CU)
    constexpr int tph = 4;
    const int tiles_x = (Wp + SPF_TPW - 1) / SPF_TPW, tiles_y = (Hp + tph - 1) / tph;
    hipLaunchKernelGGL(stem_pool_f32_kernel<tph>, dim3((unsigned)(batch * tiles_x * tiles_y)), dim3(256), 0, (hipStream_t)stream, x4, batch, height,
                       width, wpack, bias, Hc, Wc, Hp, Wp, tiles_x, tiles_y, out);
    UPS_CHECK_LAUNCH("stem_pool_f32_kernel");
    return 0;
}
