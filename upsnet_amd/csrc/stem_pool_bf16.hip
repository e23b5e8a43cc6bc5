// stem_pool_bf16.hip -- the ResNet stem as ONE launch on the bf16 matrix cores: 7x7 / stride 2 / pad 3 convolution (3 -> 64, frozen
// BN folded) + ReLU + 3x3 / stride 2 / pad 1 max-pool (gfx950). Reference: resnet.conv1 (upsnet/models/resnet.py:347-356) in the bf16
// mode of BASELINE.json configs[2].
//
// As two launches (fp32 MFMA stem 153 us + library max-pool 53 us at 1024x2048) the layer writes the 134 MB convolution output to HBM
// and reads it back; its arithmetic is 13 GFLOP = 5 us of bf16 matrix-pipe time. Here a workgroup owns 4 x 16 POOLED pixels:
//   * the 23 x 72 input pixels under them (NHWC4 fp32 image: RGB + a zero channel) are loaded once, rounded to bf16 and kept in LDS
//     (8 bytes per pixel, 13 KB);
//   * the 9 x 33 convolution outputs the pool windows touch are computed as 10 blocks of 32 pixels, MFMA operands swapped
//     (A = weights, rows = 32 output channels; B = activations, columns = pixels): K = 7 rows x 8 columns x 4 channels = 14 k-steps
//     of 16 (the eighth column and the fourth channel carry zero weights), a lane's B fragment = 2 neighbouring input pixels x 4
//     channels = ONE 16-byte LDS read, the 28 weight fragments of a wave stay in registers for the whole kernel;
//   * bias + ReLU, rounded to bf16 into an LDS tile [pixel][64 channels]; convolution outputs outside the map are stored as 0 (the
//     pool pads with -inf and every window holds a real, non-negative value, so 0 never changes a maximum);
//   * the pool reads nine 16-byte pieces per (pooled pixel, 8 channels) and writes the bf16 NHWC result: 16 bytes per lane, whole
//     128-byte lines per pooled pixel.
#include <stdlib.h>

#include "common.h"
#include "upsnet_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 sp_bf16x8;
typedef float sp_floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned sp_uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned sp_uintx2 __attribute__((ext_vector_type(2)));

#define SP_TPH 4                          // pooled rows / columns per workgroup
#define SP_TPW 16
#define SP_CTH (2 * SP_TPH + 1)           // convolution outputs under them: 9 x 33
#define SP_CTW (2 * SP_TPW + 1)
#define SP_NCT (SP_CTH * SP_CTW)          // 297
#define SP_NBLK ((SP_NCT + 31) / 32)      // 10 blocks of 32 pixels
#define SP_PH (2 * (SP_CTH - 1) + 7)      // input patch: 23 x 72 pixels (one column beyond the 7 taps: the zero eighth tap)
#define SP_PW (2 * (SP_CTW - 1) + 8)
#define SP_CTP 144                        // bytes per convolution pixel in LDS: 64 bf16 + 16

__device__ static inline unsigned sp_pack2(const float a, const float b)
{
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    unsigned short ux, uy;
    __builtin_memcpy(&ux, &x, 2);
    __builtin_memcpy(&uy, &y, 2);
    return (unsigned)ux | ((unsigned)uy << 16);
}

// weight [64, Cin <= 4, 7, 7] fp32 -> bf16 fragments [cb 2][k-step 14][lane 64][8]: lane (row l, half h) of k-step s = (ky, kx0 = 4 (s & 1))
// holds W[32 cb + l][c][ky][kx0 + 2 h + e / 4] for e = 0..7, c = e % 4; zero for kx = 7 and c >= Cin
__global__ void stem_pool_pack_weight_kernel(const float *__restrict__ w, int cin, __bf16 *__restrict__ wp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * 14 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) % 14, cb = idx / (14 * 512);
    const int co = 32 * cb + (lane & 31), ky = s >> 1, kx = 4 * (s & 1) + 2 * (lane >> 5) + (e >> 2), c = e & 3;
    wp[idx] = (__bf16)((kx < 7 && c < cin) ? w[((co * cin + c) * 7 + ky) * 7 + kx] : 0.f);
}

extern "C" int upsnet_stem_pool_pack_weight_bf16(void *stream, const float *weight, int cin, void *wpack)
{
    UPS_REQUIRE(weight && wpack && cin >= 1 && cin <= 4, "stem_pool_pack_weight_bf16: weight [64, Cin <= 4, 7, 7]");
    hipLaunchKernelGGL(stem_pool_pack_weight_kernel, dim3((2 * 14 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, cin,
                       reinterpret_cast<__bf16 *>(wpack));
    UPS_CHECK_LAUNCH("stem_pool_pack_weight_kernel");
    return 0;
}

__global__ void __launch_bounds__(256, 2)
stem_pool_bf16_kernel(const float *__restrict__ x4, const int N, const int H, const int W, const char *__restrict__ wpk, const float *__restrict__ bias,
                      const int Hc, const int Wc, const int Hp, const int Wp, const int tiles_x, const int tiles_y, unsigned short *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char PT[SP_PH * SP_PW * 8];
    __shared__ __attribute__((aligned(16))) unsigned char CT[SP_NBLK * 32 * SP_CTP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, lhalf = lane >> 5;
    const int t = blockIdx.x, per = tiles_x * tiles_y;
    const int t_n = t / per, t_r = t - t_n * per;
    const int t_y = t_r / tiles_x, t_x = t_r - t_y * tiles_x;
    const int py0 = t_y * SP_TPH, px0 = t_x * SP_TPW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;          // convolution pixel of tile position (0, 0)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // input pixel of patch position (0, 0)

    // ---- the weights of this wave's MFMAs: 2 channel blocks x 14 k-steps, resident (112 registers)
    sp_bf16x8 wf[2][14];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 14; ++s) wf[cb][s] = *reinterpret_cast<const sp_bf16x8 *>(wpk + ((cb * 14 + s) * 64 + lane) * 16);

    // ---- input patch: fp32 NHWC4 -> bf16, zero outside the image
    const float4 *xin = reinterpret_cast<const float4 *>(x4) + (size_t)t_n * H * W;
    for (int i = tid; i < SP_PH * SP_PW; i += 256) {
        const int r = i / SP_PW, c = i - r * SP_PW;
        const int iy = iy0 + r, ix = ix0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xin[(size_t)iy * W + ix];
        sp_uintx2 h;
        h.x = sp_pack2(v.x, v.y); h.y = sp_pack2(v.z, v.w);
        *reinterpret_cast<sp_uintx2 *>(PT + i * 8) = h;
    }
    __syncthreads();

    // ---- convolution: blocks of 32 tile pixels, wave w takes blocks w, w + 4, w + 8
    float4 bv[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[cb][g] = bias ? *reinterpret_cast<const float4 *>(bias + cb * 32 + 8 * g + 4 * lhalf) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int blk = wave; blk < SP_NBLK; blk += 4) {
        const int q = blk * 32 + l32;
        const int qq = q < SP_NCT ? q : 0;
        const int cyl = qq / SP_CTW, cxl = qq - cyl * SP_CTW;
        const unsigned char *base = PT + ((2 * cyl) * SP_PW + 2 * cxl) * 8 + lhalf * 16;
        sp_floatx16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const sp_bf16x8 xf = *reinterpret_cast<const sp_bf16x8 *>(base + ((s >> 1) * SP_PW + 4 * (s & 1)) * 8);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][s], xf, acc[cb], 0, 0, 0);
        }
        const int cy = cy0 + cyl, cx = cx0 + cxl;
        const bool real = q < SP_NCT && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = bv[cb][g];
                const float v0 = real ? fmaxf(acc[cb][4 * g + 0] + b.x, 0.f) : 0.f, v1 = real ? fmaxf(acc[cb][4 * g + 1] + b.y, 0.f) : 0.f;
                const float v2 = real ? fmaxf(acc[cb][4 * g + 2] + b.z, 0.f) : 0.f, v3 = real ? fmaxf(acc[cb][4 * g + 3] + b.w, 0.f) : 0.f;
                sp_uintx2 pk;
                pk.x = sp_pack2(v0, v1); pk.y = sp_pack2(v2, v3);
                *reinterpret_cast<sp_uintx2 *>(CT + q * SP_CTP + (cb * 32 + 8 * g + 4 * lhalf) * 2) = pk;
            }
    }
    __syncthreads();

    // ---- 3x3 / 2 max-pool over the tile: item = (pooled pixel, 8 channels); values are bf16 >= 0
    for (int item = tid; item < SP_TPH * SP_TPW * 8; item += 256) {
        const int pp = item >> 3, c8 = item & 7;
        const int ppy = pp / SP_TPW, ppx = pp - ppy * SP_TPW;
        const int py = py0 + ppy, px = px0 + ppx;
        float m[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const sp_uintx4 v = *reinterpret_cast<const sp_uintx4 *>(CT + ((2 * ppy + dy) * SP_CTW + 2 * ppx + dx) * SP_CTP + c8 * 16);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    m[2 * c] = fmaxf(m[2 * c], __uint_as_float(w4[c] << 16));
                    m[2 * c + 1] = fmaxf(m[2 * c + 1], __uint_as_float(w4[c] & 0xffff0000u));
                }
            }
        if (py < Hp && px < Wp) {
            sp_uintx4 o;          // (the maxima are bf16 values: the upper halves are exact)
            o.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u);
            o.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u);
            o.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xffff0000u);
            o.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xffff0000u);
            *reinterpret_cast<sp_uintx4 *>(out + (((size_t)t_n * Hp + py) * Wp + px) * 64 + c8 * 8) = o;
        }
    }
}

/* conv 7x7 / 2 / 3 (Cin <= 4 -> 64, + bias) + ReLU + max-pool 3x3 / 2 / 1 in one launch on the bf16 matrix cores (bf16 mode of the
 * backbone stem, upsnet/models/resnet.py:347-356). x4: [N,H,W,4] fp32 (RGB + zero channel, upsnet_image_to_nhwc4 / upsnet_prep_image_u8);
 * wpack: upsnet_stem_pool_pack_weight_bf16; bias [64] or NULL; out [N,Hp,Wp,64] bf16 with Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1
 * (likewise for the width). The convolution result is rounded to bf16 before the pool. */
extern "C" int upsnet_stem_pool_bf16(void *stream, const float *x4, int batch, int height, int width, const void *wpack, const float *bias,
                                     void *out)
{
    UPS_REQUIRE(x4 && wpack && out && batch > 0 && height > 0 && width > 0, "stem_pool_bf16: bad arguments");
    UPS_REQUIRE((long)batch * height * width < (1L << 28), "stem_pool_bf16: image batch too large");
    const int Hc = (height - 1) / 2 + 1, Wc = (width - 1) / 2 + 1;
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    const int tiles_x = (Wp + SP_TPW - 1) / SP_TPW, tiles_y = (Hp + SP_TPH - 1) / SP_TPH;
    hipLaunchKernelGGL(stem_pool_bf16_kernel, dim3((unsigned)(batch * tiles_x * tiles_y)), dim3(256), 0, (hipStream_t)stream, x4, batch, height,
                       width, reinterpret_cast<const char *>(wpack), bias, Hc, Wc, Hp, Wp, tiles_x, tiles_y, reinterpret_cast<unsigned short *>(out));
    UPS_CHECK_LAUNCH("stem_pool_bf16_kernel");
    return 0;
}
