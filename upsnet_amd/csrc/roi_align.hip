// roi_align.hip -- ROIAlign forward for gfx950.
//
//  * upsnet_roi_align_forward      : NCHW drop-in for the reference launcher
//                                    (upsnet/operators/src/roi_align_cuda.cpp:26-30, kernel
//                                    roi_align_kernel.cu:163-235).
//  * upsnet_fpn_roi_align_forward  : the MI355X-native path. One launch for all four FPN levels,
//                                    FPN level chosen on device (fpn_roi_align.py:36-38), features
//                                    NHWC so each bilinear tap of a (roi, bin) is ONE contiguous
//                                    C-vector: a wave reads a tap as 64 x float4 = 1 KiB, fully
//                                    coalesced, output NHWC [N, PH, PW, C] written the same way.
//
// Arithmetic is the reference's, per channel, in fp32 without FMA contraction: every output value is
// bit-identical to the CPU oracle.
#include <stdlib.h>

#include "common.h"
#include "roi_order.h"
#include "upsnet_hip.h"

// ---------------------------------------------------------------------------------------------
// shared: per-sample bilinear setup (roi_align_kernel.cu:43-95), returns false for "empty" samples
struct RoiTap {
    int y_low, y_high, x_low, x_high;
    float w1, w2, w3, w4;
};

__device__ static inline bool roi_tap(int height, int width, float y, float x, RoiTap &t)
{
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return false;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.0f - ly, hx = 1.0f - lx;
    t.y_low = y_low; t.y_high = y_high; t.x_low = x_low; t.x_high = x_high;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    return true;
}

__device__ static inline float roi_blend(const RoiTap &t, float v1, float v2, float v3, float v4)
{
    float val = t.w1 * v1;
    val = val + t.w2 * v2;
    val = val + t.w3 * v3;
    val = val + t.w4 * v4;
    return val;
}

// ---------------------------------------------------------------------------------------------
// NHWC bin evaluation for sampling_ratio == 2 (the only value the reference uses, functions/roialign.py:22): the four
// sample descriptors are set up first and all 16 corner vectors of a channel group are loaded UNCONDITIONALLY (clamped
// addresses; samples outside the map are deselected afterwards), so the 16 loads of a wave are in flight together instead
// of four dependent round trips. Summation order and arithmetic are the reference's (roi_align_kernel.cu:199-231).
__device__ static inline void roi_bin_nhwc_g2(const float4 *__restrict__ feat, const int c4n, const int height, const int width,
                                              const float roi_start_h, const float roi_start_w, const float bin_size_h,
                                              const float bin_size_w, const int ph, const int pw, const int lane,
                                              float4 *__restrict__ o4)
{
    RoiTap t0, t1, t2, t3;
    const float y0 = roi_start_h + (float)ph * bin_size_h + (0.f + .5f) * bin_size_h / 2.0f;
    const float y1 = roi_start_h + (float)ph * bin_size_h + (1.f + .5f) * bin_size_h / 2.0f;
    const float x0 = roi_start_w + (float)pw * bin_size_w + (0.f + .5f) * bin_size_w / 2.0f;
    const float x1 = roi_start_w + (float)pw * bin_size_w + (1.f + .5f) * bin_size_w / 2.0f;
    t0.y_low = t0.y_high = t0.x_low = t0.x_high = 0; t1 = t0; t2 = t0; t3 = t0;
    const bool k0 = roi_tap(height, width, y0, x0, t0), k1 = roi_tap(height, width, y0, x1, t1);
    const bool k2 = roi_tap(height, width, y1, x0, t2), k3 = roi_tap(height, width, y1, x1, t3);
#define RB_OFF(T, YY, XX) ((unsigned)((T.YY * width + T.XX) * c4n))
    const unsigned a0 = RB_OFF(t0, y_low, x_low), a1 = RB_OFF(t0, y_low, x_high), a2 = RB_OFF(t0, y_high, x_low), a3 = RB_OFF(t0, y_high, x_high);
    const unsigned b0 = RB_OFF(t1, y_low, x_low), b1 = RB_OFF(t1, y_low, x_high), b2 = RB_OFF(t1, y_high, x_low), b3 = RB_OFF(t1, y_high, x_high);
    const unsigned c0 = RB_OFF(t2, y_low, x_low), c1 = RB_OFF(t2, y_low, x_high), c2 = RB_OFF(t2, y_high, x_low), c3 = RB_OFF(t2, y_high, x_high);
    const unsigned d0 = RB_OFF(t3, y_low, x_low), d1 = RB_OFF(t3, y_low, x_high), d2 = RB_OFF(t3, y_high, x_low), d3 = RB_OFF(t3, y_high, x_high);
#undef RB_OFF
    for (int c4 = lane; c4 < c4n; c4 += 64) {
        const float4 *f = feat + c4;
        const float4 va0 = f[a0], va1 = f[a1], va2 = f[a2], va3 = f[a3];
        const float4 vb0 = f[b0], vb1 = f[b1], vb2 = f[b2], vb3 = f[b3];
        const float4 vc0 = f[c0], vc1 = f[c1], vc2 = f[c2], vc3 = f[c3];
        const float4 vd0 = f[d0], vd1 = f[d1], vd2 = f[d2], vd3 = f[d3];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define RB_ACC(K, T, V0, V1, V2, V3)                                      \
        acc.x += K ? roi_blend(T, V0.x, V1.x, V2.x, V3.x) : 0.f;         \
        acc.y += K ? roi_blend(T, V0.y, V1.y, V2.y, V3.y) : 0.f;         \
        acc.z += K ? roi_blend(T, V0.z, V1.z, V2.z, V3.z) : 0.f;         \
        acc.w += K ? roi_blend(T, V0.w, V1.w, V2.w, V3.w) : 0.f;
        RB_ACC(k0, t0, va0, va1, va2, va3)
        RB_ACC(k1, t1, vb0, vb1, vb2, vb3)
        RB_ACC(k2, t2, vc0, vc1, vc2, vc3)
        RB_ACC(k3, t3, vd0, vd1, vd2, vd3)
#undef RB_ACC
        acc.x /= 4.0f; acc.y /= 4.0f; acc.z /= 4.0f; acc.w /= 4.0f;
        o4[c4] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// NCHW drop-in (upsnet_roi_align_forward, the reference launcher's signature and layouts). What the contract fixes is the
// arithmetic of one output element (roi_align_kernel.cu:43-95,199-231); how the work is organised is this file's own: one
// workgroup per ROI (x a slice of its channels). The ROI is decoded ONCE and its sample grid -- PH*grid_h sample rows, PW*grid_w
// sample columns, any sampling_ratio incl. the adaptive ceil(roi / pooled) -- is tabulated per AXIS in LDS (low / high index,
// the two weights, 0 / 0 for a sample outside the map); a thread then owns (channel, bin) outputs of that ROI and only reads the
// table and the channel plane: the reference recomputes the whole ROI decode + bilinear setup for every output element.
struct RoiAxisN { int lo, hi; float l, h; };
__device__ static inline RoiAxisN roi_axis_n(const int size, float y)
{
    const bool empty = y < -1.0f || y > (float)size;
    if (y <= 0) y = 0;
    int lo = (int)y, hi;
    if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else { hi = lo + 1; }
    const float l = y - (float)lo;
    RoiAxisN a;
    a.lo = lo; a.hi = hi; a.l = empty ? 0.f : l; a.h = empty ? 0.f : 1.0f - l;
    return a;
}

#define ROI_NCHW_MAXAXIS 4096   // (sample rows + columns) the LDS table holds at most: 64 KiB

__global__ void __launch_bounds__(256)
roi_align_nchw_kernel(const float *__restrict__ feat, const float spatial_scale, const int channels, const int height, const int width,
                      const int pooled_h, const int pooled_w, const int sampling_ratio, const float *__restrict__ rois,
                      float *__restrict__ out, const int cap_y, const int cap_x)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = blockIdx.x;
    const float *r = rois + (long)n * 5;
    const int roi_batch_ind = (int)roundf(r[0]);
    const float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
    const float roi_end_w = r[3] * spatial_scale, roi_end_h = r[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
    const long ny = (long)pooled_h * grid_h, nx = (long)pooled_w * grid_w;
    // (an adaptive grid of a box far larger than the map can exceed the table: such a ROI evaluates its axis entries on the fly)
    const bool tabulated = ny <= cap_y && nx <= cap_x;
    RoiAxisN *ytab = reinterpret_cast<RoiAxisN *>(smem_raw), *xtab = ytab + cap_y;
#define ROI_N_YENT(PH_, IY_) roi_axis_n(height, roi_start_h + (float)(PH_) * bin_size_h + ((float)(IY_) + .5f) * bin_size_h / (float)grid_h)
#define ROI_N_XENT(PW_, IX_) roi_axis_n(width, roi_start_w + (float)(PW_) * bin_size_w + ((float)(IX_) + .5f) * bin_size_w / (float)grid_w)
    if (tabulated) {
        for (int s = threadIdx.x; s < (int)ny; s += blockDim.x) { const int ph = s / grid_h; ytab[s] = ROI_N_YENT(ph, s - ph * grid_h); }
        for (int s = threadIdx.x; s < (int)nx; s += blockDim.x) { const int pw = s / grid_w; xtab[s] = ROI_N_XENT(pw, s - pw * grid_w); }
    }
    __syncthreads();
    const float count = (float)(grid_h * grid_w);
    const int bins = pooled_h * pooled_w;
    const long per_roi = (long)channels * bins;
    const float *img = feat + (long)roi_batch_ind * channels * height * width;
    for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < per_roi; e += (long)gridDim.y * blockDim.x) {
        const int c = (int)(e / bins), bin = (int)(e - (long)c * bins);
        const int ph = bin / pooled_w, pw = bin - ph * pooled_w;
        const float *plane = img + (long)c * height * width;
        float acc = 0.f;
        for (int iy = 0; iy < grid_h; ++iy) {
            const RoiAxisN ya = tabulated ? ytab[ph * grid_h + iy] : ROI_N_YENT(ph, iy);
            for (int ix = 0; ix < grid_w; ++ix) {
                const RoiAxisN xa = tabulated ? xtab[pw * grid_w + ix] : ROI_N_XENT(pw, ix);
                const float w1 = ya.h * xa.h, w2 = ya.h * xa.l, w3 = ya.l * xa.h, w4 = ya.l * xa.l;
                float val = w1 * plane[ya.lo * width + xa.lo];
                val = val + w2 * plane[ya.lo * width + xa.hi];
                val = val + w3 * plane[ya.hi * width + xa.lo];
                val = val + w4 * plane[ya.hi * width + xa.hi];
                acc += val;
            }
        }
        out[(long)n * per_roi + e] = acc / count;
    }
#undef ROI_N_YENT
#undef ROI_N_XENT
}

extern "C" int upsnet_roi_align_forward(void *stream, const float *bottom_data, float spatial_scale,
                                        int num_rois, int height, int width, int channels,
                                        int pooled_height, int pooled_width, int sampling_ratio,
                                        const float *bottom_rois, float *top_data)
{
    UPS_REQUIRE(bottom_data && bottom_rois && top_data, "roi_align_forward: null pointer");
    UPS_REQUIRE(num_rois >= 0 && channels > 0 && height > 0 && width > 0 && pooled_height > 0 && pooled_width > 0,
                "roi_align_forward: bad shape");
    if (num_rois == 0) return 0;
    // table capacity: a fixed sampling ratio -> exact (capped); adaptive (sampling_ratio <= 0) -> 1024 samples per axis
    int ny = sampling_ratio > 0 ? pooled_height * sampling_ratio : 1024, nx = sampling_ratio > 0 ? pooled_width * sampling_ratio : 1024;
    if (ny > ROI_NCHW_MAXAXIS / 2) ny = ROI_NCHW_MAXAXIS / 2;
    if (nx > ROI_NCHW_MAXAXIS / 2) nx = ROI_NCHW_MAXAXIS / 2;
    const size_t smem = (size_t)(ny + nx) * sizeof(RoiAxisN);      // <= 64 KiB
    long per_roi = (long)channels * pooled_height * pooled_width;
    int ysplit = (int)((per_roi + 256 * 8 - 1) / (256 * 8));      // ~8 outputs per thread
    if (ysplit < 1) ysplit = 1;
    if (ysplit > 64) ysplit = 64;
    hipLaunchKernelGGL(roi_align_nchw_kernel, dim3(num_rois, ysplit), dim3(256), smem, (hipStream_t)stream, bottom_data, spatial_scale, channels,
                       height, width, pooled_height, pooled_width, sampling_ratio, bottom_rois, top_data, ny, nx);
    UPS_CHECK_LAUNCH("roi_align_nchw_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// FPN level (fpn_roi_align.py:36-38): floor(2 + log2(sqrt(w*h)/224 + 1e-6)) clipped to [0,3], fp32.
struct FpnFeat {
    const float *ptr[4];
    int h[4], w[4];
    float scale[4];
};

// One wave per (roi, ph, pw) bin; lanes span channels in float4 groups. NHWC in, NHWC out.
__global__ void __launch_bounds__(256)
fpn_roi_align_nhwc_kernel(const FpnFeat ft, const int channels, const float *__restrict__ rois,
                          const int num_rois, const int *__restrict__ num_rois_dev, const int pooled_h,
                          const int pooled_w, const int sampling_ratio, float *__restrict__ out,
                          int *__restrict__ levels_out)
{
    const int lane = threadIdx.x & 63;
    const long bin = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int nvalid = num_rois_dev ? min(*num_rois_dev, num_rois) : num_rois;
    if (bin >= (long)num_rois * pooled_h * pooled_w) return;
    const int pw = bin % pooled_w;
    const int ph = (bin / pooled_w) % pooled_h;
    const int n = bin / ((long)pooled_w * pooled_h);
    const int c4n = channels >> 2;
    float4 *o4 = reinterpret_cast<float4 *>(out + bin * channels);
    if (n >= nvalid) {  // padded tail of a fixed-size roi buffer: defined output (zeros)
        for (int c4 = lane; c4 < c4n; c4 += 64) o4[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float *r = rois + (long)n * 5;
    const float rx1 = r[1], ry1 = r[2], rx2 = r[3], ry2 = r[4];
    const int lvl = fpn_level_of(rx1, ry1, rx2, ry2);
    if (levels_out && ph == 0 && pw == 0 && lane == 0) levels_out[n] = lvl;
    const float spatial_scale = ft.scale[lvl];
    const int height = ft.h[lvl], width = ft.w[lvl];
    const float4 *feat = reinterpret_cast<const float4 *>(ft.ptr[lvl]);

    const float roi_start_w = rx1 * spatial_scale, roi_start_h = ry1 * spatial_scale;
    const float roi_end_w = rx2 * spatial_scale, roi_end_h = ry2 * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
    const float count = (float)(grid_h * grid_w);
    if (sampling_ratio == 2 && (long)height * width * c4n < (1L << 31)) {
        roi_bin_nhwc_g2(feat, c4n, height, width, roi_start_h, roi_start_w, bin_size_h, bin_size_w, ph, pw, lane, o4);
        return;
    }

    for (int c4 = lane; c4 < c4n; c4 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < grid_h; ++iy) {
            const float y = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                const float x = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
                RoiTap t;
                if (roi_tap(height, width, y, x, t)) {
                    const float4 v1 = feat[((long)t.y_low * width + t.x_low) * c4n + c4];
                    const float4 v2 = feat[((long)t.y_low * width + t.x_high) * c4n + c4];
                    const float4 v3 = feat[((long)t.y_high * width + t.x_low) * c4n + c4];
                    const float4 v4 = feat[((long)t.y_high * width + t.x_high) * c4n + c4];
                    acc.x += roi_blend(t, v1.x, v2.x, v3.x, v4.x);
                    acc.y += roi_blend(t, v1.y, v2.y, v3.y, v4.y);
                    acc.z += roi_blend(t, v1.z, v2.z, v3.z, v4.z);
                    acc.w += roi_blend(t, v1.w, v2.w, v3.w, v4.w);
                } else {
                    acc.x += 0.f; acc.y += 0.f; acc.z += 0.f; acc.w += 0.f;
                }
            }
        }
        acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        o4[c4] = acc;
    }
}

static int roi_per_bin()  // 1: one wave per (roi, bin) (the older decomposition); A/B knob, env UPSNET_ROI_PER_BIN
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("UPSNET_ROI_PER_BIN"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

// One workgroup (4 waves) per ROI, sampling_ratio == 2: the ROI is decoded and its FPN level chosen once per wave instead of
// once per bin; each wave walks bins wave, wave+4, ... with the 16 corner loads of the NEXT bin issued before the current one
// is blended (two register sets), so a wave keeps 16-32 KiB of loads in flight. NHWC in, NHWC out, C <= 256 per pass.
struct RoiBinG2 {
    unsigned o[16];
    float w[16];
};

__device__ static inline void roi_bin_setup_g2(const int c4n, const int height, const int width, const float roi_start_h,
                                               const float roi_start_w, const float bin_size_h, const float bin_size_w, const int ph,
                                               const int pw, RoiBinG2 &b)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int iy = s >> 1, ix = s & 1;
        const float y = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / 2.0f;
        const float x = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / 2.0f;
        RoiTap t;
        t.y_low = t.y_high = t.x_low = t.x_high = 0;
        t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
        const bool ok = roi_tap(height, width, y, x, t);
        b.o[4 * s + 0] = (unsigned)((t.y_low * width + t.x_low) * c4n);
        b.o[4 * s + 1] = (unsigned)((t.y_low * width + t.x_high) * c4n);
        b.o[4 * s + 2] = (unsigned)((t.y_high * width + t.x_low) * c4n);
        b.o[4 * s + 3] = (unsigned)((t.y_high * width + t.x_high) * c4n);
        b.w[4 * s + 0] = ok ? t.w1 : 0.f; b.w[4 * s + 1] = ok ? t.w2 : 0.f;
        b.w[4 * s + 2] = ok ? t.w3 : 0.f; b.w[4 * s + 3] = ok ? t.w4 : 0.f;
    }
}

__global__ void __launch_bounds__(256)
fpn_roi_align_nhwc_roi_kernel(const FpnFeat ft, const int channels, const float *__restrict__ rois, const int num_rois,
                              const int *__restrict__ num_rois_dev, const int pooled_h, const int pooled_w,
                              float *__restrict__ out, int *__restrict__ levels_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x;
    const int nvalid = num_rois_dev ? min(*num_rois_dev, num_rois) : num_rois;
    const int c4n = channels >> 2;
    const int nbins_all = pooled_h * pooled_w;
    // blockIdx.y splits the bins of a ROI when there are few ROIs (mask head: 100-200 ROIs x 196 bins)
    const int per = (nbins_all + gridDim.y - 1) / gridDim.y;
    const int bin0 = blockIdx.y * per, nbins = min(nbins_all, bin0 + per);
    float4 *o4 = reinterpret_cast<float4 *>(out + (long)n * nbins_all * channels);
    if (n >= nvalid) {  // padded tail of a fixed-size roi buffer: defined output (zeros)
        for (int i = bin0 * c4n + threadIdx.x; i < nbins * c4n; i += blockDim.x) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float *r = rois + (long)n * 5;
    const float rx1 = r[1], ry1 = r[2], rx2 = r[3], ry2 = r[4];
    const int lvl = fpn_level_of(rx1, ry1, rx2, ry2);
    if (levels_out && threadIdx.x == 0 && blockIdx.y == 0) levels_out[n] = lvl;
    const float spatial_scale = ft.scale[lvl];
    const int height = ft.h[lvl], width = ft.w[lvl];
    const float4 *__restrict__ feat = reinterpret_cast<const float4 *>(ft.ptr[lvl]);
    const float roi_start_w = rx1 * spatial_scale, roi_start_h = ry1 * spatial_scale;
    const float roi_end_w = rx2 * spatial_scale, roi_end_h = ry2 * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;

    for (int c4 = lane; c4 < c4n; c4 += 64) {
        const float4 *f = feat + c4;
        RoiBinG2 cur, nxt;
        float4 v[16], u[16];
        int bin = bin0 + wave;
        if (bin < nbins) {
            roi_bin_setup_g2(c4n, height, width, roi_start_h, roi_start_w, bin_size_h, bin_size_w, bin / pooled_w, bin % pooled_w, cur);
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = f[cur.o[q]];
        }
        for (; bin < nbins; bin += 4) {
            const int nb = bin + 4;
            if (nb < nbins) {
                roi_bin_setup_g2(c4n, height, width, roi_start_h, roi_start_w, bin_size_h, bin_size_w, nb / pooled_w, nb % pooled_w, nxt);
#pragma unroll
                for (int q = 0; q < 16; ++q) u[q] = f[nxt.o[q]];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {   // reference order: sample (iy, ix), corners 1..4 (roi_align_kernel.cu:199-231)
                float4 val;
                val.x = cur.w[4 * sq] * v[4 * sq].x; val.y = cur.w[4 * sq] * v[4 * sq].y; val.z = cur.w[4 * sq] * v[4 * sq].z; val.w = cur.w[4 * sq] * v[4 * sq].w;
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    val.x = val.x + cur.w[4 * sq + q] * v[4 * sq + q].x; val.y = val.y + cur.w[4 * sq + q] * v[4 * sq + q].y;
                    val.z = val.z + cur.w[4 * sq + q] * v[4 * sq + q].z; val.w = val.w + cur.w[4 * sq + q] * v[4 * sq + q].w;
                }
                acc.x += val.x; acc.y += val.y; acc.z += val.z; acc.w += val.w;
            }
            acc.x /= 4.0f; acc.y /= 4.0f; acc.z /= 4.0f; acc.w /= 4.0f;
            o4[(long)bin * c4n + c4] = acc;
            if (nb < nbins) {
                cur = nxt;
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = u[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// r08: one workgroup per ROI with a SEPARABLE TAP TABLE in LDS (sampling_ratio == 2).
//
// The 2*PH sample rows and 2*PW sample columns of a ROI fully determine its PH*PW*4 samples: a sample's corner offsets are
// (row offset + column offset) and its bilinear weights the products hy*hx, hy*lx, ly*hx, ly*lx of two per-axis factors -- the very
// expressions of roi_align_kernel.cu:43-95, so the values are bit-identical. The table (16 bytes per sample row / column: two byte
// offsets, two weights; weights 0 for samples outside the map) is built ONCE per workgroup by 2*(PH+PW) threads instead of once per
// (bin, wave) by all 64 lanes (the r03-r07 kernel spent ~300 VALU instructions per bin on it and 216 registers: 2 waves per SIMD,
// VALU active 40 % of the cycles of a kernel that should only move bytes).
// A wave owns one bin at a time: its 16 corner vectors (16 x 1 KiB for 256 channels) are buffer loads whose tap offset is WAVE-UNIFORM,
// i.e. a scalar register (soffset) -- no per-lane address arithmetic at all; the lane offset (16 bytes per lane) is loop-invariant.
// SETS == 1: one register set, <= 128 registers = 4 waves per SIMD (4 workgroups per CU: 1024 ROI slots, the box head's 1000 ROIs are
// ONE round), the other waves' loads cover a wave's blend. SETS == 2: two statically named sets, the loads of bin i+1 issued before
// bin i is blended, 3 waves per SIMD.
struct RoiAxis {
    unsigned lo, hi;   // byte offset of the low / high row (or column) inside the level's NHWC map
    float l, h;        // ly, hy (lx, hx); both 0 for a sample outside the map
};

__device__ static inline RoiAxis roi_axis(const int size, float y, const unsigned pitch_bytes)
{
    const bool empty = y < -1.0f || y > (float)size;     // roi_align_kernel.cu:51-55 (one axis of the joint test)
    if (y <= 0) y = 0;
    int lo = (int)y, hi;
    if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else { hi = lo + 1; }
    const float l = y - (float)lo, h = 1.0f - l;
    RoiAxis a;
    a.lo = (unsigned)lo * pitch_bytes; a.hi = (unsigned)hi * pitch_bytes;
    a.l = empty ? 0.f : l; a.h = empty ? 0.f : h;
    return a;
}

#define ROI_MAXP 32   // pooled_h, pooled_w <= 32 on this path

struct RoiTaps {       // one bin: 16 wave-uniform corner offsets (scalar registers) + the per-axis weight factors of its 2 x 2 samples
    unsigned o[16];
    float yl[2], yh[2], xl[2], xh[2];
};
// weight of corner q (0..3 = v1..v4) of sample s (= iy * 2 + ix): hy*hx, hy*lx, ly*hx, ly*lx (roi_align_kernel.cu:84-88)
#define ROI_W(T, S, Q) ((((Q) & 2) ? (T).yl[(S) >> 1] : (T).yh[(S) >> 1]) * (((Q) & 1) ? (T).xl[(S) & 1] : (T).xh[(S) & 1]))

__device__ static inline void roi_bin_taps(const RoiAxis *__restrict__ ytab, const RoiAxis *__restrict__ xtab, const int ph, const int pw, RoiTaps &t)
{
    RoiAxis y[2], x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // (uniform addresses: LDS broadcast reads; offsets go to scalar registers)
        y[i] = ytab[2 * ph + i];
        x[i] = xtab[2 * pw + i];
        y[i].lo = __builtin_amdgcn_readfirstlane(y[i].lo); y[i].hi = __builtin_amdgcn_readfirstlane(y[i].hi);
        x[i].lo = __builtin_amdgcn_readfirstlane(x[i].lo); x[i].hi = __builtin_amdgcn_readfirstlane(x[i].hi);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // sample (iy, ix) = (s >> 1, s & 1); corners in the reference's order v1..v4
        const RoiAxis &a = y[s >> 1], &b = x[s & 1];
        t.o[4 * s + 0] = a.lo + b.lo; t.o[4 * s + 1] = a.lo + b.hi; t.o[4 * s + 2] = a.hi + b.lo; t.o[4 * s + 3] = a.hi + b.hi;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { t.yl[i] = y[i].l; t.yh[i] = y[i].h; t.xl[i] = x[i].l; t.xh[i] = x[i].h; }
}

typedef unsigned roi_uintx4 __attribute__((ext_vector_type(4)));
typedef float roi_f2 __attribute__((ext_vector_type(2)));

#define ROI_LOAD16(V, T) _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                      \
        const roi_uintx4 r_ = __builtin_amdgcn_raw_buffer_load_b128(frsrc, lane_off, (T).o[q_], 0);                 \
        V[q_] = make_float4(__uint_as_float(r_.x), __uint_as_float(r_.y), __uint_as_float(r_.z), __uint_as_float(r_.w)); }   \
    __builtin_amdgcn_sched_barrier(0);   /* all 16 loads are issued before anything of the blend (the scheduler would otherwise trade them for occupancy) */

// reference order: sample (iy, ix), corners 1..4, then the mean over the 4 samples (roi_align_kernel.cu:199-231)
#define ROI_BLEND_STORE(V, T, BIN) {                                                                              \
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);                                                             \
        _Pragma("unroll") for (int sq = 0; sq < 4; ++sq) {                                                        \
            float4 val;                                                                                           \
            const float w0_ = ROI_W(T, sq, 0);                                                                    \
            val.x = w0_ * V[4 * sq].x; val.y = w0_ * V[4 * sq].y; val.z = w0_ * V[4 * sq].z; val.w = w0_ * V[4 * sq].w; \
            _Pragma("unroll") for (int q = 1; q < 4; ++q) {                                                       \
                const float wq_ = ROI_W(T, sq, q);                                                                \
                val.x = val.x + wq_ * V[4 * sq + q].x; val.y = val.y + wq_ * V[4 * sq + q].y;                     \
                val.z = val.z + wq_ * V[4 * sq + q].z; val.w = val.w + wq_ * V[4 * sq + q].w;                     \
            }                                                                                                     \
            acc.x += val.x; acc.y += val.y; acc.z += val.z; acc.w += val.w;                                       \
        }                                                                                                         \
        acc.x /= 4.0f; acc.y /= 4.0f; acc.z /= 4.0f; acc.w /= 4.0f;                                               \
        roi_uintx4 s_;                                                                                            \
        s_.x = __float_as_uint(acc.x); s_.y = __float_as_uint(acc.y); s_.z = __float_as_uint(acc.z); s_.w = __float_as_uint(acc.w); \
        __builtin_amdgcn_raw_buffer_store_b128(s_, orsrc, lane_off, (unsigned)(BIN) * row_bytes, 0); }

// r11: CORNER SHARING. The 2 x 2 samples of a bin lie bin_size / 2 apart -- 1-2 cells for a ROI at the FPN level its size assigns it
// (14-28 cells across, 14 samples per axis) -- so the 4 x 4 corner cells of a bin are rarely 16 different ones: per axis the two samples
// touch rows {lo0, hi0, lo1, hi1}, and either all four differ (class 0), or hi0 == lo1 (the second sample sits in the next cell: class
// 1, 3 unique rows), or both samples sit in the same cell (class 2, 2 unique rows). The kernel's time goes with the number of corner
// loads it issues (measured r10: 42 us + 2.15 us x loads per bin at 1000 x 7 x 7), so a bin now loads its UNIQUE cells only: (4 - ry) x
// (4 - rx) vectors instead of 16 (11.8 on average for ROIs spread log-uniformly over a level's size range). The tap offsets are
// wave-uniform scalars, so the class of a bin is a scalar branch and each of the 9 (ry, rx) bodies names its registers statically;
// every sample is still blended from its own four corners with its own four weights in the reference's order
// (roi_align_kernel.cu:84-95,199-231) -- the values are the same registers' worth of the same cells: bit-identical.
// Any other coincidence (lo == hi at the last row / column of the map) is treated as "all different": redundant loads, same values.
template <int RY, int RX, int PK>
__device__ __forceinline__ void roi_bin_shared(const __amdgpu_buffer_rsrc_t frsrc, const __amdgpu_buffer_rsrc_t orsrc, const unsigned lane_off,
                                               const RoiAxis (&y)[2], const RoiAxis (&x)[2], const unsigned out_off)
{
    // slot of (sample i, low / high) among the unique rows (columns) of the class
    constexpr int NR = 4 - RY, NC = 4 - RX;
    constexpr int rs[3][4] = {{0, 1, 2, 3}, {0, 1, 1, 2}, {0, 1, 0, 1}};     // [class][2 i + (0: low, 1: high)]
    unsigned ro[4], co[4];
    ro[rs[RY][0]] = y[0].lo; ro[rs[RY][1]] = y[0].hi; ro[rs[RY][2]] = y[1].lo; ro[rs[RY][3]] = y[1].hi;
    co[rs[RX][0]] = x[0].lo; co[rs[RX][1]] = x[0].hi; co[rs[RX][2]] = x[1].lo; co[rs[RX][3]] = x[1].hi;
    float4 v[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const roi_uintx4 q_ = __builtin_amdgcn_raw_buffer_load_b128(frsrc, lane_off, ro[r] + co[c], 0);
            v[r][c] = make_float4(__uint_as_float(q_.x), __uint_as_float(q_.y), __uint_as_float(q_.z), __uint_as_float(q_.w));
        }
    __builtin_amdgcn_sched_barrier(0);   // every load of the bin is issued before anything of the blend
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {     // sample (iy, ix) = (sq >> 1, sq & 1); corners v1..v4 = (low, low), (low, high), (high, low), (high, high)
        const int iy = sq >> 1, ix = sq & 1;
        const float wy[2] = {y[iy].h, y[iy].l}, wx[2] = {x[ix].h, x[ix].l};   // hy, ly / hx, lx
        float4 val;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float w = wy[q >> 1] * wx[q & 1];                            // hy*hx, hy*lx, ly*hx, ly*lx (roi_align_kernel.cu:84-88)
            const float4 &c = v[rs[RY][2 * iy + (q >> 1)]][rs[RX][2 * ix + (q & 1)]];
            if (PK) {
                const roi_f2 w2 = {w, w};
                const roi_f2 lo = w2 * roi_f2{c.x, c.y}, hi = w2 * roi_f2{c.z, c.w};
                if (q == 0) { val.x = lo.x; val.y = lo.y; val.z = hi.x; val.w = hi.y; }
                else {
                    const roi_f2 a = roi_f2{val.x, val.y} + lo, b = roi_f2{val.z, val.w} + hi;
                    val.x = a.x; val.y = a.y; val.z = b.x; val.w = b.y;
                }
            } else if (q == 0) {
                val.x = w * c.x; val.y = w * c.y; val.z = w * c.z; val.w = w * c.w;
            } else {
                val.x = val.x + w * c.x; val.y = val.y + w * c.y; val.z = val.z + w * c.z; val.w = val.w + w * c.w;
            }
        }
        if (PK) {
            const roi_f2 a = roi_f2{acc.x, acc.y} + roi_f2{val.x, val.y}, b = roi_f2{acc.z, acc.w} + roi_f2{val.z, val.w};
            acc.x = a.x; acc.y = a.y; acc.z = b.x; acc.w = b.y;
        } else {
            acc.x += val.x; acc.y += val.y; acc.z += val.z; acc.w += val.w;
        }
    }
    acc.x /= 4.0f; acc.y /= 4.0f; acc.z /= 4.0f; acc.w /= 4.0f;
    roi_uintx4 s_;
    s_.x = __float_as_uint(acc.x); s_.y = __float_as_uint(acc.y); s_.z = __float_as_uint(acc.z); s_.w = __float_as_uint(acc.w);
    __builtin_amdgcn_raw_buffer_store_b128(s_, orsrc, lane_off, out_off, 0);
}

// the class of one axis of a bin from its (wave-uniform) row / column offsets
__device__ __forceinline__ int roi_share_class(const RoiAxis (&a)[2])
{
    return (a[1].lo == a[0].lo && a[1].hi == a[0].hi) ? 2 : (a[1].lo == a[0].hi ? 1 : 0);
}

template <int SETS, int SHARE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SETS == 1 ? 4 : 3, SETS == 1 ? 5 : 3)))
fpn_roi_align_nhwc_tab_kernel(const FpnFeat ft, const int channels, const float *__restrict__ rois, const int num_rois,
                              const int *__restrict__ num_rois_dev, const int pooled_h, const int pooled_w,
                              float *__restrict__ out, int *__restrict__ levels_out, const int *__restrict__ order)
{
    __shared__ RoiAxis ytab[2 * ROI_MAXP], xtab[2 * ROI_MAXP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int)blockIdx.x >= num_rois) return;        // (grid.x is padded to a multiple of 8: workgroup (x, y) then runs on XCD x % 8 for every y)
    // which ROI this workgroup takes: its own index, or -- r13 -- the entry of the XCD dealing table (fpn_roi_order_kernel): output rows
    // stay at the ROI's own index, only the workgroup -> XCD assignment changes
    const int n = order ? __builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int)blockIdx.x;
    const int nvalid = num_rois_dev ? min(*num_rois_dev, num_rois) : num_rois;
    const int c4n = channels >> 2;
    const int nbins_all = pooled_h * pooled_w;
    const int per = (nbins_all + gridDim.y - 1) / gridDim.y;   // blockIdx.y splits the bins of a ROI when there are few ROIs
    const int bin0 = blockIdx.y * per, nbins = min(nbins_all, bin0 + per);
    if (n >= nvalid) {  // padded tail of a fixed-size roi buffer: defined output (zeros)
        float4 *o4 = reinterpret_cast<float4 *>(out + (long)n * nbins_all * channels);
        for (int i = bin0 * c4n + threadIdx.x; i < nbins * c4n; i += blockDim.x) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float *r = rois + (long)n * 5;
    const float rx1 = r[1], ry1 = r[2], rx2 = r[3], ry2 = r[4];
    const int lvl = __builtin_amdgcn_readfirstlane(fpn_level_of(rx1, ry1, rx2, ry2));
    if (levels_out && threadIdx.x == 0 && blockIdx.y == 0) levels_out[n] = lvl;
    const float spatial_scale = ft.scale[lvl];
    const int height = ft.h[lvl], width = ft.w[lvl];
    const unsigned row_bytes = (unsigned)channels * 4u;
    {   // the tap table: thread s < 2*PH -> sample row s, thread 64 + s (s < 2*PW) -> sample column s
        const float roi_start_w = rx1 * spatial_scale, roi_start_h = ry1 * spatial_scale;
        const float roi_end_w = rx2 * spatial_scale, roi_end_h = ry2 * spatial_scale;
        const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
        const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
        const float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;
        const int s = threadIdx.x & 63;
        if (threadIdx.x < 64) {
            if (s < 2 * pooled_h) {
                const float y = roi_start_h + (float)(s >> 1) * bin_size_h + ((float)(s & 1) + .5f) * bin_size_h / 2.0f;
                ytab[s] = roi_axis(height, y, (unsigned)width * row_bytes);
            }
        } else if (threadIdx.x < 128) {
            if (s < 2 * pooled_w) {
                const float x = roi_start_w + (float)(s >> 1) * bin_size_w + ((float)(s & 1) + .5f) * bin_size_w / 2.0f;
                xtab[s] = roi_axis(width, x, row_bytes);
            }
        }
    }
    __syncthreads();
    // buffer descriptors: the level's feature map (reads) and this ROI's output rows (stores)
    const size_t fbase = (size_t)ft.ptr[lvl];
    const unsigned flo = __builtin_amdgcn_readfirstlane((unsigned)fbase), fhi = __builtin_amdgcn_readfirstlane((unsigned)(fbase >> 32));
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)fhi << 32) | flo), 0,
                                                                           (int)((unsigned)height * (unsigned)width * row_bytes), 0x00020000);
    const size_t obase = (size_t)(out + (long)n * nbins_all * channels);
    const unsigned olo = __builtin_amdgcn_readfirstlane((unsigned)obase), ohi = __builtin_amdgcn_readfirstlane((unsigned)(obase >> 32));
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)ohi << 32) | olo), 0,
                                                                           (int)((unsigned)nbins_all * row_bytes), 0x00020000);
    for (int c4 = lane; c4 < c4n; c4 += 64) {          // 256 channels: one pass
        const unsigned lane_off = (unsigned)c4 * 16u;
        int bin = bin0 + wave, ph = bin / pooled_w, pw = bin - ph * pooled_w;
#define ROI_ADVANCE() { bin += 4; pw += 4; while (pw >= pooled_w) { pw -= pooled_w; ++ph; } }
        if (SHARE) {
            for (; bin < nbins;) {
                RoiAxis y[2], x[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {   // (uniform addresses: LDS broadcast reads; offsets go to scalar registers)
                    y[i] = ytab[2 * ph + i];
                    x[i] = xtab[2 * pw + i];
                    y[i].lo = __builtin_amdgcn_readfirstlane(y[i].lo); y[i].hi = __builtin_amdgcn_readfirstlane(y[i].hi);
                    x[i].lo = __builtin_amdgcn_readfirstlane(x[i].lo); x[i].hi = __builtin_amdgcn_readfirstlane(x[i].hi);
                }
                const int cls = roi_share_class(y) * 3 + roi_share_class(x);
                const unsigned oo = (unsigned)bin * row_bytes;
                constexpr int PK = SHARE == 2;
                switch (cls) {
                case 0: roi_bin_shared<0, 0, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 1: roi_bin_shared<0, 1, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 2: roi_bin_shared<0, 2, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 3: roi_bin_shared<1, 0, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 4: roi_bin_shared<1, 1, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 5: roi_bin_shared<1, 2, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 6: roi_bin_shared<2, 0, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                case 7: roi_bin_shared<2, 1, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                default: roi_bin_shared<2, 2, PK>(frsrc, orsrc, lane_off, y, x, oo); break;
                }
                ROI_ADVANCE()
            }
        } else if (SETS == 1) {
            for (; bin < nbins;) {
                RoiTaps t;
                float4 v[16];
                roi_bin_taps(ytab, xtab, ph, pw, t);
                ROI_LOAD16(v, t)
                ROI_BLEND_STORE(v, t, bin)
                ROI_ADVANCE()
            }
        } else {
            RoiTaps ta, tb;
            float4 v[16], u[16];
            if (bin < nbins) {
                roi_bin_taps(ytab, xtab, ph, pw, ta);
                ROI_LOAD16(v, ta)
            }
            while (bin < nbins) {
                const int bin_a = bin;
                ROI_ADVANCE()
                const bool has_b = bin < nbins;
                if (has_b) {
                    roi_bin_taps(ytab, xtab, ph, pw, tb);
                    ROI_LOAD16(u, tb)
                }
                ROI_BLEND_STORE(v, ta, bin_a)
                if (!has_b) break;
                const int bin_b = bin;
                ROI_ADVANCE()
                if (bin < nbins) {
                    roi_bin_taps(ytab, xtab, ph, pw, ta);
                    ROI_LOAD16(v, ta)
                }
                ROI_BLEND_STORE(u, tb, bin_b)
            }
        }
#undef ROI_ADVANCE
    }
}

// 0: table kernel, one register set; 1: table kernel, two sets; 2: the r03-r07 per-ROI kernel; 3: table kernel loading only the UNIQUE corner
// cells of a bin (r11); 4: the same with packed fp32 blend arithmetic; A/B knob (also env UPSNET_ROI_KERNEL)
static int g_roi_variant = -1;
static int g_roi_target_wg = 1536, g_roi_min_bins = 8;
// >= 0: that variant, whatever the environment says; < 0: back to the default behaviour (the environment variable is read again at the
// next launch: UPSNET_ROI_KERNEL = a variant number, or unset / "auto" / negative = the automatic choice)
extern "C" void upsnet_roi_tuning(int variant) { g_roi_variant = variant < 0 ? -1 : variant; }
// development knob: the bins of a ROI are split over workgroups until the launch has `target_workgroups`, `min_bins` bins each at least
extern "C" void upsnet_roi_geometry(int target_workgroups, int min_bins)
{
    g_roi_target_wg = target_workgroups > 0 ? target_workgroups : 1536;
    g_roi_min_bins = min_bins > 0 ? min_bins : 8;
}
static int roi_variant(const int bins)
{
    if (g_roi_variant == -1) {
        const char *e = getenv("UPSNET_ROI_KERNEL");
        g_roi_variant = (e && *e && (*e >= '0' && *e <= '9')) ? atoi(e) : -2;     // unset, "auto", "-1": automatic
    }
    if (g_roi_variant >= 0) return g_roi_variant;
    // auto (r11, measured on random ROIs, profiles/r11_roialign.txt): the corner-sharing form where the samples of a bin usually fall into
    // shared cells -- 14 x 14 bins: half a cell apart, 21.8 vs 24.0 us at 100 ROIs --, the plain table form for 7 x 7 bins (a cell apart:
    // the duplicates it issues hit L1 anyway -- same L1 misses, PMC -- and its straight-line body is 3 us faster at 1000 ROIs)
    return bins >= 100 ? 3 : 0;
}

// ---- ROI -> XCD dealing (r13). Workgroup b of the ROIAlign launch runs on XCD b % 8 and every XCD has its own L2: with the ROIs in score
// order (proposals) or random order, each of the eight L2s fetches its own copy of the pyramid cells its ROIs share with the other XCDs'
// ROIs -- 346 MB fetched at the fabric for a 178 MB pyramid on the 1000 x 7 x 7 launch (profiles/r11_roialign_pmc.txt), which is what
// bounds the kernel. This kernel orders the ROIs by (pyramid level, image stripe of the centre, column cell) and deals the ordered
// list so that XCD j's workgroups (b = j, j + 8, ...) take one contiguous range of it: neighbours in the image share an L2.
// order[b] = ROI index workgroup b processes; the output rows are untouched (bit-identical results). One workgroup, <= 2048 ROIs
// (roi_order.h; the proposals' table is produced inside prop_merge_kernel, proposal.hip).
__global__ void __launch_bounds__(1024)
fpn_roi_order_kernel(const float *__restrict__ rois, const int num_rois, const int *__restrict__ num_rois_dev, const float inv_stripe_h,
                     const float inv_cell_w, int *__restrict__ order)
{
    __shared__ int cnt[ROI_ORDER_LDS];
    const int nvalid = num_rois_dev ? min(*num_rois_dev, num_rois) : num_rois;
    ups_roi_order_block(rois, num_rois, nvalid, inv_stripe_h, inv_cell_w, order, cnt);      // (roi_order.h)
}

/* order_out[b] = the ROI workgroup b of upsnet_fpn_roi_align_forward_ordered should take so that each XCD's workgroups cover one contiguous
 * range of the ROIs bucketed by (pyramid level, 1/16 stripe of the image, 1/8 column cell of the image). rois [num_rois, 5] device in image
 * pixels, num_rois <= 2048; image_height / image_width: the extent the ROIs live in (only the bucket granularity depends on it). */
extern "C" int upsnet_fpn_roi_order(void *stream, const float *rois, int num_rois, const int *num_rois_dev, int image_height, int image_width,
                                    int *order_out)
{
    UPS_REQUIRE(rois && order_out && num_rois >= 0 && num_rois <= ROI_ORDER_MAX, "fpn_roi_order: 0..%d rois (got %d)", ROI_ORDER_MAX, num_rois);
    UPS_REQUIRE(image_height > 0 && image_width > 0, "fpn_roi_order: bad image extent");
    if (num_rois == 0) return 0;
    hipLaunchKernelGGL(fpn_roi_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rois, num_rois, num_rois_dev, 16.0f / (float)image_height,
                       8.0f / (float)image_width, order_out);
    UPS_CHECK_LAUNCH("fpn_roi_order_kernel");
    return 0;
}

extern "C" int upsnet_fpn_roi_align_forward_ordered(void *stream, const float *const feat_nhwc[4], const int feat_h[4],
                                                    const int feat_w[4], const float spatial_scale[4], int channels,
                                                    const float *rois, int num_rois, const int *num_rois_dev,
                                                    int pooled_height, int pooled_width, int sampling_ratio,
                                                    float *out_nhwc, int *levels_out, const int *order);

extern "C" int upsnet_fpn_roi_align_forward(void *stream, const float *const feat_nhwc[4], const int feat_h[4],
                                            const int feat_w[4], const float spatial_scale[4], int channels,
                                            const float *rois, int num_rois, const int *num_rois_dev,
                                            int pooled_height, int pooled_width, int sampling_ratio,
                                            float *out_nhwc, int *levels_out)
{
    return upsnet_fpn_roi_align_forward_ordered(stream, feat_nhwc, feat_h, feat_w, spatial_scale, channels, rois, num_rois, num_rois_dev,
                                                pooled_height, pooled_width, sampling_ratio, out_nhwc, levels_out, nullptr);
}

extern "C" int upsnet_fpn_roi_align_forward_ordered(void *stream, const float *const feat_nhwc[4], const int feat_h[4],
                                                    const int feat_w[4], const float spatial_scale[4], int channels,
                                                    const float *rois, int num_rois, const int *num_rois_dev,
                                                    int pooled_height, int pooled_width, int sampling_ratio,
                                                    float *out_nhwc, int *levels_out, const int *order)
{
    UPS_REQUIRE(feat_nhwc && rois && out_nhwc, "fpn_roi_align_forward: null pointer");
    UPS_REQUIRE(channels > 0 && (channels & 3) == 0, "fpn_roi_align_forward: channels must be a multiple of 4 (got %d)", channels);
    UPS_REQUIRE(num_rois >= 0 && pooled_height > 0 && pooled_width > 0, "fpn_roi_align_forward: bad shape");
    if (num_rois == 0) return 0;
    FpnFeat ft;
    for (int i = 0; i < 4; ++i) {
        UPS_REQUIRE(feat_nhwc[i] && feat_h[i] > 0 && feat_w[i] > 0, "fpn_roi_align_forward: bad level %d", i);
        ft.ptr[i] = feat_nhwc[i]; ft.h[i] = feat_h[i]; ft.w[i] = feat_w[i]; ft.scale[i] = spatial_scale[i];
    }
    bool small = true;   // 32-bit float4 offsets
    for (int i = 0; i < 4; ++i) small = small && (long)feat_h[i] * feat_w[i] * (channels >> 2) < (1L << 31);
    if (sampling_ratio == 2 && small && roi_per_bin() == 0) {
        const int nb = pooled_height * pooled_width;
        int nsplit = (g_roi_target_wg + num_rois - 1) / num_rois;   // aim at >= 1536 workgroups, >= 8 bins each
        if (nsplit > nb / g_roi_min_bins) nsplit = nb / g_roi_min_bins;
        if (nsplit < 1) nsplit = 1;
        const int variant = (pooled_height <= ROI_MAXP && pooled_width <= ROI_MAXP) ? roi_variant(nb) : 2;
        const int nx8 = (num_rois + 7) / 8 * 8;
        // (variant 2, the r03-r07 kernel kept for A/B runs, has no table: it runs in ROI order -- same bits)
        if (variant == 3 || variant == 4) {
            if (variant == 3)
                hipLaunchKernelGGL((fpn_roi_align_nhwc_tab_kernel<1, 1>), dim3(nx8, nsplit), dim3(256), 0, (hipStream_t)stream, ft, channels, rois, num_rois,
                                   num_rois_dev, pooled_height, pooled_width, out_nhwc, levels_out, order);
            else
                hipLaunchKernelGGL((fpn_roi_align_nhwc_tab_kernel<1, 2>), dim3(nx8, nsplit), dim3(256), 0, (hipStream_t)stream, ft, channels, rois, num_rois,
                                   num_rois_dev, pooled_height, pooled_width, out_nhwc, levels_out, order);
        } else if (variant == 0)
            hipLaunchKernelGGL((fpn_roi_align_nhwc_tab_kernel<1, 0>), dim3(nx8, nsplit), dim3(256), 0, (hipStream_t)stream, ft, channels, rois, num_rois,
                               num_rois_dev, pooled_height, pooled_width, out_nhwc, levels_out, order);
        else if (variant == 1)
            hipLaunchKernelGGL((fpn_roi_align_nhwc_tab_kernel<2, 0>), dim3(nx8, nsplit), dim3(256), 0, (hipStream_t)stream, ft, channels, rois, num_rois,
                               num_rois_dev, pooled_height, pooled_width, out_nhwc, levels_out, order);
        else
            hipLaunchKernelGGL(fpn_roi_align_nhwc_roi_kernel, dim3(num_rois, nsplit), dim3(256), 0, (hipStream_t)stream, ft, channels, rois, num_rois,
                               num_rois_dev, pooled_height, pooled_width, out_nhwc, levels_out);
        UPS_CHECK_LAUNCH("fpn_roi_align_nhwc (per-ROI workgroups)");
        return 0;
    }
    long bins = (long)num_rois * pooled_height * pooled_width;
    int blocks = (int)((bins + 3) / 4);
    hipLaunchKernelGGL(fpn_roi_align_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ft, channels,
                       rois, num_rois, num_rois_dev, pooled_height, pooled_width, sampling_ratio, out_nhwc, levels_out);
    UPS_CHECK_LAUNCH("fpn_roi_align_nhwc_kernel");
    return 0;
}

// Single-level NHWC variant (RoIAlign module on channels_last features): same kernel, level forced.
__global__ void __launch_bounds__(256)
roi_align_nhwc_kernel(const float *__restrict__ feat_, const int channels, const int height, const int width,
                      const float spatial_scale, const float *__restrict__ rois, const int num_rois,
                      const int pooled_h, const int pooled_w, const int sampling_ratio, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long bin = __builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (bin >= (long)num_rois * pooled_h * pooled_w) return;
    const int pw = bin % pooled_w;
    const int ph = (bin / pooled_w) % pooled_h;
    const int n = bin / ((long)pooled_w * pooled_h);
    const int c4n = channels >> 2;
    const float *r = rois + (long)n * 5;
    const int roi_batch_ind = (int)roundf(r[0]);
    const float4 *feat = reinterpret_cast<const float4 *>(feat_ + (long)roi_batch_ind * height * width * channels);
    const float roi_start_w = r[1] * spatial_scale, roi_start_h = r[2] * spatial_scale;
    const float roi_end_w = r[3] * spatial_scale, roi_end_h = r[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
    const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)pooled_h, bin_size_w = roi_width / (float)pooled_w;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)pooled_h);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)pooled_w);
    const float count = (float)(grid_h * grid_w);
    float4 *o4 = reinterpret_cast<float4 *>(out + bin * channels);
    if (sampling_ratio == 2 && (long)height * width * c4n < (1L << 31)) {
        roi_bin_nhwc_g2(feat, c4n, height, width, roi_start_h, roi_start_w, bin_size_h, bin_size_w, ph, pw, lane, o4);
        return;
    }
    for (int c4 = lane; c4 < c4n; c4 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < grid_h; ++iy) {
            const float y = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                const float x = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
                RoiTap t;
                if (roi_tap(height, width, y, x, t)) {
                    const float4 v1 = feat[((long)t.y_low * width + t.x_low) * c4n + c4];
                    const float4 v2 = feat[((long)t.y_low * width + t.x_high) * c4n + c4];
                    const float4 v3 = feat[((long)t.y_high * width + t.x_low) * c4n + c4];
                    const float4 v4 = feat[((long)t.y_high * width + t.x_high) * c4n + c4];
                    acc.x += roi_blend(t, v1.x, v2.x, v3.x, v4.x);
                    acc.y += roi_blend(t, v1.y, v2.y, v3.y, v4.y);
                    acc.z += roi_blend(t, v1.z, v2.z, v3.z, v4.z);
                    acc.w += roi_blend(t, v1.w, v2.w, v3.w, v4.w);
                }
            }
        }
        acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        o4[c4] = acc;
    }
}

extern "C" int upsnet_roi_align_forward_nhwc(void *stream, const float *feat_nhwc, int batch, int height, int width,
                                             int channels, float spatial_scale, const float *rois, int num_rois,
                                             int pooled_height, int pooled_width, int sampling_ratio, float *out_nhwc)
{
    (void)batch;
    UPS_REQUIRE(feat_nhwc && rois && out_nhwc, "roi_align_forward_nhwc: null pointer");
    UPS_REQUIRE(channels > 0 && (channels & 3) == 0, "roi_align_forward_nhwc: channels must be a multiple of 4 (got %d)", channels);
    if (num_rois == 0) return 0;
    long bins = (long)num_rois * pooled_height * pooled_width;
    int blocks = (int)((bins + 3) / 4);
    hipLaunchKernelGGL(roi_align_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feat_nhwc, channels,
                       height, width, spatial_scale, rois, num_rois, pooled_height, pooled_width, sampling_ratio, out_nhwc);
    UPS_CHECK_LAUNCH("roi_align_nhwc_kernel");
    return 0;
}
