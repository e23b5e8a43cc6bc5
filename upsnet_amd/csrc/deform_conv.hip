// deform_conv.hip -- deformable convolution v1 / v2 forward for gfx950.
//
//  * upsnet_deform_im2col / upsnet_mod_deform_im2col : NCHW drop-ins for the reference launchers
//    (deform_conv_kernel.cu:194-285, mod_deform_conv_kernel.cu:187-249,383-407): same column-buffer
//    layout, one thread per (c, b, h_col, w_col).
//  * upsnet_deform_conv_forward_nhwc : the MI355X-native fused operator. No column buffer in HBM:
//    a workgroup owns 128 output pixels x all Cout; for every (tap, 32-channel slab) the four waves
//    gather the four bilinear corners as contiguous NHWC channel runs, blend them in fp32 exactly as
//    the reference does, park the [32 x 128] sampled tile in LDS and contract it against the packed
//    weight slab with v_mfma_f32_32x32x2_f32 (exact fp32 multiply-add chain). The four FPN levels of
//    the FCN head share weights and are processed by ONE launch.
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

// ---------------------------------------------------------------------------------------------
// weight [Cout, Cin, kh, kw] -> wpack [(tap*Cin + c), Cout]
__global__ void dcn_pack_weight_kernel(const float *__restrict__ w, int cout, int cin, int taps, float *__restrict__ wp)
{
    const long total = (long)cout * cin * taps;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int co = idx % cout;
        const int c = (idx / cout) % cin;
        const int tap = idx / ((long)cout * cin);
        wp[idx] = w[((long)co * cin + c) * taps + tap];
    }
}

extern "C" int upsnet_deform_conv_pack_weight(void *stream, const float *weight, int cout, int cin, int kh, int kw,
                                              float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cin > 0 && kh > 0 && kw > 0, "deform_conv_pack_weight: bad args");
    const long total = (long)cout * cin * kh * kw;
    hipLaunchKernelGGL(dcn_pack_weight_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       cout, cin, kh * kw, wpack);
    UPS_CHECK_LAUNCH("dcn_pack_weight_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Fused NHWC deformable convolution, fp32 MFMA.
#define DCN_BM 128          // output pixels per workgroup
#define DCN_KS 32           // channels per K slab
#define DCN_LDA (DCN_BM + 1) // padded row length of the sampled tile (conflict-free transposed writes)

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct DcnLevels {
    const float *x[4];
    const float *off[4];
    const float *mask[4];
    float *out[4];
    int H[4], W[4], Ho[4], Wo[4];
    int tile_start[5];
    int nlev;
};

template <int NT, bool MOD>
__global__ void __launch_bounds__(256)
dcn_fused_nhwc_kernel(const DcnLevels lv, const int cin, const int kh, const int kw, const int pad_h, const int pad_w,
                      const int stride_h, const int stride_w, const int dil_h, const int dil_w,
                      const float *__restrict__ wpack, const float *__restrict__ bias, const int relu)
{
    constexpr int COUT = NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *As = reinterpret_cast<float *>(smem_raw);              // [DCN_KS][DCN_LDA]
    float *Bs = As + DCN_KS * DCN_LDA;                            // [DCN_KS][COUT]
    int *s_off = reinterpret_cast<int *>(Bs + DCN_KS * COUT);     // [4][DCN_BM] corner element offsets
    float *s_w = reinterpret_cast<float *>(s_off + 4 * DCN_BM);   // [4][DCN_BM] corner weights
    float *s_m = s_w + 4 * DCN_BM;                                // [DCN_BM] modulation (v2)
    unsigned *s_valid = reinterpret_cast<unsigned *>(s_m + DCN_BM); // [DCN_BM] 4 validity bits

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- which level / tile
    int l = 0;
    const int tile = blockIdx.x;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (q < lv.nlev && tile >= lv.tile_start[q]) l = q;
    const int H = lv.H[l], W = lv.W[l], Ho = lv.Ho[l], Wo = lv.Wo[l];
    const long npix = (long)Ho * Wo;
    const long p0 = (long)(tile - lv.tile_start[l]) * DCN_BM;
    const float *__restrict__ x = lv.x[l];
    const float *__restrict__ off = lv.off[l];
    const float *__restrict__ msk = MOD ? lv.mask[l] : nullptr;
    const int ntap = kh * kw;

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ch = tid & 31, pxs = tid >> 5;  // producer mapping: 32 channels x 8 pixel phases
    const int akr = lane >> 5, aij = lane & 31; // MFMA operand mapping (k row, i/j column)

    for (int tap = 0; tap < ntap; ++tap) {
        __syncthreads();  // previous tap's descriptors / tiles no longer in use
        if (tid < DCN_BM) {
            const long p = p0 + tid;
            unsigned vbits = 0;
            int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mm = 1.f;
            if (p < npix) {
                const int ho = (int)(p / Wo), wo = (int)(p % Wo);
                const int ki = tap / kw, kj = tap % kw;
                const float off_h = off[p * (2 * ntap) + 2 * tap];
                const float off_w = off[p * (2 * ntap) + 2 * tap + 1];
                const float h_im = (float)(ho * stride_h - pad_h + ki * dil_h) + off_h;
                const float w_im = (float)(wo * stride_w - pad_w + kj * dil_w) + off_w;
                if (h_im > -1 && w_im > -1 && h_im < (float)H && w_im < (float)W) {
                    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
                    const float hh = 1.0f - lh, hw = 1.0f - lw;
                    w1 = hh * hw; w2 = hh * lw; w3 = lh * hw; w4 = lh * lw;
                    const bool a = h_low >= 0, b = h_high <= H - 1, c = w_low >= 0, d = w_high <= W - 1;
                    const int hl = a ? h_low : 0, hhi = b ? h_high : H - 1, wl = c ? w_low : 0, whi = d ? w_high : W - 1;
                    o1 = (hl * W + wl) * cin; o2 = (hl * W + whi) * cin; o3 = (hhi * W + wl) * cin; o4 = (hhi * W + whi) * cin;
                    vbits = (a && c ? 1u : 0u) | (a && d ? 2u : 0u) | (b && c ? 4u : 0u) | (b && d ? 8u : 0u);
                }
                if (MOD) mm = msk[p * ntap + tap];
            }
            s_off[0 * DCN_BM + tid] = o1; s_off[1 * DCN_BM + tid] = o2; s_off[2 * DCN_BM + tid] = o3; s_off[3 * DCN_BM + tid] = o4;
            s_w[0 * DCN_BM + tid] = w1; s_w[1 * DCN_BM + tid] = w2; s_w[2 * DCN_BM + tid] = w3; s_w[3 * DCN_BM + tid] = w4;
            s_m[tid] = mm;
            s_valid[tid] = vbits;
        }
        __syncthreads();
        for (int cs = 0; cs < cin; cs += DCN_KS) {
            // ---- produce the sampled tile: As[ch][px] = blend(4 corners) for channel cs+ch
            const float *__restrict__ xc = x + cs + ch;
#pragma unroll 4
            for (int i = 0; i < DCN_BM / 8; ++i) {
                const int px = pxs + 8 * i;
                const unsigned vb = s_valid[px];
                const float v1 = (vb & 1u) ? xc[s_off[0 * DCN_BM + px]] : 0.f;
                const float v2 = (vb & 2u) ? xc[s_off[1 * DCN_BM + px]] : 0.f;
                const float v3 = (vb & 4u) ? xc[s_off[2 * DCN_BM + px]] : 0.f;
                const float v4 = (vb & 8u) ? xc[s_off[3 * DCN_BM + px]] : 0.f;
                float val = s_w[0 * DCN_BM + px] * v1;
                val = val + s_w[1 * DCN_BM + px] * v2;
                val = val + s_w[2 * DCN_BM + px] * v3;
                val = val + s_w[3 * DCN_BM + px] * v4;
                if (MOD) val = val * s_m[px];
                As[ch * DCN_LDA + px] = val;
            }
            // ---- stage the weight slab: rows (tap*cin + cs .. +32) of wpack, contiguous 32*COUT floats
            {
                const float4 *__restrict__ src = reinterpret_cast<const float4 *>(wpack + ((long)tap * cin + cs) * COUT);
                float4 *dst = reinterpret_cast<float4 *>(Bs);
#pragma unroll
                for (int q = 0; q < (DCN_KS * COUT / 4) / 256; ++q) dst[tid + 256 * q] = src[tid + 256 * q];
            }
            __syncthreads();
            // ---- contract: 16 k-steps of K=2, NT tiles of 32x32 per wave (wave owns pixels 32*wave..+32)
#pragma unroll 4
            for (int s = 0; s < DCN_KS / 2; ++s) {
                const float a = As[(2 * s + akr) * DCN_LDA + 32 * wave + aij];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float b = Bs[(2 * s + akr) * COUT + 32 * t + aij];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    // ---- epilogue: D layout col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
    float *__restrict__ out = lv.out[l];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = 32 * t + aij;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * akr;
            const long p = p0 + 32 * wave + row;
            if (p < npix) {
                float v = acc[t][r];
                if (bias) v = v + bv;
                if (relu) v = fmaxf(v, 0.f);
                out[p * COUT + co] = v;
            }
        }
    }
}

template <int NT, bool MOD>
static int dcn_launch(hipStream_t st, const DcnLevels &lv, int ntiles, int cin, int kh, int kw, int pad_h, int pad_w,
                      int stride_h, int stride_w, int dil_h, int dil_w, const float *wpack, const float *bias, int relu)
{
    const size_t smem = (size_t)(DCN_KS * DCN_LDA + DCN_KS * NT * 32 + 4 * DCN_BM + 4 * DCN_BM + DCN_BM + DCN_BM) * 4;
    hipLaunchKernelGGL((dcn_fused_nhwc_kernel<NT, MOD>), dim3(ntiles), dim3(256), smem, st, lv, cin, kh, kw, pad_h, pad_w,
                       stride_h, stride_w, dil_h, dil_w, wpack, bias, relu);
    UPS_CHECK_LAUNCH("dcn_fused_nhwc_kernel");
    return 0;
}

extern "C" int upsnet_deform_conv_forward_nhwc(void *stream, int nlev, const float *const x[], const float *const offset[],
                                               const float *const mask[], float *const out[], const int height[],
                                               const int width[], int cin, int cout, int kh, int kw, int pad_h, int pad_w,
                                               int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                                               const float *wpack, const float *bias, int relu)
{
    UPS_REQUIRE(nlev >= 1 && nlev <= 4, "deform_conv_forward_nhwc: nlev must be 1..4 (got %d)", nlev);
    UPS_REQUIRE(x && offset && out && height && width && wpack, "deform_conv_forward_nhwc: null pointer");
    UPS_REQUIRE(deformable_group == 1, "deform_conv_forward_nhwc: deformable_group=%d not supported by the fused kernel (use the im2col path)", deformable_group);
    UPS_REQUIRE(cin > 0 && cin % DCN_KS == 0, "deform_conv_forward_nhwc: Cin must be a multiple of 32 (got %d)", cin);
    UPS_REQUIRE(cout == 32 || cout == 64 || cout == 128 || cout == 256,
                "deform_conv_forward_nhwc: Cout must be 32/64/128/256 (got %d)", cout);
    DcnLevels lv;
    lv.nlev = nlev;
    int tiles = 0;
    for (int l = 0; l < 4; ++l) {
        if (l < nlev) {
            UPS_REQUIRE(x[l] && offset[l] && out[l] && height[l] > 0 && width[l] > 0, "deform_conv_forward_nhwc: bad level %d", l);
            UPS_REQUIRE(!mask || mask[l], "deform_conv_forward_nhwc: null mask at level %d", l);
            lv.x[l] = x[l]; lv.off[l] = offset[l]; lv.mask[l] = mask ? mask[l] : nullptr; lv.out[l] = out[l];
            lv.H[l] = height[l]; lv.W[l] = width[l];
            lv.Ho[l] = (height[l] + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
            lv.Wo[l] = (width[l] + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
            UPS_REQUIRE(lv.Ho[l] > 0 && lv.Wo[l] > 0, "deform_conv_forward_nhwc: empty output at level %d", l);
            UPS_REQUIRE((long)height[l] * width[l] * cin < 2147483647L, "deform_conv_forward_nhwc: level %d too large for 32-bit offsets", l);
            lv.tile_start[l] = tiles;
            tiles += ups_divup((long)lv.Ho[l] * lv.Wo[l], DCN_BM);
        } else {
            lv.x[l] = nullptr; lv.off[l] = nullptr; lv.mask[l] = nullptr; lv.out[l] = nullptr;
            lv.H[l] = lv.W[l] = lv.Ho[l] = lv.Wo[l] = 0;
            lv.tile_start[l] = 0x7fffffff;
        }
    }
    lv.tile_start[4] = tiles;
    hipStream_t st = (hipStream_t)stream;
#define DCN_DISPATCH(NT)                                                                                              \
    return mask ? dcn_launch<NT, true>(st, lv, tiles, cin, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, wpack, bias, relu) \
                : dcn_launch<NT, false>(st, lv, tiles, cin, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, wpack, bias, relu)
    switch (cout / 32) {
    case 1: DCN_DISPATCH(1);
    case 2: DCN_DISPATCH(2);
    case 4: DCN_DISPATCH(4);
    case 8: DCN_DISPATCH(8);
    }
#undef DCN_DISPATCH
    return ups_set_error("deform_conv_forward_nhwc: unreachable");
}
