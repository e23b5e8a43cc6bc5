// conv_bf16.hip -- dense convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
// BASELINE.json configs[2] ("hand-written MFMA 3x3/1x1 backbone + FPN convs, NHWC, bf16 compute / fp32 accumulate") and its
// fp32-equivalent variant. The fp32 kernel (conv.hip) is the default of the framework and the one the headline is measured on;
// this file is opt-in (hipconv.PRECISION / UPSNET_CONV_PRECISION):
//   SPLIT 1 ("bf16")    a, b rounded to bf16 (round to nearest even), one MFMA per product: ~3 significant digits.
//   SPLIT 3 ("bf16x3")  a = a_hi + a_lo, b = b_hi + b_lo (both parts bf16), a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi: the dropped
//                       term is ~2^-16 relative, the sum is accumulated in fp32 -- within the 1e-4 fp32-logit tolerance of the
//                       north star, at 3/16 of the MFMA time of exact fp32 products.
// Activations stay fp32 in HBM (NHWC, same tensors as the fp32 path); they are split to bf16 when the staged registers are
// written to LDS. Weights are split and packed once (upsnet_conv_pack_weight_bf16) as [K slab][column][32 k], so a lane's MFMA
// operand (8 consecutive k of one row / column) is ONE 16-byte LDS read: As/Bs are [64 rows][32 k] bf16 with an 80-byte row
// pitch (conflict-free ds_read_b128). Workgroup = 64 pixels x 64 channels, 4 waves (32 x 32 each), K slabs of one tap x 32
// channels, double-buffered LDS, loads of slab s+1 in flight while slab s is contracted, one barrier per slab; operand
// addressing (incremental 32-bit offsets), XCD-aware tile order, multi-map launches and the bias/residual/ReLU epilogue are
// those of conv.hip.
#include "conv_params.h"
#include "upsnet_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

#define CB_BM 64
#define CB_BN 64
#define CB_BK 32
#define CB_PITCH 40   // bf16 elements per LDS row (32 + 8 pad = 80 bytes)

__device__ static inline void cb_split4(const float4 v, const bool valid, bf16x4 &hi, bf16x4 &lo)
{
    const float x[4] = {valid ? v.x : 0.f, valid ? v.y : 0.f, valid ? v.z : 0.f, valid ? v.w : 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const __bf16 h = (__bf16)x[q];
        hi[q] = h;
        lo[q] = (__bf16)(x[q] - (float)h);
    }
}

template <int SPLIT>
__global__ void __launch_bounds__(256)
conv_bf16_kernel(const ConvParams p, const __bf16 *__restrict__ whi, const __bf16 *__restrict__ wlo)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ah[2][CB_BM][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bh[2][CB_BN][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Al[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? CB_BM : 1][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bl[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? CB_BN : 1][CB_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int akr = lane >> 5, aij = lane & 31;
    int m_t, n_t;
    {   // XCD-aware tile order (see conv.hip)
        const int bid = blockIdx.x, nt = p.n_tiles;
        const int per = (p.m_tiles + 7) >> 3;
        const int q = bid >> 3;
        n_t = q % nt;
        const int local = q / nt;
        m_t = (bid & 7) * per + local;
        if (local >= per || m_t >= p.m_tiles) return;
    }
    int si = 0;
#pragma unroll
    for (int q = 1; q < CV_MAXSEG; ++q) if (q < p.nseg && m_t >= p.seg[q].tile_start) si = q;
    const ConvSeg sg = p.seg[si];
    const long p0 = (long)(m_t - sg.tile_start) * CB_BM;
    const int n0 = n_t * CB_BN;
    const int ntap = p.KH * p.KW;
    const int nslabs = ntap * (p.Cin / CB_BK);

    // A staging: thread -> channels 4*ch4..+3 of pixels prow and prow + 32
    const int ch4 = tid & 7, prow = tid >> 3;
    int pix_n[2], pix_h[2], pix_w[2];
    const long HoWo = (long)sg.Ho * sg.Wo;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const long pp = p0 + prow + 32 * r;
        if (pp < sg.M) {
            const int n = (int)(pp / HoWo);
            const int rem = (int)(pp - (long)n * HoWo);
            pix_n[r] = n; pix_h[r] = (rem / sg.Wo) * p.stride - p.pad; pix_w[r] = (rem % sg.Wo) * p.stride - p.pad;
        } else { pix_n[r] = -1; pix_h[r] = 0; pix_w[r] = 0; }
    }
    const char *xbase = reinterpret_cast<const char *>(sg.x);
    // B staging: thread -> 8 consecutive k (one 16-byte octet) of column bcol
    const int bcol = tid >> 2, boct = tid & 3;
    const unsigned slab_bytes = (unsigned)p.ldw * CB_BK * 2u;
    unsigned ob = ((unsigned)(n0 + bcol) * CB_BK + 8u * boct) * 2u;   // byte offset inside the packed weights, advances one slab per fetch
    const char *whb = reinterpret_cast<const char *>(whi), *wlb = reinterpret_cast<const char *>(wlo);

    unsigned oa0 = 0, oa1 = 0;
    bool cv0 = false, cv1 = false;
    int f_cs = 0, f_ki = 0, f_kj = 0;
    bool f_newtap = true;
    float4 ra0, ra1;
    bool rv0 = false, rv1 = false;
    uint4 rbh, rbl;

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#define CB_TAP(R)                                                                                         \
    {                                                                                                     \
        const int hi = pix_h[R] + f_ki * p.dil, wi = pix_w[R] + f_kj * p.dil;                             \
        cv##R = pix_n[R] >= 0 && hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W;                            \
        const int hc = min(max(hi, 0), sg.H - 1), wc = min(max(wi, 0), sg.W - 1), nc = max(pix_n[R], 0); \
        oa##R = 4u * (unsigned)(((nc * sg.H + hc) * sg.W + wc) * p.Cin + 4 * ch4);                        \
    }
#define CB_FETCH                                                                                          \
    {                                                                                                     \
        if (f_newtap) { CB_TAP(0) CB_TAP(1) }                                                             \
        ra0 = *reinterpret_cast<const float4 *>(xbase + oa0); rv0 = cv0; oa0 += 4u * CB_BK;               \
        ra1 = *reinterpret_cast<const float4 *>(xbase + oa1); rv1 = cv1; oa1 += 4u * CB_BK;               \
        rbh = *reinterpret_cast<const uint4 *>(whb + ob);                                                 \
        if (SPLIT == 3) rbl = *reinterpret_cast<const uint4 *>(wlb + ob);                                 \
        ob += slab_bytes;                                                                                 \
        f_cs += CB_BK;                                                                                    \
        f_newtap = f_cs == p.Cin;                                                                         \
        if (f_newtap) { f_cs = 0; if (++f_kj == p.KW) { f_kj = 0; ++f_ki; } }                             \
    }
#define CB_STASH(BUF)                                                                                     \
    {                                                                                                     \
        bf16x4 h0, l0, h1, l1;                                                                            \
        cb_split4(ra0, rv0, h0, l0);                                                                      \
        cb_split4(ra1, rv1, h1, l1);                                                                      \
        *reinterpret_cast<bf16x4 *>(&Ah[BUF][prow][4 * ch4]) = h0;                                        \
        *reinterpret_cast<bf16x4 *>(&Ah[BUF][prow + 32][4 * ch4]) = h1;                                   \
        *reinterpret_cast<uint4 *>(&Bh[BUF][bcol][8 * boct]) = rbh;                                       \
        if (SPLIT == 3) {                                                                                 \
            *reinterpret_cast<bf16x4 *>(&Al[BUF][prow][4 * ch4]) = l0;                                    \
            *reinterpret_cast<bf16x4 *>(&Al[BUF][prow + 32][4 * ch4]) = l1;                               \
            *reinterpret_cast<uint4 *>(&Bl[BUF][bcol][8 * boct]) = rbl;                                   \
        }                                                                                                 \
    }

    CB_FETCH
    CB_STASH(0)
    __syncthreads();
    for (int s = 0; s < nslabs; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nslabs;
        if (more) CB_FETCH
#pragma unroll
        for (int t = 0; t < CB_BK / 16; ++t) {
            const int ko = (2 * t + akr) * 8;
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&Ah[buf][wm * 32 + aij][ko]);
            const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(&Bh[buf][wn * 32 + aij][ko]);
            if (SPLIT == 3) {   // small terms first
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&Al[buf][wm * 32 + aij][ko]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(&Bl[buf][wn * 32 + aij][ko]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        if (more) CB_STASH(buf ^ 1)
        __syncthreads();
    }
#undef CB_TAP
#undef CB_FETCH
#undef CB_STASH

    // ---- epilogue: + bias, + residual, ReLU (as conv.hip)
    const bool has_res = sg.res != nullptr, has_bias = p.bias != nullptr;
    const int co = n0 + wn * 32 + aij;
    const bool co_ok = co < p.Cout;
    const int coc = co_ok ? co : 0;
    const float bv = has_bias ? p.bias[coc] : 0.f;
    const long pbase = p0 + wm * 32 + 4 * akr;
    float rr[16];
    if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long pp = pbase + (r & 3) + 8 * (r >> 2);
            pp = pp < sg.M ? pp : sg.M - 1;
            rr[r] = sg.res[pp * p.Cout + coc];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long pp = pbase + (r & 3) + 8 * (r >> 2);
        float v = acc[r];
        if (has_bias) v = v + bv;
        if (has_res) v = v + rr[r];
        if (p.relu) v = fmaxf(v, 0.f);
        if (co_ok && pp < sg.M) sg.out[pp * p.Cout + co] = v;
    }
}

extern "C" int upsnet_conv2d_nhwc_bf16(void *stream, int nseg, const float *const x[], const float *const residual[], float *const out[],
                                       const int batch[], const int height[], const int width[], int Cin, const void *wpack_hi,
                                       const void *wpack_lo, int ldw, const float *bias, int Cout, int KH, int KW, int stride, int pad,
                                       int relu)
{
    ConvParams p;
    int rc = conv_fill(p, "conv2d_nhwc_bf16", nseg, x, residual, nullptr, nullptr, out, batch, height, width, Cin, Cout,
                       reinterpret_cast<const float *>(wpack_hi), ldw, bias, KH, KW, stride, pad, 1, relu);
    if (rc) return rc;
    UPS_REQUIRE(ldw % CB_BN == 0, "conv2d_nhwc_bf16: ldw must be a multiple of %d (got %d)", CB_BN, ldw);
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) { p.seg[i].tile_start = tiles; tiles += (int)((p.seg[i].M + CB_BM - 1) / CB_BM); }
    p.m_tiles = tiles;
    p.n_tiles = ldw / CB_BN;
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;
    const __bf16 *hi = reinterpret_cast<const __bf16 *>(wpack_hi), *lo = reinterpret_cast<const __bf16 *>(wpack_lo);
    if (lo) hipLaunchKernelGGL(conv_bf16_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, hi, lo);
    else hipLaunchKernelGGL(conv_bf16_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, hi, lo);
    UPS_CHECK_LAUNCH("conv_bf16_kernel");
    return 0;
}

// weight [Cout, Cin, kh, kw] fp32 -> hi / lo bf16 in [slab = tap * Cin/32 + c/32][column (ldw)][32 k], zero padded columns
__global__ void conv_pack_weight_bf16_kernel(const float *__restrict__ w, int cout, int cin, int taps, int ldw, __bf16 *__restrict__ hi,
                                             __bf16 *__restrict__ lo)
{
    const long total = (long)taps * cin * ldw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int kk = idx % CB_BK;
        const int col = (idx / CB_BK) % ldw;
        const long slab = idx / ((long)CB_BK * ldw);
        const int cslabs = cin / CB_BK;
        const int tap = (int)(slab / cslabs), c = (int)(slab % cslabs) * CB_BK + kk;
        const float v = col < cout ? w[((long)col * cin + c) * taps + tap] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[idx] = h;
        if (lo) lo[idx] = (__bf16)(v - (float)h);
    }
}

extern "C" int upsnet_conv_pack_weight_bf16(void *stream, const float *weight, int cout, int cin, int kh, int kw, int ldw, void *wpack_hi,
                                            void *wpack_lo)
{
    UPS_REQUIRE(weight && wpack_hi && cout > 0 && cin > 0 && cin % CB_BK == 0 && kh > 0 && kw > 0 && ldw >= cout && ldw % CB_BN == 0,
                "conv_pack_weight_bf16: bad args (Cin %% 32 == 0, ldw %% 64 == 0)");
    const long total = (long)kh * kw * cin * ldw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv_pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, kh * kw, ldw,
                       reinterpret_cast<__bf16 *>(wpack_hi), reinterpret_cast<__bf16 *>(wpack_lo));
    UPS_CHECK_LAUNCH("conv_pack_weight_bf16_kernel");
    return 0;
}
