// conv.hip -- dense AND deformable convolution as one fp32 implicit-GEMM kernel family on the MFMA units of gfx950.
//
// Reference (dense): every nn.Conv2d of the backbone / FPN / RPN / heads (upsnet/models/resnet.py:53-100,
// fpn.py:78-104, rpn.py:52-57, rcnn.py:79-87, fcn.py:88-108) followed by separate frozen-BN, bias, ReLU and
// residual-add passes. Reference (deformable): DeformConvFunction.forward = deformable_im2col into a column
// buffer in HBM + torch.mm + bias (upsnet/operators/functions/deform_conv.py:43-57, deform_conv_kernel.cu:194-242;
// v2: mod_deform_conv_kernel.cu:187-249).
//
// Here: NHWC activations, weights pre-packed once to [kh*kw*Cin, ldw] (tap-major rows, ldw = Cout rounded up to 32,
// zero padded). A workgroup owns 128 output pixels x BN output channels. K is walked in slabs of one tap x 32 input
// channels. While slab s is contracted from LDS with v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation in a fixed k order), the global loads of slab s+1 are already in flight into registers
// (double-buffered LDS, one barrier per slab):
//   dense      : one float4 per (pixel, 4 channels), zero-filled outside the image;
//   deformable : the four bilinear corners as float4 channel runs; offsets -> corner addresses + weights are
//                computed once per tap in registers with the reference's exact fp32 arithmetic; the blend (and the
//                v2 modulation) happens when the registers are written to LDS. No column buffer exists.
// Epilogue fused: + bias (the folded frozen-BN shift), + residual, ReLU, one store.
// Up to 5 feature maps that share the same weights (FPN levels: RPN head, FCN-head subnet) go in ONE launch.
#include "common.h"
#include "upsnet_hip.h"

#define CV_BM 128
#define CV_BK 32
#define CV_LDA (CV_BM + 1)
#define CV_MAXSEG 5

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvSeg {
    const float *x, *res, *off, *mask;
    float *out;
    int N, H, W, Ho, Wo;
    int tile_start;
    long M;  // N*Ho*Wo
};

struct ConvParams {
    ConvSeg seg[CV_MAXSEG];
    const float *w, *bias;
    int nseg, Cin, Cout, ldw, KH, KW, stride, pad, dil, relu;
    int m_tiles, n_tiles;
    int kord;  // K walk: 0 = tap outer / channel-slab inner (default), 1 = channel-slab outer / tap inner
};

// corner descriptor of one (pixel, tap): element offsets of the 4 corners (clamped, always loadable),
// validity bits, blend weights (deform_conv_kernel.cu:88-118 arithmetic), v2 modulation
struct DcnDesc {
    int o1, o2, o3, o4;
    float w1, w2, w3, w4, m;
    unsigned vb;
};

__device__ static inline DcnDesc dcn_desc(const ConvSeg &sg, const long pp, const int tap, const int ntap, const int h_base,
                                          const int w_base, const int cin, const bool mod)
{
    DcnDesc d;
    d.o1 = d.o2 = d.o3 = d.o4 = 0;
    d.w1 = d.w2 = d.w3 = d.w4 = 0.f;
    d.m = 1.f;
    d.vb = 0;
    if (pp < 0) return d;
    const float off_h = sg.off[pp * (2 * ntap) + 2 * tap];
    const float off_w = sg.off[pp * (2 * ntap) + 2 * tap + 1];
    const float h_im = (float)h_base + off_h;   // integer part converted to float before the add (:227-228)
    const float w_im = (float)w_base + off_w;
    const int H = sg.H, W = sg.W;
    if (h_im > -1 && w_im > -1 && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.0f - lh, hw = 1.0f - lw;
        d.w1 = hh * hw; d.w2 = hh * lw; d.w3 = lh * hw; d.w4 = lh * lw;
        const bool a = h_low >= 0, b = h_high <= H - 1, c = w_low >= 0, e = w_high <= W - 1;
        const int hl = a ? h_low : 0, hhi = b ? h_high : H - 1, wl = c ? w_low : 0, whi = e ? w_high : W - 1;
        d.o1 = (hl * W + wl) * cin; d.o2 = (hl * W + whi) * cin; d.o3 = (hhi * W + wl) * cin; d.o4 = (hhi * W + whi) * cin;
        d.vb = (a && c ? 1u : 0u) | (a && e ? 2u : 0u) | (b && c ? 4u : 0u) | (b && e ? 8u : 0u);
    }
    if (mod) d.m = sg.mask[pp * ntap + tap];
    return d;
}

__device__ static inline float dcn_blend1(const DcnDesc &d, float v1, float v2, float v3, float v4, const bool mod)
{
    v1 = (d.vb & 1u) ? v1 : 0.f;
    v2 = (d.vb & 2u) ? v2 : 0.f;
    v3 = (d.vb & 4u) ? v3 : 0.f;
    v4 = (d.vb & 8u) ? v4 : 0.f;
    float val = d.w1 * v1;
    val = val + d.w2 * v2;
    val = val + d.w3 * v3;
    val = val + d.w4 * v4;
    if (mod) val = val * d.m;
    return val;
}

// WM x WN = 32x32 tiles per wave, waves arranged WAVES_M x WAVES_N (4 waves): BM = 32*WM*WAVES_M (128 or 64) output
// pixels x BN = 32*WN*WAVES_N (128 / 64 / 32) output channels per workgroup.
// DEFORM: 0 dense, 1 deformable v1, 2 deformable v2 (modulated). PIPE: pin the k-loop software pipeline.
template <int WM, int WN, int WAVES_M, int WAVES_N, int DEFORM, int PIPE, int BK>
__global__ void __launch_bounds__(256)
conv_igemm_f32_kernel(const ConvParams p)
{
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int LDA = BM + 1;
    constexpr int CH4 = BK / 4;        // float4 channel groups per slab row (8 or 16)
    constexpr int RP = 256 / CH4;      // pixels staged per pass of the 256 threads (32 or 16)
    constexpr int PXT = BM / RP;       // pixels staged per thread per slab (2 or 4)
    static_assert((BM == 128 || BM == 64) && WAVES_M * WAVES_N == 4, "tile");
    static_assert(BN == 32 || BN == 64 || BN == 128, "BN");
    static_assert((BK == 32 || BK == 64) && PXT <= 4, "BK");
    constexpr int B_F4 = (BK * BN / 4) / 256;  // float4 per thread for the B slab (1, 2 or 4)
    static_assert(B_F4 >= 1 && B_F4 <= 4, "B staging");
    constexpr bool MOD = DEFORM == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *As = reinterpret_cast<float *>(smem_raw);     // [2][BK][LDA]
    float *Bs = As + 2 * BK * LDA;                       // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int akr = lane >> 5, aij = lane & 31;
    // XCD-aware tile order. Workgroup b is observed to run on XCD b % 8 (speed only, never correctness). Each XCD gets
    // a CONTIGUOUS range of m-tiles (so the 3x3 / bilinear halos of vertically adjacent tiles hit the same 4 MiB L2)
    // and all n-tiles of an m-tile (they share the A panel) stay on that XCD.
    int m_t, n_t;
    {
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
    const long p0 = (long)(m_t - sg.tile_start) * BM;
    const int n0 = n_t * BN;
    const int ntap = p.KH * p.KW;

    // ---- per-thread A staging geometry: PXT pixels (prow + RP r), channels 4*ch4..+3 of the slab
    const int ch4 = tid % CH4, prow = tid / CH4;
    int pix_n[4], pix_h[4], pix_w[4];
    long pix_p[4];
    const long HoWo = (long)sg.Ho * sg.Wo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long pp = p0 + prow + RP * r;
        if (r < PXT && pp < sg.M) {
            const int n = (int)(pp / HoWo);
            const int rem = (int)(pp - (long)n * HoWo);
            pix_n[r] = n; pix_h[r] = (rem / sg.Wo) * p.stride - p.pad; pix_w[r] = (rem % sg.Wo) * p.stride - p.pad;
            pix_p[r] = pp;
        } else { pix_n[r] = -1; pix_h[r] = 0; pix_w[r] = 0; pix_p[r] = -1; }
    }
    const int cin_slabs = p.Cin / BK;
    const int nslabs = ntap * cin_slabs;

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging registers (named scalars: arrays captured by lambdas / indexed in loops were demoted to scratch).
    // Loads are UNCONDITIONAL (clamped addresses); zero-fill / validity is applied when the registers are written
    // to LDS -- a "load or 0" select at the load site makes hipcc branch around every load and wait vmcnt(0) each.
    float4 a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33;  // [pixel r][corner]
    float4 rb0, rb1, rb2, rb3;
    bool rv0 = false, rv1 = false, rv2 = false, rv3 = false;
    DcnDesc d0, d1, d2, d3;
    int desc_tap = -1;
    int f_tap = 0, f_cs = 0, f_ki = 0, f_kj = 0;  // (tap, channel slab) of the NEXT slab to fetch

#define CV_LOAD_B(Q) \
    (*reinterpret_cast<const float4 *>(wrow + (long)((tid + 256 * (Q)) / (BN / 4)) * p.ldw + 4 * ((tid + 256 * (Q)) % (BN / 4))))
#define CV_FETCH_DENSE(R, A0, RV)                                                                                     \
    {                                                                                                                 \
        const int hi = pix_h[R] + ki * p.dil, wi = pix_w[R] + kj * p.dil;                                             \
        RV = pix_n[R] >= 0 && hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W;                                           \
        const int hc = min(max(hi, 0), sg.H - 1), wc = min(max(wi, 0), sg.W - 1), nc = max(pix_n[R], 0);             \
        A0 = *reinterpret_cast<const float4 *>(sg.x + (((long)nc * sg.H + hc) * sg.W + wc) * p.Cin + cs + 4 * ch4);   \
    }
#define CV_FETCH_DEFORM(R, D, A0, A1, A2, A3)                                                                         \
    {                                                                                                                 \
        const float *xb = sg.x + (long)max(pix_n[R], 0) * sg.H * sg.W * p.Cin + cs + 4 * ch4;                         \
        A0 = *reinterpret_cast<const float4 *>(xb + D.o1);                                                            \
        A1 = *reinterpret_cast<const float4 *>(xb + D.o2);                                                            \
        A2 = *reinterpret_cast<const float4 *>(xb + D.o3);                                                            \
        A3 = *reinterpret_cast<const float4 *>(xb + D.o4);                                                            \
    }
#define CV_FETCH(S)                                                                                                   \
    {                                                                                                                 \
        const int tap = f_tap, cs = f_cs, ki = f_ki, kj = f_kj;                                                       \
        if (p.kord) {                                                                                                 \
            if (++f_tap == ntap) { f_tap = 0; f_ki = 0; f_kj = 0; f_cs += BK; }                                       \
            else if (++f_kj == p.KW) { f_kj = 0; ++f_ki; }                                                            \
        } else {                                                                                                      \
            f_cs += BK;                                                                                               \
            if (f_cs == p.Cin) { f_cs = 0; ++f_tap; if (++f_kj == p.KW) { f_kj = 0; ++f_ki; } }                       \
        }                                                                                                             \
        if (DEFORM) {                                                                                                 \
            if (tap != desc_tap) {                                                                                    \
                desc_tap = tap;                                                                                       \
                d0 = dcn_desc(sg, pix_p[0], tap, ntap, pix_h[0] + ki * p.dil, pix_w[0] + kj * p.dil, p.Cin, MOD);     \
                d1 = dcn_desc(sg, pix_p[1], tap, ntap, pix_h[1] + ki * p.dil, pix_w[1] + kj * p.dil, p.Cin, MOD);     \
                if (PXT > 2) {                                                                                        \
                d2 = dcn_desc(sg, pix_p[2], tap, ntap, pix_h[2] + ki * p.dil, pix_w[2] + kj * p.dil, p.Cin, MOD);     \
                d3 = dcn_desc(sg, pix_p[3], tap, ntap, pix_h[3] + ki * p.dil, pix_w[3] + kj * p.dil, p.Cin, MOD);     \
                }                                                                                                     \
            }                                                                                                         \
            CV_FETCH_DEFORM(0, d0, a00, a01, a02, a03) CV_FETCH_DEFORM(1, d1, a10, a11, a12, a13)                     \
            if (PXT > 2) { CV_FETCH_DEFORM(2, d2, a20, a21, a22, a23) CV_FETCH_DEFORM(3, d3, a30, a31, a32, a33) }    \
        } else {                                                                                                      \
            CV_FETCH_DENSE(0, a00, rv0) CV_FETCH_DENSE(1, a10, rv1)                                                   \
            if (PXT > 2) { CV_FETCH_DENSE(2, a20, rv2) CV_FETCH_DENSE(3, a30, rv3) }                                  \
        }                                                                                                             \
        const float *wrow = p.w + ((long)tap * p.Cin + cs) * p.ldw + n0;                                              \
        rb0 = CV_LOAD_B(0);                                                                                           \
        if (B_F4 > 1) rb1 = CV_LOAD_B(1);                                                                             \
        if (B_F4 > 2) { rb2 = CV_LOAD_B(2); rb3 = CV_LOAD_B(3); }                                                     \
    }
#define CV_STASH_PX(R, VX, VY, VZ, VW)                                                                                \
    {                                                                                                                 \
        const int px = prow + RP * R;                                                                                 \
        sa[(4 * ch4 + 0) * LDA + px] = VX;                                                                         \
        sa[(4 * ch4 + 1) * LDA + px] = VY;                                                                         \
        sa[(4 * ch4 + 2) * LDA + px] = VZ;                                                                         \
        sa[(4 * ch4 + 3) * LDA + px] = VW;                                                                         \
    }
#define CV_STASH_DENSE(R, A0, RV) CV_STASH_PX(R, RV ? A0.x : 0.f, RV ? A0.y : 0.f, RV ? A0.z : 0.f, RV ? A0.w : 0.f)
#define CV_STASH_DEFORM(R, D, A0, A1, A2, A3)                                                                         \
    CV_STASH_PX(R, dcn_blend1(D, A0.x, A1.x, A2.x, A3.x, MOD), dcn_blend1(D, A0.y, A1.y, A2.y, A3.y, MOD),            \
                dcn_blend1(D, A0.z, A1.z, A2.z, A3.z, MOD), dcn_blend1(D, A0.w, A1.w, A2.w, A3.w, MOD))
#define CV_STASH(BUF)                                                                                                 \
    {                                                                                                                 \
        float *sa = As + (BUF) * BK * LDA;                                                                            \
        if (DEFORM) {                                                                                                 \
            CV_STASH_DEFORM(0, d0, a00, a01, a02, a03) CV_STASH_DEFORM(1, d1, a10, a11, a12, a13)                     \
            if (PXT > 2) { CV_STASH_DEFORM(2, d2, a20, a21, a22, a23) CV_STASH_DEFORM(3, d3, a30, a31, a32, a33) }    \
        } else {                                                                                                      \
            CV_STASH_DENSE(0, a00, rv0) CV_STASH_DENSE(1, a10, rv1)                                                   \
            if (PXT > 2) { CV_STASH_DENSE(2, a20, rv2) CV_STASH_DENSE(3, a30, rv3) }                                  \
        }                                                                                                             \
        float4 *sb = reinterpret_cast<float4 *>(Bs + (BUF) * BK * BN);                                                \
        sb[tid] = rb0;                                                                                                \
        if (B_F4 > 1) sb[tid + 256] = rb1;                                                                            \
        if (B_F4 > 2) { sb[tid + 512] = rb2; sb[tid + 768] = rb3; }                                                   \
    }

    CV_FETCH(0)
    CV_STASH(0)
    __syncthreads();
    for (int s = 0; s < nslabs; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < nslabs;
        // Deformable: CV_STASH blends with d0..d3, which CV_FETCH refreshes at tap boundaries BEFORE issuing that
        // slab's loads; fetch(s+1) and stash(s+1) always see the same descriptors, slab s was stashed earlier.
        if (more) CV_FETCH(s + 1)
        const float *a = As + buf * BK * LDA + wm * (WM * 32) + aij;
        const float *b = Bs + buf * BK * BN + wn * (WN * 32) + aij;
        // software-pipelined k loop: the LDS fragment reads of step k+1 are issued before the MFMAs of step k, and
        // the next slab's registers are written to the other LDS buffer under the last quarter of the MFMAs
        float av[2][WM], bv[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) av[0][i] = a[akr * LDA + 32 * i];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[0][j] = b[akr * BN + 32 * j];
#pragma unroll
        for (int k = 0; k < BK / 2; ++k) {
            const int cur = k & 1, nxt = cur ^ 1;
            if (k + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < WM; ++i) av[nxt][i] = a[(2 * (k + 1) + akr) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < WN; ++j) bv[nxt][j] = b[(2 * (k + 1) + akr) * BN + 32 * j];
            }
            if (PIPE && k == 3 * (BK / 2) / 4 && more) CV_STASH(buf ^ 1)
            if (PIPE) __builtin_amdgcn_sched_barrier(0);  // keep the k+1 fragment reads (and the stash) ahead of step k's MFMAs
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
        if (!PIPE && more) CV_STASH(buf ^ 1)
        __syncthreads();
    }
#undef CV_LOAD_B
#undef CV_FETCH_DENSE
#undef CV_FETCH_DEFORM
#undef CV_FETCH
#undef CV_STASH_PX
#undef CV_STASH_DENSE
#undef CV_STASH_DEFORM
#undef CV_STASH

    // ---- fused epilogue: + bias, + residual, ReLU. Residual values are loaded 16 at a time, unconditionally
    // (clamped row), before any of them is used, so the loads overlap instead of serialising.
    const bool has_res = sg.res != nullptr, has_bias = p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int co = n0 + wn * (WN * 32) + 32 * j + aij;
        const bool co_ok = co < p.Cout;
        const int coc = co_ok ? co : 0;
        const float bv = has_bias ? p.bias[coc] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const long pbase = p0 + wm * (WM * 32) + 32 * i + 4 * akr;
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
                float v = acc[i][j][r];
                if (has_bias) v = v + bv;
                if (has_res) v = v + rr[r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (co_ok && pp < sg.M) sg.out[pp * p.Cout + co] = v;
            }
        }
    }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int DEFORM, int PIPE, int BK = CV_BK>
static int conv_launch(hipStream_t st, ConvParams &p)
{
    constexpr int BN = WAVES_N * WN * 32, BM = WAVES_M * WM * 32;
    // re-tile the feature maps for this BM
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) { p.seg[i].tile_start = tiles; tiles += (int)((p.seg[i].M + BM - 1) / BM); }
    p.m_tiles = tiles;
    p.n_tiles = (p.Cout + BN - 1) / BN;
    const size_t smem = (size_t)(2 * BK * (BM + 1) + 2 * BK * BN) * sizeof(float);
    static bool attr_set = false;  // > 64 KiB of dynamic LDS must be opted into once per kernel
    if (!attr_set && smem > 64 * 1024) {
        UPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, DEFORM, PIPE, BK>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;  // see the XCD-aware tile order in the kernel
    hipLaunchKernelGGL((conv_igemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, DEFORM, PIPE, BK>), dim3(grid), dim3(256), smem, st, p);
    UPS_CHECK_LAUNCH("conv_igemm_f32_kernel");
    return 0;
}

// Tile choice (measured on MI355X over every conv shape of UPSNet-50 @1024x2048, tools/sweep_conv_tiles.py): on this chip
// the fp32 MFMA kernel is occupancy-bound -- the 64-pixel tiles (88-150 registers, 33-50 KiB LDS -> 3-5 waves per SIMD)
// beat the 128x128 tile (208 registers, 66 KiB -> 2 waves) on almost every layer. 64x128 wins for the large 3x3 layers
// and for the deformable variant (A operand = expensive gather, computed once per 128 output channels); 64x64 elsewhere.
// upsnet_conv_tuning(pipe, force_tile) overrides for A/B runs.
static int g_pipe = -1, g_force_tile = 0, g_kord = 0;
extern "C" void upsnet_conv_tuning(int pipe, int force_tile)
{
    g_pipe = pipe < 0 ? -1 : (pipe & 1);
    g_kord = pipe < 0 ? 0 : ((pipe & 2) ? 1 : 0);  // bit 1 of `pipe` selects the slab-outer K walk (A/B runs only)
    g_force_tile = force_tile;
}

template <int DEFORM, int PIPE>
static int conv_dispatch2(hipStream_t st, ConvParams &p)
{
    long M = 0;
    for (int i = 0; i < p.nseg; ++i) M += p.seg[i].M;
    const bool n128 = p.ldw % 128 == 0, n64 = p.ldw % 64 == 0;
    int tile = g_force_tile;
    if (!tile) {
        if (DEFORM) tile = n128 ? 4 : (n64 ? 5 : 3);
        else if (n128 && p.KH * p.KW > 1 && M >= 32768) tile = 4;
        else tile = n64 ? 5 : 3;
    }
    if (tile == 1 && !n128) tile = n64 ? 2 : 3;
    if ((tile == 2 || tile == 5) && !n64) tile = 3;
    if (tile == 4 && !n128) tile = n64 ? 5 : 3;
    if (tile == 6 && (!n64 || p.Cin % 64 != 0)) tile = n64 ? 5 : 3;
    switch (tile) {
    case 1: return conv_launch<2, 2, 2, 2, DEFORM, PIPE>(st, p);   // 128 x 128
    case 2: return conv_launch<1, 2, 4, 1, DEFORM, PIPE>(st, p);   // 128 x 64
    case 4: return conv_launch<1, 2, 2, 2, DEFORM, PIPE>(st, p);   // 64 x 128
    case 5: return conv_launch<1, 1, 2, 2, DEFORM, PIPE>(st, p);   // 64 x 64
    case 6: return conv_launch<1, 1, 2, 2, DEFORM, PIPE, 64>(st, p);  // 64 x 64, 64-channel K slabs
    default: return conv_launch<1, 1, 4, 1, DEFORM, PIPE>(st, p);  // 128 x 32
    }
}

template <int DEFORM>
static int conv_dispatch(hipStream_t st, ConvParams &p)
{
    const int pipe = g_pipe >= 0 ? g_pipe : 1;
    p.kord = g_kord;
    return pipe ? conv_dispatch2<DEFORM, 1>(st, p) : conv_dispatch2<DEFORM, 0>(st, p);
}

static int conv_fill(ConvParams &p, const char *who, int nseg, const float *const x[], const float *const res[],
                     const float *const off[], const float *const mask[], float *const out[], const int batch[],
                     const int height[], const int width[], int Cin, int Cout, const float *wpack, int ldw, const float *bias,
                     int KH, int KW, int stride, int pad, int dil, int relu)
{
    UPS_REQUIRE(nseg >= 1 && nseg <= CV_MAXSEG, "%s: 1..%d feature maps per launch (got %d)", who, CV_MAXSEG, nseg);
    UPS_REQUIRE(x && out && height && width && wpack, "%s: null pointer", who);
    UPS_REQUIRE(Cin > 0 && Cin % CV_BK == 0, "%s: Cin must be a multiple of 32 (got %d)", who, Cin);
    UPS_REQUIRE(Cout > 0 && ldw % 32 == 0 && ldw >= Cout, "%s: ldw must be Cout rounded up to 32 (got %d for Cout=%d)", who, ldw, Cout);
    UPS_REQUIRE(KH >= 1 && KW >= 1 && KH * KW <= 49 && stride >= 1 && pad >= 0 && dil >= 1, "%s: bad kernel/stride/pad/dilation", who);
    p.w = wpack; p.bias = bias; p.nseg = nseg; p.Cin = Cin; p.Cout = Cout; p.ldw = ldw; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad = pad; p.dil = dil; p.relu = relu;
    int tiles = 0;
    for (int i = 0; i < CV_MAXSEG; ++i) {
        ConvSeg &s = p.seg[i];
        if (i < nseg) {
            const int nb = batch ? batch[i] : 1;
            UPS_REQUIRE(x[i] && out[i] && nb > 0 && height[i] > 0 && width[i] > 0, "%s: bad feature map %d", who, i);
            s.x = x[i]; s.out = out[i]; s.res = res ? res[i] : nullptr; s.off = off ? off[i] : nullptr; s.mask = mask ? mask[i] : nullptr;
            s.N = nb; s.H = height[i]; s.W = width[i];
            s.Ho = (height[i] + 2 * pad - (dil * (KH - 1) + 1)) / stride + 1;
            s.Wo = (width[i] + 2 * pad - (dil * (KW - 1) + 1)) / stride + 1;
            UPS_REQUIRE(s.Ho > 0 && s.Wo > 0, "%s: empty output for feature map %d", who, i);
            UPS_REQUIRE((long)height[i] * width[i] * Cin < 2147483647L, "%s: feature map %d too large for 32-bit offsets", who, i);
            s.M = (long)nb * s.Ho * s.Wo;
            s.tile_start = tiles;
            tiles += (int)((s.M + 127) / 128);
        } else {
            s.x = s.res = s.off = s.mask = nullptr; s.out = nullptr;
            s.N = s.H = s.W = s.Ho = s.Wo = 0; s.M = 0; s.tile_start = 0x7fffffff;
        }
    }
    p.m_tiles = tiles;
    return 0;
}

extern "C" int upsnet_conv2d_nhwc_f32(void *stream, int nseg, const float *const x[], const float *const residual[],
                                      float *const out[], const int batch[], const int height[], const int width[], int Cin,
                                      const float *wpack, int ldw, const float *bias, int Cout, int KH, int KW, int stride,
                                      int pad, int relu)
{
    ConvParams p;
    int rc = conv_fill(p, "conv2d_nhwc_f32", nseg, x, residual, nullptr, nullptr, out, batch, height, width, Cin, Cout, wpack, ldw,
                       bias, KH, KW, stride, pad, 1, relu);
    if (rc) return rc;
    return conv_dispatch<0>((hipStream_t)stream, p);
}

extern "C" int upsnet_deform_conv_forward_nhwc(void *stream, int nlev, const float *const x[], const float *const offset[],
                                               const float *const mask[], float *const out[], const int height[],
                                               const int width[], int cin, int cout, int kh, int kw, int pad_h, int pad_w,
                                               int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                                               const float *wpack, int ldw, const float *bias, int relu)
{
    UPS_REQUIRE(deformable_group == 1, "deform_conv_forward_nhwc: deformable_group=%d not supported by the fused kernel (use the im2col path)", deformable_group);
    UPS_REQUIRE(pad_h == pad_w && stride_h == stride_w && dil_h == dil_w, "deform_conv_forward_nhwc: square stride/pad/dilation only");
    UPS_REQUIRE(nlev >= 1 && nlev <= 4 && offset, "deform_conv_forward_nhwc: nlev must be 1..4 and offsets given");
    for (int l = 0; l < nlev; ++l) UPS_REQUIRE(offset[l] && (!mask || mask[l]), "deform_conv_forward_nhwc: null offset/mask at level %d", l);
    ConvParams p;
    int rc = conv_fill(p, "deform_conv_forward_nhwc", nlev, x, nullptr, offset, mask, out, nullptr, height, width, cin, cout, wpack,
                       ldw, bias, kh, kw, stride_h, pad_h, dil_h, relu);
    if (rc) return rc;
    return mask ? conv_dispatch<2>((hipStream_t)stream, p) : conv_dispatch<1>((hipStream_t)stream, p);
}

// weight [Cout, Cin, kh, kw] -> wpack [(tap*Cin + c), ldw], zero padded columns
__global__ void conv_pack_weight_kernel(const float *__restrict__ w, int cout, int cin, int taps, int ldw, float *__restrict__ wp)
{
    const long total = (long)ldw * cin * taps;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int co = idx % ldw;
        const int c = (idx / ldw) % cin;
        const int tap = idx / ((long)ldw * cin);
        wp[idx] = co < cout ? w[((long)co * cin + c) * taps + tap] : 0.f;
    }
}

extern "C" int upsnet_conv_pack_weight(void *stream, const float *weight, int cout, int cin, int kh, int kw, int ldw, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cin > 0 && kh > 0 && kw > 0 && ldw >= cout, "conv_pack_weight: bad args");
    const long total = (long)ldw * cin * kh * kw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, kh * kw, ldw, wpack);
    UPS_CHECK_LAUNCH("conv_pack_weight_kernel");
    return 0;
}
