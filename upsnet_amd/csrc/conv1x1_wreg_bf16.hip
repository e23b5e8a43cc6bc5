// conv1x1_wreg_bf16.hip -- 1x1 convolution of bf16 activations on the bf16 matrix cores with BOTH MFMA operands loaded from global
// memory in fragment order: no LDS, no barrier (gfx950). The 1x1 layers the bf16 mode of BASELINE.json configs[2] runs outside the
// fused identity blocks: conv1 / conv3 / the projection of the first bottleneck of every stage (upsnet/models/resnet.py:84-100,
// stride 1 or 2) and the FPN laterals with the top-down add (upsnet/models/fpn.py:34,90-96).
//
// These layers are HBM-bound by an order of magnitude (res2 conv3 + shortcut: 150 MB = 19 us at 8 TB/s against 2 us of matrix-pipe
// time) and ran at 20-25 % of that on conv_bf16_kernel (93 us): two K slabs between a prologue and an epilogue of 2-byte stores
// (lane = channel). Here
//   * A = weights (rows = 32 output channels): 16 bytes per lane straight from the packed array [k / 32][column][32 k]
//     (upsnet_conv_pack_weight_bf16); B = activations (columns = 32 pixels): lane (pixel l, k half h) loads the 16 bytes
//     x[pixel][k0 + 8 h .. + 7] of ITS pixel -- any pixel stride, so the stride-2 layers cost nothing extra. The four waves of a
//     workgroup take different channel groups of the same pixels: the activation lines are fetched from L2 once and re-read from L1;
//   * every wave runs its own K walk, D k-steps of both operands in flight (32 KB per wave), no synchronisation at all;
//   * the accumulators (lane = pixel) pass through a wave-private LDS tile once, after the K walk, so that the shortcut is read and
//     the result written in whole 128-byte lines (16 bytes per lane, the lanes of a pixel side by side).
// Same products and K order as conv_bf16_kernel.
#include <stdlib.h>

#include "conv_params.h"
#include "upsnet_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 c1_bf16x8;
typedef unsigned c1_uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned c1_uintx2 __attribute__((ext_vector_type(2)));

#define C1_EPP 272        // bytes per pixel of the epilogue tile: 64 fp32 + 16 (conflict-free 16-byte rows)

__device__ static inline __amdgpu_buffer_rsrc_t c1_rsrc(const void *ptr, const unsigned bytes)
{
    const size_t a = reinterpret_cast<size_t>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ static inline c1_bf16x8 c1_as_bf16x8(const c1_uintx4 v)
{
    c1_bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

__device__ static inline unsigned c1_pack2(const float a, const float b)
{
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    unsigned short ux, uy;
    __builtin_memcpy(&ux, &x, 2);
    __builtin_memcpy(&uy, &y, 2);
    return (unsigned)ux | ((unsigned)uy << 16);
}

// NWC: waves along the output channels (64 each): 4 -> 256 channels x 64 pixels per workgroup, 2 -> 128 channels x 128 pixels
template <int NWC, int OUT16>
__global__ void __launch_bounds__(256, 2) conv1x1_wreg_bf16_kernel(const ConvParams p, const char *__restrict__ wpk)
{
    constexpr int D = 8;                          // k-steps (16 channels) of both operands in flight
    constexpr int TP = 64 * (4 / NWC);            // pixels per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, lhalf = lane >> 5;
    const int wc = wave % NWC, wp = wave / NWC;
    int m_t, n_t;
    {   // XCD-aware tile order (block b runs on XCD b % 8): consecutive pixel tiles of a map on one XCD
        const int bid = blockIdx.x;
        const int per = (p.m_tiles + 7) >> 3;
        const int q = bid >> 3;
        n_t = q % p.n_tiles;
        m_t = (bid & 7) * per + q / p.n_tiles;
        if (m_t >= p.m_tiles) return;
    }
    int si = 0;
#pragma unroll
    for (int q = 1; q < CV_MAXSEG; ++q) if (q < p.nseg && m_t >= p.seg[q].tile_start) si = q;
    const ConvSeg sg = p.seg[si];
    const long m0 = (long)(m_t - sg.tile_start) * TP;
    const int hw = sg.Ho * sg.Wo;
    const __amdgpu_buffer_rsrc_t xrsrc = c1_rsrc(sg.x, (unsigned)(sg.N * sg.H * sg.W) * 2u * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t wrsrc = c1_rsrc(wpk, (unsigned)p.Cin * (unsigned)p.ldw * 2u);

    // this lane's two pixels: input byte offset (k half included), output / shortcut element offsets
    unsigned xoff[2], ooff[2], roff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long m = m0 + (wp * 2 + j) * 32 + l32;
        xoff[j] = 0x80000000u; ooff[j] = 0x20000000u; roff[j] = 0x20000000u;
        if (m < sg.M) {
            const int n = (int)(m / hw), r = (int)(m - (long)n * hw);
            const int oy = r / sg.Wo, ox = r - oy * sg.Wo;
            xoff[j] = (unsigned)((n * sg.H + oy * p.stride) * sg.W + ox * p.stride) * (unsigned)p.Cin * 2u + 16u * (unsigned)lhalf;
            ooff[j] = (unsigned)m * (unsigned)p.Cout;
            if (p.res_up == 2)   // transposed 2x2 / stride 2: element offset of output pixel (2 oy, 2 ox) of the [N, 2 Ho, 2 Wo, Cout / 4] map
                ooff[j] = (unsigned)((n * 2 * sg.Ho + 2 * oy) * 2 * sg.Wo + 2 * ox) * (unsigned)(p.Cout >> 2);
            roff[j] = p.res_up == 1 ? (unsigned)((n * (sg.Ho >> 1) + (oy >> 1)) * (sg.Wo >> 1) + (ox >> 1)) * (unsigned)p.Cout : ooff[j];
        }
    }
    const int cb0 = (n_t * NWC + wc) * 2;         // first of this wave's two 32-channel blocks
    const unsigned wvo = (unsigned)(cb0 * 32 + l32) * 64u + 16u * (unsigned)lhalf;
    const unsigned slab_bytes = (unsigned)p.ldw * 64u;
#define C1_WLOAD(I, KG) c1_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo + (I) * 2048u, (unsigned)((KG) >> 1) * slab_bytes + 32u * (unsigned)((KG) & 1), 0))
#define C1_XLOAD(J, KG) c1_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff[J], 32u * (unsigned)(KG), 0))

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ksteps = p.Cin >> 4;
    c1_bf16x8 wq[D][2], xq[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < ksteps) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wq[d][i] = C1_WLOAD(i, d);
#pragma unroll
            for (int j = 0; j < 2; ++j) xq[d][j] = C1_XLOAD(j, d);
        }
    for (int base = 0; base < ksteps; base += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (base + d < ksteps) {
                c1_bf16x8 wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = wq[d][i];
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = xq[d][j];
                if (base + d + D < ksteps) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) wq[d][i] = C1_WLOAD(i, base + d + D);
#pragma unroll
                    for (int j = 0; j < 2; ++j) xq[d][j] = C1_XLOAD(j, base + d + D);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef C1_WLOAD
#undef C1_XLOAD

    // ---- epilogue: + bias, + shortcut (same size, or at half resolution through a nearest x2 upsampling), ReLU.
    // The accumulators (lane = pixel, 4 consecutive channels per register group) go through a wave-private LDS tile [64 pixels][64
    // channels] fp32, so that the shortcut is read and the result written with 16 bytes per lane and the lanes of a pixel side by
    // side: every load / store instruction covers whole 128-byte lines (8-byte accesses scattered over 32 pixels per instruction
    // ran at 1.7 TB/s).
    __shared__ __attribute__((aligned(16))) unsigned char EP[4][64 * C1_EPP];
    __shared__ unsigned EPO[4][64][2];
    unsigned char *ep = EP[wave];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (lhalf == 0) { EPO[wave][j * 32 + l32][0] = ooff[j]; EPO[wave][j * 32 + l32][1] = roff[j]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(ep + (j * 32 + l32) * C1_EPP + (i * 32 + 8 * g + 4 * lhalf) * 4) =
                    make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    }
    // the tile is written by (pixel, channel group) lanes and read back by a different lane mapping: order the LDS accesses of the
    // wave (compiler ordering only -- a wavefront's own LDS accesses execute in order; no instruction is emitted)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool has_res = sg.res != nullptr, has_bias = p.bias != nullptr, res16 = (p.io & 4) != 0;
    const unsigned oelems = (unsigned)sg.M * (unsigned)p.Cout;
    const unsigned relems = p.res_up == 1 ? (unsigned)(sg.N * (sg.Ho >> 1) * (sg.Wo >> 1)) * (unsigned)p.Cout : oelems;
    const bool scatter = p.res_up == 2;          // GEMM column (ky, kx, c) -> output pixel (2 oy + ky, 2 ox + kx), channel c
    const int c4 = p.Cout >> 2;
    const __amdgpu_buffer_rsrc_t orsrc = c1_rsrc(sg.out, oelems * (OUT16 ? 2u : 4u));
    const __amdgpu_buffer_rsrc_t rrsrc = c1_rsrc(has_res ? sg.res : sg.out, has_res ? relems * (res16 ? 2u : 4u) : 0u);
    constexpr int CPL = OUT16 ? 8 : 4;            // channels per lane and pass (16-byte stores)
    constexpr int LPP = 64 / CPL;                 // lanes per pixel
#pragma unroll
    for (int k = 0; k < 64 / (64 / LPP); ++k) {
        const int pl = (lane + 64 * k) / LPP, cq = (lane + 64 * k) % LPP;
        const int ch = cb0 * 32 + cq * CPL;
        const unsigned oo = EPO[wave][pl][0], ro = EPO[wave][pl][1];
        float v[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c += 4) {
            const float4 a = *reinterpret_cast<const float4 *>(ep + pl * C1_EPP + (cq * CPL + c) * 4);
            v[c] = a.x; v[c + 1] = a.y; v[c + 2] = a.z; v[c + 3] = a.w;
        }
        if (has_bias) {
#pragma unroll
            for (int c = 0; c < CPL; c += 4) {
                const float4 b = *reinterpret_cast<const float4 *>(p.bias + (scatter ? ch % c4 : ch) + c);
                v[c] = v[c] + b.x; v[c + 1] = v[c + 1] + b.y; v[c + 2] = v[c + 2] + b.z; v[c + 3] = v[c + 3] + b.w;
            }
        }
        if (has_res) {
            if (res16) {
                if (CPL == 8) {
                    const c1_uintx4 r = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (ro + (unsigned)ch) * 2u, 0, 0);
                    const unsigned rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) { v[2 * c] = v[2 * c] + __uint_as_float(rw[c] << 16); v[2 * c + 1] = v[2 * c + 1] + __uint_as_float(rw[c] & 0xffff0000u); }
                } else {
                    const c1_uintx2 r = __builtin_amdgcn_raw_buffer_load_b64(rrsrc, (ro + (unsigned)ch) * 2u, 0, 0);
                    v[0] = v[0] + __uint_as_float(r.x << 16); v[1] = v[1] + __uint_as_float(r.x & 0xffff0000u);
                    v[2] = v[2] + __uint_as_float(r.y << 16); v[3] = v[3] + __uint_as_float(r.y & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPL; c += 4) {
                    const c1_uintx4 r = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (ro + (unsigned)(ch + c)) * 4u, 0, 0);
                    v[c] = v[c] + __uint_as_float(r.x); v[c + 1] = v[c + 1] + __uint_as_float(r.y);
                    v[c + 2] = v[c + 2] + __uint_as_float(r.z); v[c + 3] = v[c + 3] + __uint_as_float(r.w);
                }
            }
        }
        if (p.relu) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        c1_uintx4 pk;
        if (OUT16) { pk.x = c1_pack2(v[0], v[1]); pk.y = c1_pack2(v[2], v[3]); pk.z = c1_pack2(v[CPL - 4], v[CPL - 3]); pk.w = c1_pack2(v[CPL - 2], v[CPL - 1]); }
        else { pk.x = __float_as_uint(v[0]); pk.y = __float_as_uint(v[1]); pk.z = __float_as_uint(v[2]); pk.w = __float_as_uint(v[3]); }
        unsigned oe = oo + (unsigned)ch;
        if (scatter) { const int q = ch / c4; oe = oo + (unsigned)(((q >> 1) * 2 * sg.Wo + (q & 1)) * c4 + (ch - q * c4)); }
        __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, oe * (OUT16 ? 2u : 4u), 0, 0);
    }
}

static int g_c1_on = -1;       // -1: from the environment on first use (UPSNET_BF16_WREG1)
static bool g_c1_all = false;  // tests: take every layer the kernel can compute, not only those it is faster on

/* A/B switch of the no-LDS 1x1 kernel of the bf16 mode: 0 = the layers stay on conv_bf16_kernel, 1 = default (the layers it is
 * faster on), 2 = every layer it can compute. Same results. */
extern "C" int upsnet_conv1x1_bf16_tuning(int enable)
{
    UPS_REQUIRE(enable >= 0 && enable <= 2, "conv1x1_bf16_tuning: enable 0/1/2");
    g_c1_on = enable != 0; g_c1_all = enable == 2;
    return 0;
}

// Does this launch fit the kernel? (1x1 / pad 0 / plain bf16 is checked by the caller.) bf16 inputs, channels in blocks of 64 / 128.
// Taken where it wins (tools/microbench_conv1x1_bf16.py): layers with a shortcut and a short K walk -- conv3 + shortcut of the first
// bottlenecks (89 -> 46 us at res2, 48 -> 31 us at res3), the P2 lateral (92 -> 75 us). With K >= 512 the 32-byte-per-pixel
// activation loads of four waves saturate the texture path (projection /2 of res3: 37 vs 26 us) and conv_bf16_kernel's LDS staging
// is the better K walk.
bool conv1x1_wreg_bf16_supported(const ConvParams &p)
{
    if (g_c1_on < 0) g_c1_on = !(getenv("UPSNET_BF16_WREG1") != nullptr && getenv("UPSNET_BF16_WREG1")[0] == '0');
    if (!(g_c1_on && (p.io & 1) && p.Cin % 64 == 0 && p.Cout % 128 == 0 && p.ldw == p.Cout && (p.stride == 1 || p.stride == 2))) return false;
    if (g_c1_all) return true;
    bool res = true;
    for (int i = 0; i < p.nseg; ++i) res = res && p.seg[i].res != nullptr;
    return res && p.Cin <= 256;
}

int conv1x1_wreg_bf16_launch(hipStream_t st, ConvParams &p, const void *wpack_hi)
{
    // 256 channels x 64 pixels per workgroup, or 128 x 128 where the channel count is not a multiple of 256
    const int nwc = p.Cout % 256 == 0 ? 4 : 2;
    const int tp = 64 * (4 / nwc);
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) { p.seg[i].tile_start = tiles; tiles += (int)((p.seg[i].M + tp - 1) / tp); }
    p.m_tiles = tiles;
    p.n_tiles = p.Cout / (64 * nwc);
    const int grid = 8 * ((tiles + 7) / 8) * p.n_tiles;
    const char *w = reinterpret_cast<const char *>(wpack_hi);
    const bool out16 = (p.io & 2) != 0;
    if (nwc == 4) {
        if (out16) hipLaunchKernelGGL((conv1x1_wreg_bf16_kernel<4, 1>), dim3(grid), dim3(256), 0, st, p, w);
        else hipLaunchKernelGGL((conv1x1_wreg_bf16_kernel<4, 0>), dim3(grid), dim3(256), 0, st, p, w);
    } else {
        if (out16) hipLaunchKernelGGL((conv1x1_wreg_bf16_kernel<2, 1>), dim3(grid), dim3(256), 0, st, p, w);
        else hipLaunchKernelGGL((conv1x1_wreg_bf16_kernel<2, 0>), dim3(grid), dim3(256), 0, st, p, w);
    }
    UPS_CHECK_LAUNCH("conv1x1_wreg_bf16_kernel");
    ups_set_form("conv1x1_wreg<%d,%d>", nwc, out16 ? 1 : 0);
    return 0;
}

/* ConvTranspose2d(kernel 2, stride 2, pad 0) (+ bias, + ReLU) of a bf16 NHWC map on the bf16 matrix cores: one GEMM
 * [N H W, Cin] x [Cin, 4 Cout] whose epilogue scatters column (ky, kx, c) of pixel (y, x) to output pixel (2 y + ky, 2 x + kx) -- the
 * mask head's upsampling layer (upsnet/models/rcnn.py:132-133) in the bf16 mode of BASELINE.json configs[2].
 * x [N,H,W,Cin] bf16; wpack_hi: upsnet_conv_pack_weight_bf16 of the [4 Cout, Cin, 1, 1] matrix with rows (ky, kx, c), ldw = 4 Cout;
 * bias [Cout] fp32 or NULL; out [N,2H,2W,Cout] fp32 (out_bf16 = 0) or bf16. Cin % 64 == 0, Cout % 32 == 0. */
extern "C" int upsnet_deconv2x2_nhwc_bf16(void *stream, const void *x, int batch, int height, int width, int Cin, const void *wpack_hi,
                                          int ldw, const float *bias, int Cout, int relu, void *out, int out_bf16)
{
    UPS_REQUIRE(x && wpack_hi && out, "deconv2x2_nhwc_bf16: null pointer");
    UPS_REQUIRE(Cin % 64 == 0 && Cout % 32 == 0 && ldw == 4 * Cout, "deconv2x2_nhwc_bf16: Cin %% 64 == 0, Cout %% 32 == 0, ldw == 4 Cout");
    ConvParams p;
    const float *xs[1] = {reinterpret_cast<const float *>(x)};
    float *outs[1] = {reinterpret_cast<float *>(out)};
    int rc = conv_fill(p, "deconv2x2_nhwc_bf16", 1, xs, nullptr, nullptr, nullptr, outs, &batch, &height, &width, Cin, 4 * Cout,
                       reinterpret_cast<const float *>(wpack_hi), ldw, bias, 1, 1, 1, 0, 1, relu != 0);
    if (rc) return rc;
    p.res_up = 2;
    p.io = 1 | (out_bf16 ? 2 : 0);
    return conv1x1_wreg_bf16_launch((hipStream_t)stream, p, wpack_hi);
}
