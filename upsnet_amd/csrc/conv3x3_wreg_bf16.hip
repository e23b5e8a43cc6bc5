// conv3x3_wreg_bf16.hip -- 3x3 / stride 1 / pad 1 convolution with output channels in blocks of 256 on the bf16 matrix cores, weights
// fed to the MFMA straight from L2 (gfx950). The FPN output convolutions (upsnet/models/fpn.py:38-41,98-101), the RPN's shared 3x3
// (upsnet/models/rpn.py:29,45), the four 3x3 layers of the mask head (upsnet/models/rcnn.py:122-131) -- all 256 -> 256 -- and conv2 of
// the res5 bottlenecks (512 -> 512) in the bf16 mode of BASELINE.json configs[2].
//
// conv3x3_bf16_halo_kernel (conv_bf16.hip) stages BOTH operands through LDS and synchronises once per tap: 8 MFMAs per wave and
// barrier, matrix pipe 31 % busy (profiles/r06). Here
//   * the MFMA operands are swapped: A = weights (rows = 32 output channels), B = activations (columns = 32 pixels). The packed
//     weights [tap * Cin/32 + c/32][column][32 k] (upsnet_conv_pack_weight_bf16) already hold, for a lane (column l, k half h) of a
//     k-step t, its eight k values in 16 consecutive bytes -- one buffer_load_dwordx4 per lane and MFMA operand (lane address in one
//     register, slab / k-step as the scalar offset), three k-steps ahead, no LDS, no barrier;
//   * a workgroup owns a TH x 16 pixel tile and 256 output channels (the haloed activation patch is read once, not once per
//     128-channel half): wave = 64 channels x 4 blocks of 2 x 16 pixels, 128 accumulator registers; TH = 8 (4 waves, two workgroups
//     per CU) by default; TH = 2 (one pixel block per wave) on maps with few tiles, there with 128-channel workgroups (NCB = 1) when
//     256-channel ones still leave half the CUs idle; TH = 16 (8 waves, one workgroup per CU) behind upsnet_conv_bf16_tuning;
//   * only the activations go through LDS, one 32-channel slab of the haloed patch at a time (double-buffered, fp32 -> bf16 on the
//     way in): ONE barrier per slab = per 144 MFMAs of a wave; activation fragments are read one k-step ahead;
//   * an accumulator lane holds 4 x 4 consecutive channels of one pixel: 16-byte stores.
// Same K order (slab, tap, k-step) and the same products as the halo kernel: bit-identical results for every tile form.
// Measured (tools/microbench_conv3x3_bf16.py, profiles/r09_bf16_micro.txt): FPN out 335 -> 212 us, RPN 299 -> 211 us (0.95-1.2
// PFLOP/s); PMC on those launches: matrix pipe 47 % busy, LDS bank conflicts 7 % of the LDS cycles, SQ_WAIT_INST_ANY 56 % of the wave
// cycles -- at two waves per SIMD the kernel waits on operand delivery (LDS and L1 each at about half their peak), not on the MFMAs.
#include <stdlib.h>

#include "conv_params.h"
#include "upsnet_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 w3_bf16x8;
typedef unsigned w3_uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned w3_uintx2 __attribute__((ext_vector_type(2)));

#define W3_TW 16
#define W3_PW (W3_TW + 2)
#define W3_XP 80          // bytes per patch pixel in LDS: 32 bf16 + 16 (consecutive pixels -> distinct banks for ds_read_b128)

__device__ static inline __amdgpu_buffer_rsrc_t w3_rsrc(const void *ptr, const unsigned bytes)
{
    const size_t a = reinterpret_cast<size_t>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// MFMA column l (0..31) -> pixel of a 2 x 16 block: each ds_read_b128 lane group reads 16 consecutive pixels of one image row
__device__ static inline int w3_perm(const int l)
{
    const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28;
    const int k = l < 4 ? l : l < 12 ? l - 4 : l < 16 ? l - 8 : l < 20 ? l - 8 : l < 28 ? l - 12 : l - 16;
    return (g1 ? 16 : 0) + k;
}

__device__ static inline w3_bf16x8 w3_as_bf16x8(const w3_uintx4 v)
{
    w3_bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

__device__ static inline unsigned w3_pack2(const float a, const float b)
{
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    unsigned short ux, uy;
    __builtin_memcpy(&ux, &x, 2);
    __builtin_memcpy(&uy, &y, 2);
    return (unsigned)ux | ((unsigned)uy << 16);
}

// NCB: 32-channel blocks per wave: 2 (256 output channels per workgroup) or 1 (128: twice the workgroups, for maps with few tiles)
template <int TH, int IO, int NCB = 2>
__global__ void __launch_bounds__(TH == 16 ? 512 : 256, TH == 16 ? 1 : 2) conv3x3_wreg_bf16_kernel(const ConvParams p, const char *__restrict__ wpk)
{
    constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
    constexpr int NT = TH == 16 ? 512 : 256;                        // threads: 4 waves per 8 tile rows
    constexpr int NPB = TH >= 8 ? 4 : TH / 2;                       // 2 x 16 pixel blocks per wave
    constexpr int NPATCH = (TH + 2) * W3_PW, NROWS = (NPATCH + 31) / 32 * 32;
    constexpr int UPR = IN16 ? 4 : 8;                               // 16-byte units per patch pixel and slab
    constexpr int NLD = (NROWS * UPR + NT - 1) / NT;
    constexpr int WD = 3;                                           // weight k-steps in flight (18 per slab, a multiple of WD; 6 measured the same)
    __shared__ __attribute__((aligned(16))) unsigned char XS[2][NROWS * W3_XP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, lhalf = lane >> 5;
    const int wc = wave & 3, wp = wave >> 2;
    int m_t, n_t;
    {   // XCD-aware tile order: consecutive tiles of a map share an XCD's L2 (block b runs on XCD b % 8)
        const int bid = blockIdx.x;
        const int per = (p.m_tiles + 7) >> 3;
        const int q = bid >> 3;
        n_t = q % p.n_tiles;                      // 256-channel block of the output
        m_t = (bid & 7) * per + q / p.n_tiles;
        if (m_t >= p.m_tiles) return;
    }
    int si = 0;
#pragma unroll
    for (int q = 1; q < CV_MAXSEG; ++q) if (q < p.nseg && m_t >= p.seg[q].tile_start) si = q;
    const ConvSeg sg = p.seg[si];
    const int cslabs = p.Cin / 32;
    const int tiles_x = (sg.Wo + W3_TW - 1) / W3_TW, tiles_y = (sg.Ho + TH - 1) / TH;
    const int t_loc = m_t - sg.tile_start;
    const int t_n = t_loc / (tiles_x * tiles_y), t_rem = t_loc - t_n * (tiles_x * tiles_y);
    const int t_y = t_rem / tiles_x, t_x = t_rem - t_y * tiles_x;
    const int y0 = TH * t_y, x0 = W3_TW * t_x;
    const unsigned XB = IN16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t xrsrc = w3_rsrc(sg.x, (unsigned)(sg.N * sg.H * sg.W) * XB * (unsigned)p.Cin);

    // patch loader: unit u = tid + NT j -> patch pixel u / UPR, 16 bytes (4 fp32 / 8 bf16 channels) u % UPR of the slab
    unsigned po[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int u = tid + NT * j, q = u / UPR;
        const int py = y0 - 1 + q / W3_PW, px = x0 - 1 + q % W3_PW;
        po[j] = (q < NPATCH && py >= 0 && py < sg.H && px >= 0 && px < sg.W)
                    ? ((unsigned)((t_n * sg.H + py) * sg.W + px) * (unsigned)p.Cin) * XB + 16u * (unsigned)(u % UPR) : 0x80000000u;
    }
    w3_uintx4 rx[NLD];
#define W3_FETCH_X(CS) { _Pragma("unroll") for (int j = 0; j < NLD; ++j) rx[j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, po[j], (unsigned)(CS) * (32u * XB), 0); }
#define W3_STASH_X(BUF)                                                                                                   \
    { _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                                   \
        const int u = tid + NT * j, q = u / UPR;                                                                          \
        if (NROWS * UPR % NT == 0 || u < NROWS * UPR) {                                                                   \
            if (IN16) *reinterpret_cast<w3_uintx4 *>(&XS[BUF][q * W3_XP + (u % UPR) * 16]) = rx[j];                        \
            else {                                                                                                        \
                w3_uintx2 h_;                                                                                             \
                h_.x = w3_pack2(__uint_as_float(rx[j].x), __uint_as_float(rx[j].y));                                      \
                h_.y = w3_pack2(__uint_as_float(rx[j].z), __uint_as_float(rx[j].w));                                      \
                *reinterpret_cast<w3_uintx2 *>(&XS[BUF][q * W3_XP + (u % UPR) * 8]) = h_;                                  \
            }                                                                                                             \
        } } }
    // weight fragment of this wave's 32-column block I, slab S (= tap * cslabs + cs), k-step T: 16 bytes per lane; the lane part of
    // the address is one register, the slab / k-step part wave-uniform (SGPR offset of the buffer load)
    const __amdgpu_buffer_rsrc_t wrsrc = w3_rsrc(wpk, 9u * (unsigned)p.Cin * (unsigned)p.ldw * 2u);
    const int cb0 = (4 * n_t + wc) * NCB;       // first 32-channel block of this wave
    const unsigned wvo = (unsigned)(cb0 * 32 + l32) * 64u + 16u * (unsigned)lhalf;
    const unsigned slab_bytes = (unsigned)p.ldw * 64u;
#define W3_WLOAD(I, S, T) w3_as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo + (I) * 2048u, (unsigned)(S) * slab_bytes + 32u * (unsigned)(T), 0))

    floatx16 acc[NCB][NPB];
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
        for (int j = 0; j < NPB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int prow[NPB];                   // lane's pixel in block j: byte offset of the top-left patch pixel of its 3x3 window
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int pp = w3_perm(l32), y = 2 * (wp * 4 + j) + (pp >> 4), x = pp & 15;
        prow[j] = (y * W3_PW + x) * W3_XP + lhalf * 16;
    }

    w3_bf16x8 wq[WD][NCB];
    W3_FETCH_X(0)
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int i = 0; i < NCB; ++i) wq[d][i] = W3_WLOAD(i, (d >> 1) * cslabs, d & 1);
    __builtin_amdgcn_sched_barrier(0);
    W3_STASH_X(0)
    __syncthreads();
    for (int cs = 0; cs < cslabs; ++cs) {
        const int buf = cs & 1;
        const bool more = cs + 1 < cslabs;
        if (more) W3_FETCH_X(cs + 1)
        __builtin_amdgcn_sched_barrier(0);
#define W3_XFRAG(J, KK) (*reinterpret_cast<const w3_bf16x8 *>(&XS[buf][prow[J] + ((((KK) >> 1) / 3) * W3_PW + (((KK) >> 1) % 3)) * W3_XP + ((KK) & 1) * 32]))
        w3_bf16x8 xf[NPB], xn[NPB];                  // activation fragments one k-step ahead (inside a slab)
#pragma unroll
        for (int j = 0; j < NPB; ++j) xf[j] = W3_XFRAG(j, 0);
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {            // k-step kk = (tap, t) of this slab
            w3_bf16x8 wf[NCB];
#pragma unroll
            for (int i = 0; i < NCB; ++i) wf[i] = wq[kk % WD][i];
            {   // k-step kk + WD: same slab of channels while it lasts, then the first steps of the next one
                const int kn = kk + WD;
                const int ntap = (kn % 18) >> 1, nt = kn & 1;
                const int ncs = kn < 18 ? cs : cs + 1;
                if (kn < 18 || more) {
#pragma unroll
                    for (int i = 0; i < NCB; ++i) wq[kk % WD][i] = W3_WLOAD(i, ntap * cslabs + ncs, nt);
                }
            }
            if (kk + 1 < 18) {
#pragma unroll
                for (int j = 0; j < NPB; ++j) xn[j] = W3_XFRAG(j, kk + 1);
            }
#pragma unroll
            for (int j = 0; j < NPB; ++j)
#pragma unroll
                for (int i = 0; i < NCB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NPB; ++j) xf[j] = xn[j];
            __builtin_amdgcn_sched_barrier(0);
        }
#undef W3_XFRAG
        if (more) W3_STASH_X(buf ^ 1)
        __syncthreads();
    }
#undef W3_FETCH_X
#undef W3_STASH_X
#undef W3_WLOAD

    // ---- epilogue: + bias, ReLU; lane = one pixel, 4 x 4 consecutive channels per 32-channel block
    const __amdgpu_buffer_rsrc_t orsrc = w3_rsrc(sg.out, (unsigned)sg.M * (unsigned)p.Cout * (OUT16 ? 2u : 4u));
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
        float4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b[g] = has_bias ? *reinterpret_cast<const float4 *>(p.bias + (cb0 + i) * 32 + 8 * g + 4 * lhalf) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const int pp = w3_perm(l32);
            const int oy = y0 + 2 * (wp * 4 + j) + (pp >> 4), ox = x0 + (pp & 15);
            const unsigned pix = (oy < sg.Ho && ox < sg.Wo) ? (unsigned)((t_n * sg.Ho + oy) * sg.Wo + ox) * (unsigned)p.Cout : 0x20000000u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[i][j][4 * g + 0] + b[g].x, v1 = acc[i][j][4 * g + 1] + b[g].y;
                float v2 = acc[i][j][4 * g + 2] + b[g].z, v3 = acc[i][j][4 * g + 3] + b[g].w;
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const unsigned e = pix + (unsigned)((cb0 + i) * 32 + 8 * g + 4 * lhalf);
                if (OUT16) {
                    w3_uintx2 pk;
                    pk.x = w3_pack2(v0, v1); pk.y = w3_pack2(v2, v3);
                    __builtin_amdgcn_raw_buffer_store_b64(pk, orsrc, e * 2u, 0, 0);
                } else {
                    w3_uintx4 pk;
                    pk.x = __float_as_uint(v0); pk.y = __float_as_uint(v1); pk.z = __float_as_uint(v2); pk.w = __float_as_uint(v3);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, e * 4u, 0, 0);
                }
            }
        }
    }
}

static int g_wreg_on = -1, g_wreg_th = -1;     // -1: from the environment on first use (UPSNET_BF16_WREG, UPSNET_BF16_WREG_TH)

/* A/B switch of the 3x3 bf16 kernel of this file: enable 0 = the layers stay on conv3x3_bf16_halo_kernel, 1 = this file's kernel
 * (default); tile_rows 0 = automatic, 8 or 16 = force the tile height, 2 = 2-row tiles with 256-channel workgroups, 1 = 2-row tiles
 * with 128-channel workgroups. Results do not depend on either. */
extern "C" int upsnet_conv_bf16_tuning(int enable, int tile_rows)
{
    UPS_REQUIRE((enable == 0 || enable == 1) && (tile_rows == 0 || tile_rows == 1 || tile_rows == 2 || tile_rows == 8 || tile_rows == 16),
                "conv_bf16_tuning: enable 0/1, tile_rows 0/1/2/8/16");
    g_wreg_on = enable; g_wreg_th = tile_rows;
    return 0;
}

// Does this launch fit the kernel? (3x3 / 1 / 1 is checked by the caller.) Output channels in blocks of 256, no residual, plain bf16
// products.
bool conv3x3_wreg_bf16_supported(const ConvParams &p)
{
    if (g_wreg_on < 0) {
        g_wreg_on = !(getenv("UPSNET_BF16_WREG") != nullptr && getenv("UPSNET_BF16_WREG")[0] == '0');
        g_wreg_th = getenv("UPSNET_BF16_WREG_TH") ? atoi(getenv("UPSNET_BF16_WREG_TH")) : 0;
    }
    if (!g_wreg_on || p.Cout % 256 != 0 || p.ldw != p.Cout || p.Cin % 32 != 0 || p.res_up) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].res) return false;
    return true;
}

int conv3x3_wreg_bf16_launch(hipStream_t st, ConvParams &p, const void *wpack_hi)
{
    // 8 x 16 pixel tiles, two workgroups per CU. (16 x 16 tiles -- 8 waves, the patch halo and the weight fragments shared by twice
    // the pixels, one workgroup per CU -- measured 3-7 % slower on the FPN / RPN maps and 50 % slower on the mask head's 14 x 14
    // ROIs: tools/microbench_conv3x3_bf16.py; kept behind upsnet_conv_bf16_tuning.) Small maps (res4 / res5 at 1024x2048: 64 / 16
    // tiles of 8 x 16): 2 x 16 tiles, four times the workgroups (each streams the whole weight block for 32 pixels: only where
    // the chip would otherwise idle).
    auto count = [&](int th) {
        int t = 0;
        for (int i = 0; i < p.nseg; ++i) t += p.seg[i].N * ((p.seg[i].Ho + th - 1) / th) * ((p.seg[i].Wo + W3_TW - 1) / W3_TW);
        return t;
    };
    const int n256 = p.Cout / 256;
    const int th = (g_wreg_th == 8 || g_wreg_th == 16) ? g_wreg_th : (g_wreg_th == 1 || g_wreg_th == 2) ? 2 : (count(8) * n256 < 72 ? 2 : 8);
    // 128-channel workgroups (one 32-channel block per wave) where 2 x 16 tiles x 256-channel blocks still leave half the CUs idle
    // (res5 at 1024x2048: 64 tiles x 2)
    const bool narrow = th == 2 && (g_wreg_th == 1 || (g_wreg_th == 0 && count(2) * n256 <= 128));
    p.n_tiles = narrow ? p.Cout / 128 : n256;
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) {
        p.seg[i].tile_start = tiles;
        tiles += p.seg[i].N * ((p.seg[i].Ho + th - 1) / th) * ((p.seg[i].Wo + W3_TW - 1) / W3_TW);
    }
    p.m_tiles = tiles;
    const int grid = 8 * ((tiles + 7) / 8) * p.n_tiles;
    const char *w = reinterpret_cast<const char *>(wpack_hi);
#define W3_GO(TH_, IO_, NCB_) hipLaunchKernelGGL((conv3x3_wreg_bf16_kernel<TH_, IO_, NCB_>), dim3(grid), dim3(TH_ == 16 ? 512 : 256), 0, st, p, w)
#define W3_GO_IO(TH_, NCB_) switch (p.io & 3) { case 0: W3_GO(TH_, 0, NCB_); break; case 1: W3_GO(TH_, 1, NCB_); break; case 2: W3_GO(TH_, 2, NCB_); break; default: W3_GO(TH_, 3, NCB_); break; }
    if (th == 16) W3_GO_IO(16, 2)
    else if (th == 8) W3_GO_IO(8, 2)
    else if (narrow) W3_GO_IO(2, 1)
    else W3_GO_IO(2, 2)
#undef W3_GO_IO
#undef W3_GO
    UPS_CHECK_LAUNCH("conv3x3_wreg_bf16_kernel");
    ups_set_form("conv3x3_wreg<%d,%d>", th, (th == 2 && narrow) ? 1 : 2);
    return 0;
}
