// conv_bf16.hip -- dense convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
// BASELINE.json configs[2] ("hand-written MFMA 3x3/1x1 backbone + FPN convs, NHWC, bf16 compute / fp32 accumulate") and its
// fp32-equivalent variant. The fp32 kernel (conv.hip) is the default of the framework and the one the headline is measured on;
// this file is opt-in (hipconv.PRECISION / UPSNET_CONV_PRECISION):
//   SPLIT 1 ("bf16")    a, b rounded to bf16 (round to nearest even), one MFMA per product: ~3 significant digits.
//   SPLIT 3 ("bf16x3")  a = a_hi + a_lo, b = b_hi + b_lo (both parts bf16), a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi: the dropped
//                       term is ~2^-16 relative, the sum is accumulated in fp32 -- within the 1e-4 fp32-logit tolerance of the
//                       north star, at 3/16 of the MFMA time of exact fp32 products.
// Activations are fp32 in HBM by default (NHWC, same tensors as the fp32 path) and are rounded / split to bf16 when the staged
// registers are written to LDS. r08 (bf16 mode only): a layer may also READ a bf16 NHWC tensor (IO bit 0: 16-byte loads carry 8
// channels and go to LDS unconverted), WRITE bf16 (IO bit 1: fp32 accumulator + bias + residual + ReLU, rounded once) and add a
// bf16 residual (run-time flag) -- the backbone then keeps its activations in bf16 between layers: the 1x1 layers of the bf16
// mode were HBM-bound on fp32 activations (res3 conv3: 151 MB for 4.3 GFLOP = 40 us at 4 TB/s, 4 % of the bf16 MFMA peak). Weights are split and packed once (upsnet_conv_pack_weight_bf16) as [K slab][column][32 k], so a lane's MFMA
// operand (8 consecutive k of one row / column) is ONE 16-byte LDS read: As/Bs are [64 rows][32 k] bf16 with an 80-byte row
// pitch (conflict-free ds_read_b128). Workgroup = 64 pixels x 64 channels, 4 waves (32 x 32 each), K slabs of one tap x 32
// channels, double-buffered LDS, loads of slab s+1 in flight while slab s is contracted, one barrier per slab; operand
// addressing (incremental 32-bit offsets), XCD-aware tile order, multi-map launches and the bias/residual/ReLU epilogue are
// those of conv.hip.
#include <stdlib.h>

#include "conv_params.h"
#include "upsnet_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define CB_BM 128
#define CB_BN 128
#define CB_BK 32
#define CB_PITCH 32   // bf16 elements per LDS row (64 bytes, no padding): the four 16-byte octets of a row are XOR-swizzled by (row >> 2) & 3
#define CB_SW(ROW, OCT) (8 * ((OCT) ^ (((ROW) >> 2) & 3)))   // element offset of octet OCT inside row ROW (conflict-free b64 / b128)

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

// epilogue helpers: buffer descriptor over a tensor, element load (fp32 or bf16 -> fp32), element store (fp32 or bf16); an element
// index of 0x3FFFFFFF is beyond every tensor (x 4 and x 2 bytes both exceed the 2 GiB limit of the entry point without wrapping): the load returns 0, the store is dropped
__device__ static inline __amdgpu_buffer_rsrc_t cb_rsrc(const void *ptr, const unsigned bytes)
{
    const size_t a = reinterpret_cast<size_t>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ static inline float cb_load_elem(const __amdgpu_buffer_rsrc_t r, const unsigned elem, const bool is16)
{
    if (is16) {
        const unsigned short b = __builtin_amdgcn_raw_buffer_load_b16(r, elem * 2u, 0, 0);
        return __uint_as_float((unsigned)b << 16);
    }
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, elem * 4u, 0, 0));
}
template <bool IS16>
__device__ static inline void cb_store_elem(const __amdgpu_buffer_rsrc_t r, const unsigned elem, const float v)
{
    if (IS16) {
        const __bf16 h = (__bf16)v;      // round to nearest even
        unsigned short b;
        __builtin_memcpy(&b, &h, 2);
        __builtin_amdgcn_raw_buffer_store_b16(b, r, elem * 2u, 0, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, elem * 4u, 0, 0);
    }
}

// Tile (r06): 128 pixels x 128 channels per workgroup, 4 waves of 64 x 64 (2 x 2 MFMA blocks, 64 accumulator registers). The r01
// form (64 x 64 per workgroup, 32 x 32 per wave) moved 16 KiB from L2 for every 2 MFMAs of a wave: 21.8 flop per L2 byte, i.e.
// 20 TB/s of L2 traffic at the 445 TFLOP/s it reached -- it was L2-bound at 18 % of the bf16 peak. This form doubles the
// arithmetic intensity (43.7 flop/B: A 16 KiB fp32 + B 8 KiB bf16 per 32-channel slab of a 128 x 128 tile), issues 8 (bf16) or
// 24 (bf16x3) MFMAs per wave between two barriers instead of 2 / 6, and reads 4 LDS fragments per 4 / 12 MFMAs instead of 2 per 1 / 3.
// WN: 32-column MFMA blocks per wave: 2 -> 128 output channels per workgroup (used by both modes: 69 KiB of LDS for bf16x3 with
// the unpadded swizzled rows = two workgroups per CU), 1 -> 64 (kept for measurements: slower, 629 vs ~520 us on FPN P2).
template <int SPLIT, int WN, int IO = 0>
__global__ void __launch_bounds__(256, 2)
conv_bf16_kernel(const ConvParams p, const __bf16 *__restrict__ whi, const __bf16 *__restrict__ wlo)
{
    constexpr int BN = 64 * WN;
    constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
    constexpr unsigned XB = IN16 ? 2u : 4u;      // bytes per input element
    static_assert(!(IN16 && SPLIT == 3), "bf16 inputs carry no low part");
    __shared__ __attribute__((aligned(16))) __bf16 Ah[2][CB_BM][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bh[2][BN][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Al[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? CB_BM : 1][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bl[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? BN : 1][CB_PITCH];

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
    const int n0 = n_t * BN;
    const int ntap = p.KH * p.KW;
    const int nslabs = ntap * (p.Cin / CB_BK);

    // A staging: thread -> channels 4*ch4..+3 of pixels prow + 32 r, r = 0..3. The byte offset of every (tap, pixel) input
    // position is tabulated once in LDS (bit 31 = outside the image / beyond the map: the buffer load then returns the zero
    // padding), and K is walked channel slab outermost, TAP innermost: the 128-byte lines of a slab are shared by neighbouring
    // taps (a 3x3 tap shifts the 128-pixel window by one pixel / one row), so 8 of 9 tap loads of a line hit the CU's L1
    // instead of all of them going to L2 (the tap-outer walk touched 128 KiB of lines per tap: nothing survived in 32 KiB).
    const int ch4 = tid & 7, prow = tid >> 3;
    const long HoWo = (long)sg.Ho * sg.Wo;
    __shared__ unsigned toff[9][CB_BM];
    for (int idx = tid; idx < ntap * CB_BM; idx += 256) {
        const int tap = idx / CB_BM, px = idx - tap * CB_BM;
        const long pp = p0 + px;
        unsigned o = 0x80000000u;
        if (pp < sg.M) {
            const int n = (int)(pp / HoWo);
            const int rem = (int)(pp - (long)n * HoWo);
            const int ki = tap / p.KW, kj = tap - ki * p.KW;
            const int hi = (rem / sg.Wo) * p.stride - p.pad + ki * p.dil, wi = (rem % sg.Wo) * p.stride - p.pad + kj * p.dil;
            if (hi >= 0 && hi < sg.H && wi >= 0 && wi < sg.W) o = XB * (unsigned)(((n * sg.H + hi) * sg.W + wi) * p.Cin);
        }
        toff[tap][px] = o;
    }
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * XB * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)xhi << 32) | xlo), 0, (int)xbytes, 0x00020000);
    // B staging: thread -> 8 consecutive k (one 16-byte octet) of columns bcol and bcol + 64
    const int bcol = tid >> 2, boct = tid & 3;
    const unsigned slab_bytes = (unsigned)p.ldw * CB_BK * 2u;
    const unsigned ob0 = ((unsigned)(n0 + bcol) * CB_BK + 8u * boct) * 2u;   // byte offset inside a slab of the packed weights
    const char *whb = reinterpret_cast<const char *>(whi), *wlb = reinterpret_cast<const char *>(wlo);
    const int cslabs = p.Cin / CB_BK;

    int f_cs = 0, f_tap = 0;   // (channel slab, tap) of the next step to fetch
    float4 ra0, ra1, ra2, ra3;
    // bf16 inputs: thread -> octet (8 channels, 16 bytes) oct16 of pixels prow16 and prow16 + 64: two loads, two ds_write_b128
    const int oct16 = tid & 3, prow16 = tid >> 2;
    uintx4 rx0, rx1;
    uint4 rbh0, rbh1, rbl0, rbl1;
    const unsigned ob64 = 64u * CB_BK * 2u;   // byte distance of column bcol + 64 inside a slab

    floatx16 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __syncthreads();   // offset table complete

#define CB_LDX(D, O) { const uintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (O), 0, 0); \
        D = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }
#define CB_FETCH                                                                                          \
    {                                                                                                     \
        if (IN16) {                                                                                       \
            const unsigned c_ = (unsigned)(f_cs * (CB_BK * 2) + oct16 * 16);                              \
            rx0 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, toff[f_tap][prow16] + c_, 0, 0);           \
            rx1 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, toff[f_tap][prow16 + 64] + c_, 0, 0);      \
        } else {                                                                                          \
        const unsigned c_ = (unsigned)(f_cs * (CB_BK * 4) + ch4 * 16);                                    \
        CB_LDX(ra0, toff[f_tap][prow] + c_) CB_LDX(ra1, toff[f_tap][prow + 32] + c_)                      \
        CB_LDX(ra2, toff[f_tap][prow + 64] + c_) CB_LDX(ra3, toff[f_tap][prow + 96] + c_)                 \
        }                                                                                                 \
        const unsigned ob = ob0 + (unsigned)(f_tap * cslabs + f_cs) * slab_bytes;                         \
        rbh0 = *reinterpret_cast<const uint4 *>(whb + ob);                                                \
        if (WN == 2) rbh1 = *reinterpret_cast<const uint4 *>(whb + ob + ob64);                            \
        if (SPLIT == 3) { rbl0 = *reinterpret_cast<const uint4 *>(wlb + ob); if (WN == 2) rbl1 = *reinterpret_cast<const uint4 *>(wlb + ob + ob64); } \
        if (++f_tap == ntap) { f_tap = 0; ++f_cs; }                                                       \
    }
#define CB_STASH_PX(BUF, R)                                                                               \
    {                                                                                                     \
        bf16x4 h_, l_;                                                                                    \
        cb_split4(ra##R, true, h_, l_);                                                                  \
        *reinterpret_cast<bf16x4 *>(&Ah[BUF][prow + 32 * R][CB_SW(prow + 32 * R, ch4 >> 1) + 4 * (ch4 & 1)]) = h_;                               \
        if (SPLIT == 3) *reinterpret_cast<bf16x4 *>(&Al[BUF][prow + 32 * R][CB_SW(prow + 32 * R, ch4 >> 1) + 4 * (ch4 & 1)]) = l_;               \
    }
#define CB_STASH(BUF)                                                                                     \
    {                                                                                                     \
        if (IN16) {                                                                                       \
            *reinterpret_cast<uintx4 *>(&Ah[BUF][prow16][CB_SW(prow16, oct16)]) = rx0;                    \
            *reinterpret_cast<uintx4 *>(&Ah[BUF][prow16 + 64][CB_SW(prow16 + 64, oct16)]) = rx1;          \
        } else { CB_STASH_PX(BUF, 0) CB_STASH_PX(BUF, 1) CB_STASH_PX(BUF, 2) CB_STASH_PX(BUF, 3) }        \
        *reinterpret_cast<uint4 *>(&Bh[BUF][bcol][CB_SW(bcol, boct)]) = rbh0;                                      \
        if (WN == 2) *reinterpret_cast<uint4 *>(&Bh[BUF][bcol + 64][CB_SW(bcol + 64, boct)]) = rbh1;                   \
        if (SPLIT == 3) {                                                                                 \
            *reinterpret_cast<uint4 *>(&Bl[BUF][bcol][CB_SW(bcol, boct)]) = rbl0;                                  \
            if (WN == 2) *reinterpret_cast<uint4 *>(&Bl[BUF][bcol + 64][CB_SW(bcol + 64, boct)]) = rbl1;                \
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
            const int ko = 2 * t + akr;   // octet of the slab this lane's fragment holds
            bf16x8 ah[2], bh[WN];
#pragma unroll
            for (int i = 0; i < 2; ++i) ah[i] = *reinterpret_cast<const bf16x8 *>(&Ah[buf][wm * 64 + 32 * i + aij][CB_SW(wm * 64 + 32 * i + aij, ko)]);
#pragma unroll
            for (int j = 0; j < WN; ++j) bh[j] = *reinterpret_cast<const bf16x8 *>(&Bh[buf][wn * (32 * WN) + 32 * j + aij][CB_SW(wn * (32 * WN) + 32 * j + aij, ko)]);
            if (SPLIT == 3) {   // small terms first
                bf16x8 al[2], bl[WN];
#pragma unroll
                for (int i = 0; i < 2; ++i) al[i] = *reinterpret_cast<const bf16x8 *>(&Al[buf][wm * 64 + 32 * i + aij][CB_SW(wm * 64 + 32 * i + aij, ko)]);
#pragma unroll
                for (int j = 0; j < WN; ++j) bl[j] = *reinterpret_cast<const bf16x8 *>(&Bl[buf][wn * (32 * WN) + 32 * j + aij][CB_SW(wn * (32 * WN) + 32 * j + aij, ko)]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
        if (more) CB_STASH(buf ^ 1)
        __syncthreads();
    }
#undef CB_LDX
#undef CB_FETCH
#undef CB_STASH

    // ---- epilogue: + bias, + residual, ReLU (as conv.hip), per 32 x 32 block of the wave; residual fp32 or bf16 (p.io bit 2),
    // output fp32 or bf16 (OUT16), both through buffer descriptors with 32-bit element offsets
    const bool has_res = sg.res != nullptr, has_bias = p.bias != nullptr;
    const bool res16 = (p.io & 4) != 0;
    const int Hr = sg.Ho >> 1, Wr = sg.Wo >> 1;
    const unsigned oelems = (unsigned)sg.M * (unsigned)p.Cout;
    const unsigned relems = p.res_up ? (unsigned)(sg.N * Hr * Wr) * (unsigned)p.Cout : oelems;
    const __amdgpu_buffer_rsrc_t orsrc = cb_rsrc(sg.out, oelems * (OUT16 ? 2u : 4u));
    const __amdgpu_buffer_rsrc_t rrsrc = cb_rsrc(has_res ? sg.res : sg.out, relems * (res16 ? 2u : 4u));
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int co = n0 + wn * (32 * WN) + 32 * j + aij;
        const bool co_ok = co < p.Cout;
        const float bv = (has_bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long pbase = p0 + wm * 64 + 32 * i + 4 * akr;
            unsigned ridx[16];       // element index of the residual pixel's first channel
            if (has_res && p.res_up) {
                // residual at half resolution, read through a nearest x2 upsampling (FPN top-down add, fpn.py:34,90-96). The 32 rows
                // of the block are consecutive output pixels from a multiple of 32: with Wo % 32 == 0 they lie in one image row
                const long HoWo_ = (long)sg.Ho * sg.Wo;
                if ((sg.Wo & 31) == 0) {
                    const long pb0 = p0 + wm * 64 + 32 * i;
                    const long pb = pb0 < sg.M ? pb0 : 0;
                    const int n_b = (int)(pb / HoWo_);
                    const int rem_b = (int)(pb - (long)n_b * HoWo_);
                    const int h_b = rem_b / sg.Wo, w_b = rem_b - h_b * sg.Wo;
                    const unsigned rb = (unsigned)((n_b * Hr + (h_b >> 1)) * Wr + (w_b >> 1) + 2 * akr);
#pragma unroll
                    for (int r = 0; r < 16; ++r) ridx[r] = (rb + (unsigned)(((r & 3) + 8 * (r >> 2)) >> 1)) * (unsigned)p.Cout;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        long pp = pbase + (r & 3) + 8 * (r >> 2);
                        pp = pp < sg.M ? pp : sg.M - 1;
                        const int n = (int)(pp / HoWo_);
                        const int rem = (int)(pp - (long)n * HoWo_);
                        const int h = rem / sg.Wo, w = rem - h * sg.Wo;
                        ridx[r] = (unsigned)((n * Hr + (h >> 1)) * Wr + (w >> 1)) * (unsigned)p.Cout;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ridx[r] = (unsigned)(pbase + (r & 3) + 8 * (r >> 2)) * (unsigned)p.Cout;
            }
            float rr[16];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = co_ok && pbase + (r & 3) + 8 * (r >> 2) < sg.M;
                    rr[r] = cb_load_elem(rrsrc, ok ? ridx[r] + (unsigned)co : 0x3FFFFFFFu, res16);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long pp = pbase + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r];
                if (has_bias) v = v + bv;
                if (has_res) v = v + rr[r];
                if (p.relu) v = fmaxf(v, 0.f);
                cb_store_elem<OUT16>(orsrc, (co_ok && pp < sg.M) ? (unsigned)pp * (unsigned)p.Cout + (unsigned)co : 0x3FFFFFFFu, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a HALOED INPUT PATCH in LDS (r06). The general kernel above stages the A tile of every
// (channel slab, tap) step separately: 9 x 128 pixels x 32 channels per slab, although the nine taps look at the same
// (8+2) x (16+2) pixels. Here a workgroup owns an 8 x 16 block of output pixels (x 128 output channels) and stages the 180-pixel
// patch of a channel slab ONCE (bf16, 64 B per pixel, octets XOR-swizzled by (pixel >> 2) & 3); the A fragment of tap (ki, kj) is
// the same LDS read shifted by 18 ki + kj pixels. Per slab the L2 -> LDS traffic of a workgroup falls from 9 x (16 KiB fp32 A +
// 8 KiB B) = 216 KiB to 23 KiB A + 72 KiB B = 95 KiB (99 flop per byte instead of 43.7), the L1 -> LDS stream from 96 to 41 B/clk
// per CU at the full MFMA rate, and the fp32 -> bf16 conversions are done once per pixel instead of 9 times.
// B: one tap's [128 columns][32 k] tile per step through LDS (double-buffered), as above. One barrier per step (8 / 24 MFMAs).
#define H3_TH 8
#define H3_TW 16
#define H3_PW (H3_TW + 2)
#define H3_NPX ((H3_TH + 2) * H3_PW)                      // 180 patch pixels
#define H3_LD 6                                           // float4 loads per thread and slab: 6 * 256 >= 180 * 8
#define H3_SW(PIX, OCT) ((PIX) * 32 + 8 * ((OCT) ^ (((PIX) >> 2) & 3)))   // bf16 element offset of octet OCT of patch pixel PIX

// MFMA row l (0..31) -> pixel of the 2 x 16 block: lanes {0-3,12-15,20-27} -> row 0, x = 0..15 in that order; the others -> row 1
__device__ static inline int h3_perm(const int l)
{
    const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28;
    const int k = l < 4 ? l : l < 12 ? l - 4 : l < 16 ? l - 8 : l < 20 ? l - 8 : l < 28 ? l - 12 : l - 16;
    return (g1 ? 16 : 0) + k;
}

// TPS: taps per step (= per barrier and per weight stage): 1 in both modes. TPS = 3 (one kernel row per barrier, 24 MFMAs per wave
// and step in bf16 like bf16x3 has per tap) was measured SLOWER in bf16: 345 vs 227 us on FPN-P2 (244 registers, six weight
// loads + stores per thread in a row); kept as a template parameter for further experiments.
template <int SPLIT, int TPS, int IO = 0>
__global__ void __launch_bounds__(256, 2)
conv3x3_bf16_halo_kernel(const ConvParams p, const __bf16 *__restrict__ whi, const __bf16 *__restrict__ wlo)
{
    constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
    constexpr unsigned XB = IN16 ? 2u : 4u;
    static_assert(!(IN16 && SPLIT == 3), "bf16 inputs carry no low part");
    __shared__ __attribute__((aligned(16))) __bf16 Ph[2][H3_NPX * 32];
    __shared__ __attribute__((aligned(16))) __bf16 Pl[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? H3_NPX * 32 : 8];
    __shared__ __attribute__((aligned(16))) __bf16 Bh[2][TPS][CB_BN][CB_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bl[SPLIT == 3 ? 2 : 1][SPLIT == 3 ? TPS : 1][SPLIT == 3 ? CB_BN : 1][CB_PITCH];

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
    const int n0 = n_t * CB_BN;
    const int cslabs = p.Cin / CB_BK;
    // tile (image t_n, 8 x 16 block (t_y, t_x)) of this workgroup
    const int tiles_x = (sg.Wo + H3_TW - 1) / H3_TW, tiles_y = (sg.Ho + H3_TH - 1) / H3_TH;
    const int t_loc = m_t - sg.tile_start;
    const int t_n = t_loc / (tiles_x * tiles_y), t_rem = t_loc - t_n * (tiles_x * tiles_y);
    const int t_y = t_rem / tiles_x, t_x = t_rem - t_y * tiles_x;
    const int y0 = H3_TH * t_y - 1, x0 = H3_TW * t_x - 1;       // image position of patch pixel (0, 0)

    // patch loader: element e = tid + 256 j -> patch pixel e >> 3, channels 4 (e & 7) .. +3 of the slab; byte offsets with bit 31 =
    // zero padding / beyond the patch
    // (bf16 inputs: element e -> patch pixel e >> 2, channels 8 (e & 3) .. +7: half as many 16-byte loads)
    constexpr int NLD = IN16 ? H3_LD / 2 : H3_LD;
    unsigned po[H3_LD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = tid + 256 * j, px = IN16 ? e >> 2 : e >> 3;
        const int hy = y0 + px / H3_PW, wx = x0 + px % H3_PW;
        po[j] = 0x80000000u;
        if (px < H3_NPX && hy >= 0 && hy < sg.H && wx >= 0 && wx < sg.W)
            po[j] = XB * (unsigned)(((t_n * sg.H + hy) * sg.W + wx) * p.Cin + (IN16 ? 8 * (e & 3) : 4 * (e & 7)));
    }
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * XB * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)xhi << 32) | xlo), 0, (int)xbytes, 0x00020000);
    // B staging: thread -> 8 consecutive k (one 16-byte octet) of columns bcol and bcol + 64
    const int bcol = tid >> 2, boct = tid & 3;
    const unsigned slab_bytes = (unsigned)p.ldw * CB_BK * 2u;
    const unsigned ob0 = ((unsigned)(n0 + bcol) * CB_BK + 8u * boct) * 2u;
    const unsigned ob64 = 64u * CB_BK * 2u;
    const char *whb = reinterpret_cast<const char *>(whi), *wlb = reinterpret_cast<const char *>(wlo);
    // A fragments: row r = 32 i + aij of this wave's 64 pixels = tile pixel 64 wm + r = (y, x) -> patch pixel (y + ki, x + kj)
    // The 32 rows of an MFMA block are two image rows of 16 pixels. ds_read_b128 is serviced in the fixed lane groups
    // {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS): MFMA row l is therefore mapped to pixel h3_perm(l) such
    // that each group reads 16 CONSECUTIVE pixels of one image row -- with the (pixel >> 2) & 3 octet swizzle that is conflict-free
    // for every tap shift (the identity mapping measured 24 % conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; now 0).
    int pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int q = 64 * wm + 32 * i + h3_perm(aij); pb[i] = (q >> 4) * H3_PW + (q & 15); }

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[H3_LD];
    uint4 rbh0[TPS], rbh1[TPS], rbl0[TPS], rbl1[TPS];

#define H3_LDX(D, O) { const uintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (O), 0, 0); \
        D = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }
#define H3_FETCH_PATCH(CS) { _Pragma("unroll") for (int j = 0; j < NLD; ++j) H3_LDX(ra[j], po[j] + (unsigned)(CS) * (CB_BK * XB)) }
#define H3_STASH_PATCH(BUF)                                                                               \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                 \
            const int e = tid + 256 * j, px = IN16 ? e >> 2 : e >> 3, c4 = e & 7;                         \
            if (IN16) {                                                                                   \
                if (px < H3_NPX) *reinterpret_cast<float4 *>(&Ph[BUF][H3_SW(px, e & 3)]) = ra[j];         \
            } else if (px < H3_NPX) {                                                                     \
                bf16x4 h_, l_;                                                                            \
                cb_split4(ra[j], true, h_, l_);                                                           \
                *reinterpret_cast<bf16x4 *>(&Ph[BUF][H3_SW(px, c4 >> 1) + 4 * (c4 & 1)]) = h_;            \
                if (SPLIT == 3) *reinterpret_cast<bf16x4 *>(&Pl[BUF][H3_SW(px, c4 >> 1) + 4 * (c4 & 1)]) = l_; \
            }                                                                                             \
        }                                                                                                 \
    }
#define H3_FETCH_B(CS, TAP0)                                                                              \
    {                                                                                                     \
        _Pragma("unroll") for (int u = 0; u < TPS; ++u) {                                                 \
            const unsigned ob = ob0 + (unsigned)(((TAP0) + u) * cslabs + (CS)) * slab_bytes;              \
            rbh0[u] = *reinterpret_cast<const uint4 *>(whb + ob);                                         \
            rbh1[u] = *reinterpret_cast<const uint4 *>(whb + ob + ob64);                                  \
            if (SPLIT == 3) { rbl0[u] = *reinterpret_cast<const uint4 *>(wlb + ob); rbl1[u] = *reinterpret_cast<const uint4 *>(wlb + ob + ob64); } \
        }                                                                                                 \
    }
#define H3_STASH_B(BUF)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int u = 0; u < TPS; ++u) {                                                 \
            *reinterpret_cast<uint4 *>(&Bh[BUF][u][bcol][CB_SW(bcol, boct)]) = rbh0[u];                   \
            *reinterpret_cast<uint4 *>(&Bh[BUF][u][bcol + 64][CB_SW(bcol + 64, boct)]) = rbh1[u];         \
            if (SPLIT == 3) {                                                                             \
                *reinterpret_cast<uint4 *>(&Bl[BUF][u][bcol][CB_SW(bcol, boct)]) = rbl0[u];               \
                *reinterpret_cast<uint4 *>(&Bl[BUF][u][bcol + 64][CB_SW(bcol + 64, boct)]) = rbl1[u];     \
            }                                                                                             \
        }                                                                                                 \
    }

    // ---- prologue: patch of slab 0 and the weights of step 0 (taps 0 .. TPS-1)
    H3_FETCH_PATCH(0)
    H3_FETCH_B(0, 0)
    H3_STASH_PATCH(0)
    H3_STASH_B(0)
    __syncthreads();
    int bb = 0;                                  // B buffer of the current step
    // (the steps of a slab are unrolled: the tap shifts are immediates. A variant with a runtime tap loop and the weights of step
    // s+2 prefetched into a second register set was measured slower (250 vs 221 us bf16 / 449 vs 420 us bf16x3 on FPN-P2); with
    // unrolled taps AND two weight sets the kernel needs > 256 registers. PMC on FPN-P2 with one tap per step in bf16: MFMA pipe
    // 31 % busy, LDS 25 %, 0 bank-conflict cycles.)
    constexpr int NST = 9 / TPS;                 // steps per slab
    for (int cs = 0; cs < cslabs; ++cs) {
        const int pbuf = cs & 1;
        const bool more_slabs = cs + 1 < cslabs;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const bool last = !more_slabs && st == NST - 1;
            // loads of the next step's weights; the next slab's patch is fetched at the first step and stashed at tap 5 / row 1
            if (!last) { if (st < NST - 1) H3_FETCH_B(cs, (st + 1) * TPS) else H3_FETCH_B(cs + 1, 0) }
            if (st == 0 && more_slabs) H3_FETCH_PATCH(cs + 1)
#pragma unroll
            for (int u = 0; u < TPS; ++u) {
                const int tap = st * TPS + u;
                const int sh = (tap / 3) * H3_PW + (tap % 3);
#pragma unroll
                for (int t = 0; t < CB_BK / 16; ++t) {
                    const int ko = 2 * t + akr;
                    bf16x8 ah[2], bh[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) ah[i] = *reinterpret_cast<const bf16x8 *>(&Ph[pbuf][H3_SW(pb[i] + sh, ko)]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) bh[j] = *reinterpret_cast<const bf16x8 *>(&Bh[bb][u][wn * 64 + 32 * j + aij][CB_SW(wn * 64 + 32 * j + aij, ko)]);
                    if (SPLIT == 3) {
                        bf16x8 al[2], bl[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) al[i] = *reinterpret_cast<const bf16x8 *>(&Pl[pbuf][H3_SW(pb[i] + sh, ko)]);
#pragma unroll
                        for (int j = 0; j < 2; ++j) bl[j] = *reinterpret_cast<const bf16x8 *>(&Bl[bb][u][wn * 64 + 32 * j + aij][CB_SW(wn * 64 + 32 * j + aij, ko)]);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
            if (!last) H3_STASH_B(bb ^ 1)
            if (st == (TPS == 1 ? 5 : 1) && more_slabs) H3_STASH_PATCH(pbuf ^ 1)
            __syncthreads();
            bb ^= 1;
        }
    }
#undef H3_LDX
#undef H3_FETCH_PATCH
#undef H3_STASH_PATCH
#undef H3_FETCH_B
#undef H3_STASH_B

    // ---- epilogue: + bias, + residual, ReLU; accumulator row -> tile pixel (y, x) -> output pixel (fp32 or bf16 output / residual)
    const bool has_res = sg.res != nullptr, has_bias = p.bias != nullptr;
    const bool res16 = (p.io & 4) != 0;
    const unsigned oelems = (unsigned)sg.M * (unsigned)p.Cout;
    const __amdgpu_buffer_rsrc_t orsrc = cb_rsrc(sg.out, oelems * (OUT16 ? 2u : 4u));
    const __amdgpu_buffer_rsrc_t rrsrc = cb_rsrc(has_res ? sg.res : sg.out, oelems * (res16 ? 2u : 4u));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + 32 * j + aij;
        const bool co_ok = co < p.Cout;
        const float bv = (has_bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned oidx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 64 * wm + 32 * i + h3_perm(4 * akr + (r & 3) + 8 * (r >> 2));
                const int ho = H3_TH * t_y + (q >> 4), wo = H3_TW * t_x + (q & 15);
                oidx[r] = (co_ok && ho < sg.Ho && wo < sg.Wo) ? (unsigned)((t_n * sg.Ho + ho) * sg.Wo + wo) * (unsigned)p.Cout + (unsigned)co : 0x3FFFFFFFu;
            }
            float rr[16];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = cb_load_elem(rrsrc, oidx[r], res16);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                if (has_bias) v = v + bv;
                if (has_res) v = v + rr[r];
                if (p.relu) v = fmaxf(v, 0.f);
                cb_store_elem<OUT16>(orsrc, oidx[r], v);
            }
        }
    }
}

// conv1x1_wreg_bf16.hip
bool conv1x1_wreg_bf16_supported(const ConvParams &p);
int conv1x1_wreg_bf16_launch(hipStream_t st, ConvParams &p, const void *wpack_hi);
// conv3x3_wreg_bf16.hip
bool conv3x3_wreg_bf16_supported(const ConvParams &p);
int conv3x3_wreg_bf16_launch(hipStream_t st, ConvParams &p, const void *wpack_hi);

extern "C" int upsnet_conv2d_nhwc_bf16(void *stream, int nseg, const float *const x[], const float *const residual[], float *const out[],
                                       const int batch[], const int height[], const int width[], int Cin, const void *wpack_hi,
                                       const void *wpack_lo, int ldw, const float *bias, int Cout, int KH, int KW, int stride, int pad,
                                       int relu)
{
    ConvParams p;
    // relu: bit 0 = ReLU, bit 1 = the residual is at half resolution (nearest x2 upsampled add), bits 2 / 3 / 4 = the input / the
    // output / the residual tensors are bf16 (NHWC, same shapes) instead of fp32 -- bf16 mode only (wpack_lo == NULL)
    const int res_up = (relu >> 1) & 1, io = (relu >> 2) & 7;
    relu &= 1;
    int rc = conv_fill(p, "conv2d_nhwc_bf16", nseg, x, residual, nullptr, nullptr, out, batch, height, width, Cin, Cout,
                       reinterpret_cast<const float *>(wpack_hi), ldw, bias, KH, KW, stride, pad, 1, relu);
    if (rc) return rc;
    if (res_up) {
        UPS_REQUIRE(residual && KH == 1 && KW == 1, "conv2d_nhwc_bf16: the upsampled residual needs a residual and a 1x1 kernel");
        for (int i = 0; i < p.nseg; ++i)
            UPS_REQUIRE(p.seg[i].Ho % 2 == 0 && p.seg[i].Wo % 2 == 0, "conv2d_nhwc_bf16: the upsampled residual needs even output dims");
        p.res_up = 1;
    }
    UPS_REQUIRE(ldw % CB_BN == 0, "conv2d_nhwc_bf16: ldw must be a multiple of %d (got %d)", CB_BN, ldw);
    UPS_REQUIRE(io == 0 || wpack_lo == nullptr, "conv2d_nhwc_bf16: bf16 tensors only in the plain bf16 mode (no low weight part)");
    p.io = ((io & 1) ? 1 : 0) | ((io & 2) ? 2 : 0) | ((io & 4) ? 4 : 0);
    for (int i = 0; i < p.nseg; ++i)
        UPS_REQUIRE(p.seg[i].M * Cout < (1L << 29), "conv2d_nhwc_bf16: output %d exceeds 2 GiB; split the batch", i);
    UPS_REQUIRE(KH * KW <= 9, "conv2d_nhwc_bf16: at most 9 taps");
    for (int i = 0; i < p.nseg; ++i)   // bit 31 of a pixel offset flags the zero padding
        UPS_REQUIRE((long)p.seg[i].N * p.seg[i].H * p.seg[i].W * Cin < (1L << 29), "conv2d_nhwc_bf16: feature map %d exceeds 2 GiB; split the batch", i);
    const __bf16 *hi = reinterpret_cast<const __bf16 *>(wpack_hi), *lo = reinterpret_cast<const __bf16 *>(wpack_lo);
    static const bool no_halo = getenv("UPSNET_BF16_HALO") != nullptr && getenv("UPSNET_BF16_HALO")[0] == '0';
    if (!lo && KH == 1 && KW == 1 && pad == 0 && conv1x1_wreg_bf16_supported(p))   // bf16 activations: both operands from global memory
        return conv1x1_wreg_bf16_launch((hipStream_t)stream, p, wpack_hi);
    if (!lo && KH == 3 && KW == 3 && stride == 1 && pad == 1 && conv3x3_wreg_bf16_supported(p))   // 256 -> 256 layers: weights from L2
        return conv3x3_wreg_bf16_launch((hipStream_t)stream, p, wpack_hi);
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && !no_halo) {   // haloed-patch kernel: 8 x 16 pixel tiles
        int t3 = 0;
        for (int i = 0; i < p.nseg; ++i) {
            p.seg[i].tile_start = t3;
            t3 += p.seg[i].N * ((p.seg[i].Ho + H3_TH - 1) / H3_TH) * ((p.seg[i].Wo + H3_TW - 1) / H3_TW);
        }
        p.m_tiles = t3;
        p.n_tiles = ldw / CB_BN;
        const int g3 = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;
        if (lo) hipLaunchKernelGGL((conv3x3_bf16_halo_kernel<3, 1>), dim3(g3), dim3(256), 0, (hipStream_t)stream, p, hi, lo);
        else switch (io & 3) {
            case 0: hipLaunchKernelGGL((conv3x3_bf16_halo_kernel<1, 1, 0>), dim3(g3), dim3(256), 0, (hipStream_t)stream, p, hi, lo); break;
            case 1: hipLaunchKernelGGL((conv3x3_bf16_halo_kernel<1, 1, 1>), dim3(g3), dim3(256), 0, (hipStream_t)stream, p, hi, lo); break;
            case 2: hipLaunchKernelGGL((conv3x3_bf16_halo_kernel<1, 1, 2>), dim3(g3), dim3(256), 0, (hipStream_t)stream, p, hi, lo); break;
            default: hipLaunchKernelGGL((conv3x3_bf16_halo_kernel<1, 1, 3>), dim3(g3), dim3(256), 0, (hipStream_t)stream, p, hi, lo); break;
        }
        UPS_CHECK_LAUNCH("conv3x3_bf16_halo_kernel");
        ups_set_form("conv3x3_halo<%d>", lo ? 3 : 1);
        return 0;
    }
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) { p.seg[i].tile_start = tiles; tiles += (int)((p.seg[i].M + CB_BM - 1) / CB_BM); }
    p.m_tiles = tiles;
    p.n_tiles = ldw / CB_BN;
    // small layers (res4 / res5 / P5 maps: fewer 128 x 128 tiles than CUs): 128 x 64 tiles, twice the workgroups (plain bf16 mode)
    static const int narrow_below = getenv("UPSNET_BF16_NARROW_BELOW") ? atoi(getenv("UPSNET_BF16_NARROW_BELOW")) : 384;
    const bool narrow = !lo && p.m_tiles * p.n_tiles < narrow_below;
    if (narrow) p.n_tiles = ldw / 64;
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;
#define CB_GO(WN_, IO_) hipLaunchKernelGGL((conv_bf16_kernel<1, WN_, IO_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, hi, lo)
    if (lo) hipLaunchKernelGGL((conv_bf16_kernel<3, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, hi, lo);
    else if (narrow) switch (io & 3) { case 0: CB_GO(1, 0); break; case 1: CB_GO(1, 1); break; case 2: CB_GO(1, 2); break; default: CB_GO(1, 3); break; }
    else switch (io & 3) { case 0: CB_GO(2, 0); break; case 1: CB_GO(2, 1); break; case 2: CB_GO(2, 2); break; default: CB_GO(2, 3); break; }
#undef CB_GO
    UPS_CHECK_LAUNCH("conv_bf16_kernel");
    ups_set_form("conv_bf16<%d,%d>", lo ? 3 : 1, narrow ? 1 : 2);
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
