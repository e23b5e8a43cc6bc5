// deform_fused.hip -- deformable convolution v1 / v2 (sampling + GEMM fused, no column buffer), second generation.
//
// Reference: DeformConvFunction.forward = deformable_im2col into a column buffer in HBM + torch.mm + bias
// (upsnet/operators/functions/deform_conv.py:43-57, deform_conv_kernel.cu:88-118,194-242; v2: mod_deform_conv_kernel.cu:187-249).
//
// The first generation (conv.hip, DEFORM = 1 / 2) was the dense implicit-GEMM kernel with a bilinear-gather loader. Profiled on
// MI355X (profiles/r05_*): 58 % of the fp32 MFMA peak, 18-178 registers spilled to scratch, and 2.63 GB of fabric traffic per
// launch for 236 MB of algorithmic bytes -- the K walk went tap by tap over ALL input channels, so the 1 KiB pixel vectors a tap
// touched had left the XCD's 4 MiB L2 (24 MB pass through it per tap) before the neighbouring tap came back to them.
//
// This kernel changes the three things behind those numbers:
//   * K is walked CHANNEL SLAB outermost, tap innermost: for 32 channels (one 128-byte line per pixel) all kh*kw taps follow
//     each other. The corners of neighbouring taps coincide or are adjacent, so a line is re-used out of L1 / L2 eight slabs
//     later at most (working set per workgroup and slab: ~3 rows x 66 pixels x 128 B = 25 KiB).
//   * The per-(pixel, tap) sampling data -- the four corner addresses (out-of-image corners point beyond the buffer, where a
//     buffer load returns the reference's 0), the four bilinear weights (and the v2 modulation), computed with the reference's
//     exact fp32 expressions -- are built ONCE per workgroup into an LDS table (36 B x 64 pixels x taps) instead of living in
//     registers / being recomputed per tap and channel: no spills, no validity selects, and the slab-outer walk costs two
//     ds_read_b128 per pixel and step.
//   * The B operand (weights) never touches LDS: packed in MFMA fragment order it is one 16-byte buffer load per lane and step,
//     prefetched four steps ahead in a register ring (as conv_wino.hip). LDS holds only the blended A tile (2 x 8 KiB) and the
//     table (34 KiB for 3x3), so occupancy is set by registers alone (3 workgroups / CU) and two register sets of corners are in flight: the
//     gather of slab s+2 is issued while slab s+1 is blended and slab s is contracted.
// Tile: 8 x 8 output pixels x 128 output channels per workgroup (4 waves; wave w owns column block w and both 32-row blocks), fp32
// products and accumulation on v_mfma_f32_32x32x2_f32 in a fixed order (bit-repeatable). Epilogue: + bias, ReLU, NHWC store.
#include <stdlib.h>

#include "conv_params.h"
#include "upsnet_hip.h"

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define DF_BM 64
#define DF_BN 128
#define DF_PITCH 65                // 16-byte units per channel quarter of an A buffer: 64 pixels + 1 (odd pitch: the loader's ds_write_b128
                                   // -- 8 lanes = 8 quarters of a pixel -- spreads over the banks, the fragment ds_read_b128 reads consecutive
                                   // pixels anyway, and every fragment address is one lane base + an immediate)
#define DF_ABUF (8 * DF_PITCH)     // float4 units of one A buffer: 8 channel quarters
#define DF_RING 4

// Weights [Cout, Cin, kh, kw] -> fragment order [cb = co/32][cs = c/32][tap][h = (c%32)/8][half = (c%8)/4][co%32][c%4]; the column
// blocks are padded to a multiple of 4 (zero weights) so that every wave of the last 128-channel tile reads valid memory.
__global__ void dcn_pack_weight_frag_kernel(const float *__restrict__ w, int cout, int cin, int taps, int nblk, float *__restrict__ wp)
{
    const long total = (long)nblk * 32 * cin * taps;
    const int cslabs = cin >> 5;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int c4 = idx & 3, col = (idx >> 2) & 31, half = (idx >> 7) & 1, h = (idx >> 8) & 3;
        long r = idx >> 10;
        const int tap = r % taps; r /= taps;
        const int cs = r % cslabs; const int cb = r / cslabs;
        const int co = 32 * cb + col, c = 32 * cs + 8 * h + 4 * half + c4;
        wp[idx] = co < cout ? w[((long)co * cin + c) * taps + tap] : 0.f;
    }
}

extern "C" size_t upsnet_dcn_packed_weight_floats(int cout, int cin, int kh, int kw)
{
    if (cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return 0;
    return (size_t)((cout + DF_BN - 1) / DF_BN) * DF_BN * (size_t)cin * kh * kw;
}

extern "C" int upsnet_dcn_pack_weight(void *stream, const float *weight, int cout, int cin, int kh, int kw, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cin > 0 && cin % 32 == 0 && kh > 0 && kw > 0, "dcn_pack_weight: bad args (Cin %% 32 must be 0)");
    const int nblk = (cout + DF_BN - 1) / DF_BN * 4;
    const long total = (long)nblk * 32 * cin * kh * kw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(dcn_pack_weight_frag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, kh * kw, nblk, wpack);
    UPS_CHECK_LAUNCH("dcn_pack_weight_frag_kernel");
    return 0;
}

// SETS: corner register sets in flight (2: the gather of step s+2 overlaps the blend of step s+1; 1: one step of lookahead;
// 3: one set, "early" schedule -- see DF_STEP_EARLY); WPE: waves per SIMD the register budget is set for.
template <bool MOD, int SETS, int WPE>
__global__ void __launch_bounds__(256, WPE) dcn_fused_f32_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4 *As = reinterpret_cast<float4 *>(smem_raw);                       // [2][8 q][DF_PITCH]
    // sampling table [tap][64 px]: byte offsets of the 4 corners' channel vectors (bit 31 set = outside the image: the buffer load
    // then returns 0, the value the reference substitutes), the 4 bilinear weights, v2: the modulation
    uintx4 *dsc_o = reinterpret_cast<uintx4 *>(smem_raw + 2 * DF_ABUF * 16);
    float4 *dsc_w = reinterpret_cast<float4 *>(dsc_o + p.KH * p.KW * DF_BM);
    float *dsc_m = reinterpret_cast<float *>(dsc_w + p.KH * p.KW * DF_BM);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lhalf = lane >> 5, l32 = lane & 31;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): each XCD gets a contiguous range of m-tiles (vertically adjacent
    // tiles share their sampling halos in that XCD's L2) and all n-tiles of an m-tile. Same scheme as conv_igemm_f32_kernel.
    // Split-K (p.ksplit > 1, single small maps: the backbone's DCN bottlenecks have 96-273 tiles for 256 CUs): workgroup kz walks
    // its share of the (channel slab, tap) steps and writes raw partial sums; conv_splitk_reduce adds bias / ReLU in a fixed order.
    int m_t, n_t, kz;
    {
        const int nt = p.n_tiles;
        const int per = (p.m_tiles + 7) >> 3;
        const int base_grid = 8 * per * nt;
        kz = (int)blockIdx.x / base_grid;
        const int bid = (int)blockIdx.x - kz * base_grid;
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
    // 2-D pixel tile: 8 x 8 output pixels (pixel px of the tile = (px >> 3, px & 7)). The samples of a tile then cover
    // ~(8 + 2r)^2 input pixels for offsets of radius r instead of ~(64 + 2r) x (3 + 2r) for a 64 x 1 row segment: at r = 3 that is
    // 25 KiB instead of 80 KiB per 32-channel slab, so the 96 workgroups of an XCD keep their corner lines in its 4 MiB L2.
    const int tiles_x = (sg.Wo + 7) >> 3, tiles_y = (sg.Ho + 7) >> 3;
    const int t_loc = m_t - sg.tile_start;
    const int t_n = t_loc / (tiles_x * tiles_y), t_rem = t_loc - t_n * (tiles_x * tiles_y);
    const int t_y = t_rem / tiles_x, t_x = t_rem - t_y * tiles_x;
    const int ntap = p.KH * p.KW;
    const int cslabs = p.Cin >> 5;
    const int nsl_all = cslabs * ntap;             // (channel slab, tap) steps of the K walk, tap innermost
    const int s_per = (nsl_all + p.ksplit - 1) / p.ksplit;
    const int s_begin = kz * s_per;
    const int nsl = min(nsl_all - s_begin, s_per);  // steps of this workgroup (the launcher guarantees >= 1)

    // ---- sampling table of this tile, once: deform_conv_kernel.cu:227-240 (positions), :88-118 (corners, weights -- the same fp32
    // expressions, evaluated here instead of per channel)
    {
        const unsigned cin4_ = 4u * (unsigned)p.Cin;
        for (int idx = tid; idx < ntap * DF_BM; idx += 256) {
            const int tap = idx >> 6, px = idx & 63;
            const int ho = 8 * t_y + (px >> 3), wo = 8 * t_x + (px & 7);
            const bool inside = ho < sg.Ho && wo < sg.Wo;
            const long pp = ((long)t_n * sg.Ho + ho) * sg.Wo + wo;
            uintx4 o;
            o.x = o.y = o.z = o.w = 0x80000000u;
            float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
            float m = 1.0f;
            if (inside) {
                const int n = t_n;
                const int ki = tap / p.KW, kj = tap - ki * p.KW;
                const int h_base = ho * p.stride - p.pad + ki * p.dil, w_base = wo * p.stride - p.pad + kj * p.dil;
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
                    wt = make_float4(hh * hw, hh * lw, lh * hw, lh * lw);
                    const bool a = h_low >= 0, b = h_high <= H - 1, c = w_low >= 0, e = w_high <= W - 1;
                    const unsigned base = (unsigned)(n * H * W) * cin4_;
                    if (a && c) o.x = base + (unsigned)(h_low * W + w_low) * cin4_;
                    if (a && e) o.y = base + (unsigned)(h_low * W + w_high) * cin4_;
                    if (b && c) o.z = base + (unsigned)(h_high * W + w_low) * cin4_;
                    if (b && e) o.w = base + (unsigned)(h_high * W + w_high) * cin4_;
                }
                if (MOD) m = sg.mask[pp * ntap + tap];
            }
            dsc_o[idx] = o;
            dsc_w[idx] = wt;
            if (MOD) dsc_m[idx] = m;
        }
    }

    // ---- loader geometry: thread = (pixel prow [+32], channel quarter q of the slab)
    const int q = tid & 7, prow = tid >> 3;
    // corners are fetched with buffer loads: uniform resource (base, size) + one 32-bit byte offset per lane -- no 64-bit address
    // VALU, and (unlike flat loads) nothing that the LDS wait counter has to wait for
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * 4u * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)xhi << 32) | xlo), 0, (int)xbytes, 0x00020000);
    const unsigned st0 = (unsigned)(q * DF_PITCH + prow), st1 = st0 + 32u;
    // fragment units of step h: quarter 2h + lhalf, rows l32 (block 0) and 32 + l32 (block 1)
    // B: lane's float4 of global step g = 4 s + h sits at wbase + g * 1024 + lhalf * 512 + l32 * 16
    const int cb = 4 * n_t + wave;
    const size_t waddr = reinterpret_cast<size_t>(p.w) + (size_t)cb * (size_t)nsl_all * 4096u;
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)waddr), whi = __builtin_amdgcn_readfirstlane((unsigned)(waddr >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)whi << 32) | wlo), 0, nsl_all * 4096, 0x00020000);
    const unsigned b_lane = (unsigned)(lhalf * 512 + l32 * 16);
    const int gmax = nsl_all * 4 - 1;

    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float4 xc00, xc01, xc02, xc03, xc10, xc11, xc12, xc13;   // register set X: [pixel][corner]
    float4 yc00, yc01, yc02, yc03, yc10, yc11, yc12, yc13;   // register set Y
    float4 breg[DF_RING];
    int f_cs = s_begin / ntap, f_tap = s_begin - (s_begin / ntap) * ntap;   // (channel slab, tap) of the NEXT step to fetch
    unsigned f_kill = 0u;     // early schedule: 0x80000000 once every step of this workgroup is fetched (the gather then reads nothing)
    int f_left = nsl;

#define DF_LDX(D, O) { const uintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (O), 0, 0); \
        D = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }
#define DF_FETCH_PX(P, R, O)                                                                                           \
    {                                                                                                                  \
        const unsigned c_ = ((unsigned)(f_cs * 128) | f_kill) + (unsigned)(q * 16);                                    \
        DF_LDX(P##c##R##0, (O).x + c_) DF_LDX(P##c##R##1, (O).y + c_) DF_LDX(P##c##R##2, (O).z + c_) DF_LDX(P##c##R##3, (O).w + c_) \
    }
    // gather of the step whose corner offsets were pre-read into no0 / no1; then pre-read the offsets of the step after it
#define DF_FETCH(P)                                                                                                    \
    {                                                                                                                  \
        DF_FETCH_PX(P, 0, no0) DF_FETCH_PX(P, 1, no1)                                                                  \
        if (++f_tap == ntap) { f_tap = 0; ++f_cs; }                                                                    \
    }
#define DF_NEXT_OFFSETS { no0 = dsc_o[f_tap * DF_BM + prow]; no1 = dsc_o[f_tap * DF_BM + prow + 32]; }
    // blend pixel R of register set P (sampled for tap TAP) and write its 4-channel unit into A buffer BUF
#define DF_STASH_PX(P, R, PX, UNIT, TAP, BUF)                                                                          \
    {                                                                                                                  \
        const float4 w_ = dsc_w[(TAP) * DF_BM + (PX)];                                                                 \
        float4 v_;                                                                                                     \
        v_.x = ((w_.x * (P##c##R##0).x + w_.y * (P##c##R##1).x) + w_.z * (P##c##R##2).x) + w_.w * (P##c##R##3).x;      \
        v_.y = ((w_.x * (P##c##R##0).y + w_.y * (P##c##R##1).y) + w_.z * (P##c##R##2).y) + w_.w * (P##c##R##3).y;      \
        v_.z = ((w_.x * (P##c##R##0).z + w_.y * (P##c##R##1).z) + w_.z * (P##c##R##2).z) + w_.w * (P##c##R##3).z;      \
        v_.w = ((w_.x * (P##c##R##0).w + w_.y * (P##c##R##1).w) + w_.z * (P##c##R##2).w) + w_.w * (P##c##R##3).w;      \
        if (MOD) { const float m_ = dsc_m[(TAP) * DF_BM + (PX)]; v_.x = v_.x * m_; v_.y = v_.y * m_; v_.z = v_.z * m_; v_.w = v_.w * m_; } \
        As[(BUF) * DF_ABUF + (UNIT)] = v_;                                                                             \
    }
#define DF_BLOAD(SLOT, G) { const uintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_lane, (unsigned)min((G), gmax) * 1024u, 0); \
        breg[SLOT] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }
#define DF_FRAG(BUF, H, A0, A1)                                                                                        \
    {                                                                                                                  \
        const int q_ = 2 * (H) + lhalf;                                                                                \
        A0 = As[(BUF) * DF_ABUF + q_ * DF_PITCH + l32];                                                                \
        A1 = As[(BUF) * DF_ABUF + q_ * DF_PITCH + 32 + l32];                                                           \
    }
    // One K step s (buffer BUF holds its blended A tile). Tap index of step s+1 (the one being stashed) is passed as STAP.
    //   u = 0: gather of step s+2 into FSET;  u = 1, 2: blend + stash of step s+1 (SSET) into the other buffer;
    //   u = 3: barrier, then the first fragments of step s+1;  every u: ring refill, next fragments, 8 MFMAs.
#define DF_STEP(BUF, FSET, SSET, DO_FETCH, DO_STASH, STAP)                                                             \
    {                                                                                                                  \
        const bool do_fetch_ = (DO_FETCH), do_stash_ = (DO_STASH);                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                \
            float4 n0_, n1_;                                                                                           \
            if (u == 0 && do_fetch_) DF_FETCH(FSET)                                                                    \
            if (u < 3) DF_FRAG(BUF, u + 1, n0_, n1_)                                                                   \
            if (u == 1 && do_stash_) DF_STASH_PX(SSET, 0, prow, st0, STAP, (BUF) ^ 1)                                  \
            if (u == 2 && do_stash_) DF_STASH_PX(SSET, 1, prow + 32, st1, STAP, (BUF) ^ 1)                             \
            if (u == 3) { __syncthreads(); DF_FRAG((BUF) ^ 1, 0, n0_, n1_) }                                           \
            const float4 bf_ = breg[u];                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bf_.x, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bf_.x, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bf_.y, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bf_.y, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bf_.z, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bf_.z, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bf_.w, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bf_.w, acc1, 0, 0, 0);                                   \
            DF_BLOAD(u, g + 4 + u)                                                                                     \
            if (u == 3 && do_fetch_) DF_NEXT_OFFSETS                                                                   \
            a0 = n0_; a1 = n1_;                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                              \
        g += 4;                                                                                                        \
    }

    // Early schedule (SETS == 3, one register set), per pixel of the thread: blend + stash of step s+1 (pixel 0 at u = 0, pixel 1
    // at u = 1), then -- one sub-step later, into the registers just freed -- its gather of step s+2 (u = 1, u = 2): a gather has
    // three sub-steps (24 MFMAs of its wave) to arrive instead of one or two. Everything in the loop is unconditional: past the end
    // the gather runs with out-of-range offsets (f_kill: no memory access) and the stash fills the buffer nobody reads. (With the
    // conditional form the compiler placed register copies, and with them the wait for the gather, directly behind the loads.)
#define DF_STEP_EARLY(BUF, STAP)                                                                                       \
    {                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                \
            float4 n0_, n1_;                                                                                           \
            if (u < 3) DF_FRAG(BUF, u + 1, n0_, n1_)                                                                   \
            if (u == 0) DF_STASH_PX(x, 0, prow, st0, STAP, (BUF) ^ 1)                                                  \
            if (u == 1) {                                                                                              \
                f_kill = f_left > 0 ? 0u : 0x80000000u; --f_left;                                                      \
                DF_FETCH_PX(x, 0, no0)                                                                                 \
                DF_STASH_PX(x, 1, prow + 32, st1, STAP, (BUF) ^ 1)                                                     \
            }                                                                                                          \
            if (u == 2) {                                                                                              \
                DF_FETCH_PX(x, 1, no1)                                                                                 \
                if (++f_tap == ntap) { f_tap = 0; ++f_cs; }                                                            \
                DF_NEXT_OFFSETS                                                                                        \
            }                                                                                                          \
            if (u == 3) { __syncthreads(); DF_FRAG((BUF) ^ 1, 0, n0_, n1_) }                                           \
            const float4 bf_ = breg[u];                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bf_.x, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bf_.x, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bf_.y, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bf_.y, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bf_.z, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bf_.z, acc1, 0, 0, 0);                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bf_.w, acc0, 0, 0, 0);                                   \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bf_.w, acc1, 0, 0, 0);                                   \
            DF_BLOAD(u, g + 4 + u)                                                                                     \
            a0 = n0_; a1 = n1_;                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                              \
        g += 4;                                                                                                        \
    }

    __syncthreads();   // descriptor table complete
    // ---- prologue: step 0 -> buffer 0 (set X), gather of step 1 in flight (set Y), first ring of B fragments
    uintx4 no0, no1;                                           // corner offsets of the next step to fetch (pre-read from the table)
    DF_NEXT_OFFSETS
    --f_left;
    DF_FETCH(x)
    const int tap0 = s_begin - (s_begin / ntap) * ntap;       // tap of this workgroup's first step
    if (SETS != 3) {
#pragma unroll
        for (int u = 0; u < DF_RING; ++u) DF_BLOAD(u, 4 * s_begin + u)
    }
    DF_NEXT_OFFSETS
    if (SETS == 2 && nsl > 1) { DF_FETCH(y) DF_NEXT_OFFSETS }
    DF_STASH_PX(x, 0, prow, st0, tap0, 0)
    DF_STASH_PX(x, 1, prow + 32, st1, tap0, 0)
    if (SETS == 3) {
        // step 1 in flight across the barrier, and the first B ring, issued in the order of a steady-state step (B, gather of pixel
        // 0, B, gather of pixel 1, B, B) so that the loop's wait counts -- the merge of this entry state and the back edge -- are
        // the steady-state ones
        f_kill = f_left > 0 ? 0u : 0x80000000u; --f_left;
        __builtin_amdgcn_sched_barrier(0);
        DF_BLOAD(0, 4 * s_begin)
        __builtin_amdgcn_sched_barrier(0);
        DF_FETCH_PX(x, 0, no0)
        __builtin_amdgcn_sched_barrier(0);
        DF_BLOAD(1, 4 * s_begin + 1)
        __builtin_amdgcn_sched_barrier(0);
        DF_FETCH_PX(x, 1, no1)
        if (++f_tap == ntap) { f_tap = 0; ++f_cs; }
        DF_NEXT_OFFSETS
        __builtin_amdgcn_sched_barrier(0);
        DF_BLOAD(2, 4 * s_begin + 2)
        DF_BLOAD(3, 4 * s_begin + 3)
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    float4 a0, a1;
    DF_FRAG(0, 0, a0, a1)
    int g = 4 * s_begin;
    int s_tap = tap0 + 1 == ntap ? 0 : tap0 + 1;   // tap of step s+1
    if (SETS == 3) {
        for (int s = 0; s < nsl; ++s) {      // one step per iteration (buffer parity at run time): nothing but the back edge
            const int cur = s & 1;             // carries the in-flight gather, so it stays in the registers it was loaded into
            DF_STEP_EARLY(cur, s_tap)
            if (++s_tap == ntap) s_tap = 0;
        }
    } else if (SETS == 2) {
        for (int s = 0; s < nsl; s += 2) {
            DF_STEP(0, x, y, s + 2 < nsl, s + 1 < nsl, s_tap)
            if (++s_tap == ntap) s_tap = 0;
            if (s + 1 >= nsl) break;
            DF_STEP(1, y, x, s + 3 < nsl, s + 2 < nsl, s_tap)
            if (++s_tap == ntap) s_tap = 0;
        }
    } else {
        for (int s = 0; s < nsl; s += 2) {
            DF_STEP(0, x, x, s + 1 < nsl, s + 1 < nsl, s_tap)
            if (++s_tap == ntap) s_tap = 0;
            if (s + 1 >= nsl) break;
            DF_STEP(1, x, x, s + 2 < nsl, s + 2 < nsl, s_tap)
            if (++s_tap == ntap) s_tap = 0;
        }
    }
#undef DF_LDX
#undef DF_FETCH_PX
#undef DF_FETCH
#undef DF_STASH_PX
#undef DF_BLOAD
#undef DF_FRAG
#undef DF_STEP
#undef DF_STEP_EARLY

    // ---- epilogue: + bias, ReLU, NHWC store. Accumulator element r of lane (lhalf, l32): row 8 (r >> 2) + 4 lhalf + (r & 3),
    // column l32 of the 32x32 block
    const int co = 32 * cb + l32;
    const bool co_ok = co < p.Cout;
    const float bv = (p.bias != nullptr && co_ok) ? p.bias[co] : 0.f;
    // (r08: 32-bit byte offsets through a buffer descriptor; a pixel beyond the map / a padded column is an out-of-range offset)
    const unsigned crow = (unsigned)p.Cout * 4u;
    const bool split = p.ksplit > 1;
    const size_t oaddr = reinterpret_cast<size_t>(split ? (float *)(p.partial + (long)kz * p.m_total * p.Cout) : sg.out);
    const unsigned obytes = __builtin_amdgcn_readfirstlane((unsigned)(split ? p.m_total : (long)sg.N * sg.Ho * sg.Wo) * crow);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(oaddr >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)oaddr)),
        0, (int)obytes, 0x00020000);
    const unsigned tile0 = (unsigned)((t_n * sg.Ho + 8 * t_y) * sg.Wo + 8 * t_x) * crow + 4u * (unsigned)co;   // tile origin + the lane's column
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = 32 * i + 4 * lhalf + (r & 3) + 8 * (r >> 2);       // = tile row 4 i + (r >> 2), column 4 lhalf + (r & 3)
            const int dy = px >> 3, dx = px & 7;
            float v = i == 0 ? acc0[r] : acc1[r];
            if (!split) {         // (split-K: raw partial sums [kz][pixel][Cout]; bias / ReLU in the reduce kernel)
                v = v + bv;
                if (p.relu) v = fmaxf(v, 0.f);
            }
            const bool ok = co_ok && 8 * t_y + dy < sg.Ho && 8 * t_x + dx < sg.Wo;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, ok ? tile0 + (unsigned)(dy * sg.Wo + dx) * crow : 0x80000000u, 0, 0);
        }
    }
}

// development knob for A/B runs: 0 = auto (default: 5), 1 = one corner set at 3 waves / SIMD, 2 = two sets at 2 waves / SIMD,
// 5 = one set, early schedule, at 3 (r07: 859 vs 938 us on the 256 -> 128 layer over four levels, 443 vs 459 on the 128 -> 128
// one). (One set at 4 waves and two sets at 3 need more than their register budget: both spilled inside the loop, 2x slower.)
static int g_dcn_variant = 0;
extern "C" void upsnet_dcn_tuning(int variant) { g_dcn_variant = variant; }

static int dcn_fused_launch(void *stream, int nlev, const float *const x[], const float *const offset[], const float *const mask[],
                            float *const out[], const int height[], const int width[], int cin, int cout, int kh, int kw, int pad, int stride,
                            int dil, const float *wpack, const float *bias, int relu, int ksplit, void *workspace)
{
    UPS_REQUIRE(nlev >= 1 && nlev <= 4 && offset, "deform_conv_fused_nhwc: nlev must be 1..4 and offsets given");
    for (int l = 0; l < nlev; ++l) UPS_REQUIRE(offset[l] && (!mask || mask[l]), "deform_conv_fused_nhwc: null offset/mask at level %d", l);
    UPS_REQUIRE(kh * kw >= 1 && kh * kw <= 25, "deform_conv_fused_nhwc: at most 25 taps (got %dx%d)", kh, kw);
    ConvParams p;
    const int ldw = (cout + 31) / 32 * 32;
    int rc = conv_fill(p, "deform_conv_fused_nhwc", nlev, x, nullptr, offset, mask, out, nullptr, height, width, cin, cout, wpack, ldw, bias,
                       kh, kw, stride, pad, dil, relu);
    if (rc) return rc;
    for (int i = 0; i < p.nseg; ++i) {   // bit 31 of a corner offset flags "outside the image": offsets of real pixels must stay below it
        UPS_REQUIRE((long)p.seg[i].N * p.seg[i].H * p.seg[i].W * cin < (1L << 29), "deform_conv_fused_nhwc: feature map %d exceeds 2 GiB; split the batch", i);
        UPS_REQUIRE((long)p.seg[i].N * p.seg[i].Ho * p.seg[i].Wo * cout < (1L << 29), "deform_conv_fused_nhwc: output %d exceeds 2 GiB; split the batch", i);
    }
    int tiles = 0;
    for (int i = 0; i < p.nseg; ++i) {   // 8 x 8 pixel tiles
        p.seg[i].tile_start = tiles;
        tiles += p.seg[i].N * ((p.seg[i].Ho + 7) / 8) * ((p.seg[i].Wo + 7) / 8);
    }
    p.m_tiles = tiles;
    p.n_tiles = (cout + DF_BN - 1) / DF_BN;
    const int nsl = (cin / 32) * kh * kw;
    if (ksplit > 1) {
        UPS_REQUIRE(nlev == 1 && workspace && ksplit <= 8 && cout % 4 == 0, "deform_conv_fused_nhwc_splitk: one map, a workspace, ksplit <= 8, Cout %% 4 == 0");
        UPS_REQUIRE(((nsl + ksplit - 1) / ksplit) * (ksplit - 1) < nsl, "deform_conv_fused_nhwc_splitk: %d K steps cannot be split %d ways", nsl, ksplit);
        p.ksplit = ksplit;
        p.partial = (float *)workspace;
        p.m_total = p.seg[0].M;
    }
    const size_t smem = (size_t)2 * DF_ABUF * 16 + (size_t)kh * kw * DF_BM * (16 + 16 + 4);
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles * p.ksplit;
    // auto = one corner set, early schedule, at 3 waves per SIMD everywhere. (Two sets -- deeper gather lookahead -- for grids below
    // UPSNET_DCN_SMALL_GRID workgroups was measured on the R101-DCN backbone: 106.6 vs 107.7 img/s, so the default threshold is 0.)
    const int dcn_small = atoi(getenv("UPSNET_DCN_SMALL_GRID") ? getenv("UPSNET_DCN_SMALL_GRID") : "0");
    static const int dcn_auto = atoi(getenv("UPSNET_DCN_VARIANT") ? getenv("UPSNET_DCN_VARIANT") : "5");   // A/B runs of whole models
    const int v = g_dcn_variant ? g_dcn_variant : (grid < dcn_small ? 2 : dcn_auto);
#define DF_LAUNCH(SETS, WPE)                                                                                           \
    if (mask) hipLaunchKernelGGL((dcn_fused_f32_kernel<true, SETS, WPE>), dim3(grid), dim3(256), smem, (hipStream_t)stream, p); \
    else hipLaunchKernelGGL((dcn_fused_f32_kernel<false, SETS, WPE>), dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    if (v == 1) { DF_LAUNCH(1, 3) } else if (v == 2) { DF_LAUNCH(2, 2) } else if (v == 6) { DF_LAUNCH(3, 4) } else { DF_LAUNCH(3, 3) }
#undef DF_LAUNCH
    UPS_CHECK_LAUNCH("dcn_fused_f32_kernel");
    if (p.ksplit > 1)
        return conv_splitk_reduce((hipStream_t)stream, p.partial, p.ksplit, p.m_total, p.seg[0].M, cout, bias, nullptr, relu, out[0]);
    return 0;
}

extern "C" int upsnet_deform_conv_fused_nhwc(void *stream, int nlev, const float *const x[], const float *const offset[],
                                             const float *const mask[], float *const out[], const int height[], const int width[],
                                             int cin, int cout, int kh, int kw, int pad, int stride, int dil, const float *wpack,
                                             const float *bias, int relu)
{
    return dcn_fused_launch(stream, nlev, x, offset, mask, out, height, width, cin, cout, kh, kw, pad, stride, dil, wpack, bias, relu, 1, nullptr);
}

extern "C" size_t upsnet_deform_conv_fused_splitk_workspace_bytes(int height, int width, int cout, int kh, int kw, int pad, int stride,
                                                                   int dil, int ksplit)
{
    const long Ho = (height + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (width + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || ksplit < 1) return 0;
    return (size_t)ksplit * Ho * Wo * cout * sizeof(float);
}

extern "C" int upsnet_deform_conv_fused_nhwc_splitk(void *stream, const float *x, const float *offset, const float *mask, float *out,
                                                    int height, int width, int cin, int cout, int kh, int kw, int pad, int stride, int dil,
                                                    const float *wpack, const float *bias, int relu, int ksplit, void *workspace)
{
    const float *xs[1] = {x}, *os_[1] = {offset}, *ms[1] = {mask};
    float *outs[1] = {out};
    const int hh[1] = {height}, ww[1] = {width};
    return dcn_fused_launch(stream, 1, xs, os_, mask ? ms : nullptr, outs, hh, ww, cin, cout, kh, kw, pad, stride, dil, wpack, bias, relu,
                            ksplit, workspace);
}
