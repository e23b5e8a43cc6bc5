// conv_wino36.hip -- Winograd F(4x4, 3x3) convolution, fp32 MFMA, NHWC, all 36 transform-domain accumulators resident (r11).
//
// The 3x3 / stride 1 / pad 1 nn.Conv2d (+ folded BN, bias, ReLU) layers of the reference's ResNet-FPN on their big maps
// (upsnet/models/resnet.py:64-77 conv2, fpn.py:60-98 output convolutions, rpn.py:34-47 the RPN convolution, rcnn.py:96-116 the mask head):
// the same contract as conv_wino.hip (F(2x2, 3x3), 16 multiplies per 4 outputs = 4.0 per output), with 4x4 output tiles: 36 multiplies
// per 16 outputs = 2.25 per output, 0.5625 of the F(2x2) kernel's matrix work and a quarter of the direct form's.
//
// Numerics. Interpolation points {0, 3/4, -3/4, 3/2, -3/2, inf} (Cook-Toom; tools/winograd_error_cpu.py): every entry of B^T and A^T
// is a dyadic rational (exact in fp32), G is applied once at pack time in double precision. Worst error / (1e-4 + 1e-4 |ref|) on a
// 256 -> 256 layer with unit-variance activations (host emulation, fp32 roundings at every stage): 0.054, against 0.013 for F(2x2), 0.097
// for the asymmetric set {0, 1, -1, 1/2, -2, inf} this kernel started with and 0.23-0.25 for the textbook sets {0, +-1, +-2, inf} /
// {0, +-1, +-1/2, inf}; on the GPU (tools/bench_winograd36.py, the model's shapes) 0.07-0.15 against 0.014-0.032 for the F(2x2) kernel.
// (The r03-r05 rejection of F(4x4) -- "12x the rounding error" -- was measured with the textbook points on the uncalibrated model, whose
// activations were 50-140.) A symmetric set also halves the transforms' arithmetic: the +-a rows share their even / odd parts (12 FMAs per
// 1-D pass of six values instead of 16).
//
// GEMM view: a row is one 4x4 OUTPUT TILE (its 6x6 input patch d), a column one output channel; for each of the 36 positions
// xi = (i, j): M_xi = V_xi x U_xi with V = B^T d B, U = G g G^T. A workgroup (8 waves, two per SIMD, 256 registers each) owns 32 tiles
// (512 output pixels) x 64 channels: wave (b, g) holds the 32x32 block of column block b for the NINE positions [9 g, 9 g + 9) = 144
// accumulator registers. Per slab of 16 input channels: thread (tile, channel) -- all 512 threads -- loads the tile's 36 patch pixels,
// transforms them and writes V_xi to LDS as 16-byte units [xi][q = c / 4][tile ^ swz(q)], swz = {0, 4, 2, 6} (conflict-free for the 4-byte stash -- the
// 32 lanes of a half wave are 2 tiles x 16 channels: tile bit 0, both bits of q and the channel's low bits select 32 different banks -- and for the
// ds_read_b128 fragment reads), double-buffered, one barrier per slab; the loads, the row pass and the column pass + stash of the NEXT
// slab are spread over the 18 steps (4 MFMAs each) of the current one. The MFMA is v_mfma_f32_32x32x2_f32: lane l supplies
// A[row l % 32][k l / 32], so a float4 fragment (4 channels of a tile) feeds four MFMAs. The B operand (U) does not go through LDS:
// packed as [n-tile][slab][xi][q][64 channels][4] so that a lane's fragment is one 16-byte load, contiguous across the wave, prefetched
// in a ring of three registers.
// What bounds the slab loop (tools/ubench/mfma_valu.hip, profiles/r12_mfma_valu.txt): on gfx950 the fp32 MFMA does not overlap with VALU
// instructions -- each costs 2.8-5 cycles of matrix-pipe time per wave wherever it is placed, in front of a dependent MFMA chain or
// inside its gaps -- so the loop's time is (MFMA time) + (VALU, LDS-write and load issue). Ablations on FPN P2 (410 us): transforms 34 us,
// stash 30 us, B loads 15 us, patch loads 9 us, prologue + epilogue ~50 us, matrix work 246 us. Hence: 12-FMA transforms, no per-load
// address arithmetic (below), 152 VALU instructions per wave and slab against 72 MFMAs (r11 first version: 278).
// Output transform Y = A^T M A: the 36 values of a (tile, channel) live in four waves, so the accumulators go through LDS once
// ([element r][xi][lane], one column block at a time: 144 KiB), and wave w finishes elements r = 2 w, 2 w + 1 -- both 1-D passes, bias,
// ReLU, 16 stores of 128 contiguous bytes per wave.
// LDS: max(2 x 72 KiB double-buffered V, 144 KiB exchange), one workgroup per CU.
#include <cstdlib>
#include <type_traits>
#include "conv_params.h"
#include "upsnet_hip.h"

#define W36_TM 32            // 4x4-output tiles per workgroup
#define W36_TN 64            // output channels per workgroup
#define W36_PTS 36
#define W36_VBUF (W36_PTS * 4 * W36_TM * 16)      // bytes of one V buffer: 36 xi x 4 q x 32 tiles x 16 B = 73728
#define W36_BSTEP (2 * W36_TN * 16)               // bytes of one step (position, k half) of the packed weights: 2 q x 64 channels x 16 B
#define W36_STEPS (2 * W36_PTS)                   // steps per slab of 16 input channels; 18 per wave
#define W36_XCH (16 * W36_PTS * 64 * 4)           // bytes of the accumulator exchange of one column block: 147456
#define W36_RING 3          // B fragments in flight per wave; divides the 18 steps of a slab, so a step's ring slot is the same in every slab

typedef unsigned w36_uintx4 __attribute__((ext_vector_type(4)));

// B^T d (one 1-D pass of the input transform), points {0, 3/4, -3/4, 3/2, -3/2, inf}. The +- pairs share their even and odd parts:
//   r0 = 81/64 d0 - 45/16 d2 + d4                          r5 = 81/64 d1 - 45/16 d3 + d5
//   r1, r2 = (d4 - 9/4 d2) +- 3/4 (d3 - 9/4 d1)            r3, r4 = (d4 - 9/16 d2) +- 3/2 (d3 - 9/16 d1)
// twelve FMAs per pass of six values (every constant is exact in fp32).
#define W36_BT(D0, D1, D2, D3, D4, D5)                                                                               \
    {                                                                                                                \
        const float a0_ = D0, a1_ = D1, a2_ = D2, a3_ = D3, a4_ = D4, a5_ = D5;                                       \
        const float e1_ = __builtin_fmaf(-2.25f, a2_, a4_), o1_ = __builtin_fmaf(-2.25f, a1_, a3_);                   \
        const float e3_ = __builtin_fmaf(-0.5625f, a2_, a4_), o3_ = __builtin_fmaf(-0.5625f, a1_, a3_);               \
        D0 = __builtin_fmaf(1.265625f, a0_, __builtin_fmaf(-2.8125f, a2_, a4_));                                     \
        D1 = __builtin_fmaf(0.75f, o1_, e1_);                                                                        \
        D2 = __builtin_fmaf(-0.75f, o1_, e1_);                                                                       \
        D3 = __builtin_fmaf(1.5f, o3_, e3_);                                                                         \
        D4 = __builtin_fmaf(-1.5f, o3_, e3_);                                                                        \
        D5 = __builtin_fmaf(1.265625f, a1_, __builtin_fmaf(-2.8125f, a3_, a5_));                                     \
    }
// A^T m (one 1-D pass of the output transform): y0 = m0 + (m1 + m2) + (m3 + m4), y1 = 3/4 (m1 - m2) + 3/2 (m3 - m4),
//   y2 = 9/16 (m1 + m2) + 9/4 (m3 + m4), y3 = 27/64 (m1 - m2) + 27/8 (m3 - m4) + m5
#define W36_AT(M0, M1, M2, M3, M4, M5, Y0, Y1, Y2, Y3)                                                               \
    {                                                                                                                \
        const float s1_ = (M1) + (M2), d1_ = (M1) - (M2), s2_ = (M3) + (M4), d2_ = (M3) - (M4);                      \
        Y0 = ((M0) + s1_) + s2_;                                                                                     \
        Y1 = __builtin_fmaf(1.5f, d2_, 0.75f * d1_);                                                                 \
        Y2 = __builtin_fmaf(2.25f, s2_, 0.5625f * s1_);                                                              \
        Y3 = __builtin_fmaf(3.375f, d2_, __builtin_fmaf(0.421875f, d1_, (M5)));                                      \
    }

// SK (r13): split-K instance for single maps with fewer workgroups than CUs (res3 / res4 conv2, FPN P4 at 1024x2048: 64-128 workgroups): workgroup
// (tile, kz) walks slabs [kz, kz + 1) * nslabs / ksplit and stores the output transform of its PARTIAL M (the transform is linear) without bias
// / ReLU into p.partial[kz]; conv_splitk_reduce adds the partials in a fixed order, + bias, ReLU. The unsplit instance is the r12 kernel.
template <bool SK>
__global__ void __launch_bounds__(512, 1) conv_wino36_f32_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave >> 2, pg = wave & 3;             // column block (32 channels), position group (9 positions)
    const int lhalf = lane >> 5, l32 = lane & 31;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): each XCD gets a contiguous range of m-tiles, and all n-tiles of an
    // m-tile (they share the input patches) stay on that XCD. Same scheme as conv_wino16_f32_kernel.
    int m_t, n_t, kz = 0;
    {
        const int nt = p.n_tiles;
        const int per = (p.m_tiles + 7) >> 3;
        int bid = (int)blockIdx.x;
        if constexpr (SK) { const int base_grid = 8 * per * nt; kz = bid / base_grid; bid -= kz * base_grid; }
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
    const long p0 = (long)(m_t - sg.tile_start) * W36_TM;
    const int nslabs_all = p.Cin >> 4;
    // this workgroup's slabs [s_first, s_first + nslabs): all of them, or (SK) the kz-th share (launcher: never empty)
    const int s_first = SK ? (kz * nslabs_all) / p.ksplit : 0;
    const int nslabs = SK ? ((kz + 1) * nslabs_all) / p.ksplit - s_first : nslabs_all;
    const long HoWo = (long)sg.Ho * sg.Wo;               // tiles per image (Ho, Wo count 4x4 output tiles here)

    // ---- loader geometry: thread = (tile tid / 16, channel tid % 16 of the slab)
    const int ltile = tid >> 4, lch = tid & 15;
    // Every vector instruction in the slab loop costs matrix-pipe time (tools/ubench/mfma_valu.hip: the fp32 MFMA and the VALU do not
    // overlap on gfx950 -- 2.8 to 5 cycles per instruction per wave, wherever it is placed), so the loop is written for the fewest VALU
    // instructions. Two forms of the patch loads:
    //  * FAST (every workgroup but the few at the end of a map): a load's address is a per-lane origin + a wave-uniform (row, column)
    //    offset in an SGPR -- no per-load address arithmetic. A patch pixel outside the image is NOT kept out of the load (it returns some
    //    other pixel of the map) but zeroed afterwards, by the waves that have such pixels (tiles along the map's border: `fix`,
    //    wave-uniform), from two 6-bit masks of invalid patch rows / columns. The bounds check of a buffer load covers the VECTOR offset
    //    only, so (a) a "negative" origin (pixel (-1, w) or (h, -1) of the first image) would read 0 whatever the scalar offset adds: the
    //    patch is addressed from four origins -- (h0, w0) for element (0, 0), (h0, w0 + 1) for the rest of row 0, (h0 + 1, w0) for the rest
    //    of column 0, (h0 + 1, w0 + 1) for the other 25 -- each non-negative wherever one of its elements lies inside the image; and (b)
    //    origin + scalar offset may point up to 4 rows + 4 pixels beyond the element's own image: harmless inside the tensor (the next
    //    image), NOT allowed beyond its end -- so
    //  * SAFE (workgroups whose last tile could reach beyond the end of the tensor, `risky` -- the last 1-2 tile rows of the last image): the r10 form, every load with its own
    //    per-lane offset, rows / columns outside the image flagged beyond the bounds (no access, reads 0). 16 of FPN P2's 1024 workgroups.
    int hrow0, wcol0;            // n * H + h0 (h0 = 4 ty - 1), w0 = 4 tx - 1
    int hlo, hhi;                // valid range of hrow0 + r: [n * H, n * H + H) (empty for a tile beyond the map: every row reads 0)
    unsigned tbase, nrmask = 0, ncmask = 0;
    bool lane_fix;
    {
        const long pp = p0 + ltile;
        const bool tile_ok = pp < sg.M;                       // (a tile beyond the map is computed as the last tile; never stored)
        const long ppc = tile_ok ? pp : sg.M - 1;
        const int n = (int)(ppc / HoWo);
        const int rem = (int)(ppc - (long)n * HoWo);
        const int ty = rem / sg.Wo, tx = rem - ty * sg.Wo;
        const int h0 = 4 * ty - 1;
        hlo = n * sg.H; hhi = tile_ok ? hlo + sg.H : hlo;
        hrow0 = hlo + h0;
        wcol0 = 4 * tx - 1;
#pragma unroll
        for (int r = 0; r < 6; ++r) nrmask |= (h0 + r >= 0 && h0 + r < sg.H) ? 0u : (1u << r);
#pragma unroll
        for (int c = 0; c < 6; ++c) ncmask |= (wcol0 + c >= 0 && wcol0 + c < sg.W) ? 0u : (1u << c);
        lane_fix = tile_ok && (nrmask | ncmask) != 0;
        tbase = (unsigned)(hrow0 * sg.W + wcol0) * (4u * (unsigned)p.Cin) + 4u * (unsigned)lch;
    }
    const bool fix = __builtin_amdgcn_ballot_w64(lane_fix) != 0ull;
    // workgroup-uniform, from the real extent (r13; the r12 rule "last two tile rows of the last image" assumed H > 3): the FAST loads of a
    // tile (n, ty, tx) reach pixel (4 ty + 4, 4 tx + 4) of image n -- up to 4 pixels into map row n H + 4 ty + 5 -- and tiles are ordered
    // by (n, ty, tx), so the workgroup's LAST tile decides: that row must exist inside the tensor
    bool risky;
    {
        const long plast = min(p0 + (long)W36_TM, sg.M) - 1;
        const long nl = plast / HoWo;
        const long tyl = (plast - nl * HoWo) / sg.Wo;
        risky = nl * sg.H + 4 * tyl + 5 >= (long)sg.N * sg.H;
    }
    const unsigned cin4 = 4u * (unsigned)p.Cin;
    const unsigned rowpitch = (unsigned)sg.W * cin4;
    const unsigned cin4_s = __builtin_amdgcn_readfirstlane(cin4);
    const unsigned rowpitch_s = __builtin_amdgcn_readfirstlane((unsigned)sg.W) * cin4_s;
    // SAFE form: cbase + c * cin4 is the byte offset of patch column c (+ this thread's channel); a column outside the image gets bit 30,
    // a row outside bit 31 = an offset beyond the feature map (< 1 GiB, checked at launch). The offsets are recomputed every slab.
    const unsigned cbase = (unsigned)wcol0 * cin4 + 4u * (unsigned)lch;
#define W36_ROWOFF(R) (((unsigned)min(max(hrow0 + (R), hlo), max(hhi - 1, hlo)) * rowpitch) | ((hrow0 + (R) >= hlo && hrow0 + (R) < hhi) ? 0u : 0x80000000u))
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * 4u * (unsigned)p.Cin - (SK ? (unsigned)s_first * 64u : 0u));
    const char *xbase = reinterpret_cast<const char *>(((size_t)xhi << 32) | xlo) + (SK ? (size_t)s_first * 64 : 0);   // (+ the first slab's channels)
    // LDS: stash address of this thread's 4 bytes inside unit [xi][q = c / 4][tile ^ swz(q)]; fragment unit of step (xi, h): [xi][2 h + lhalf][row ^ swz(q)].
    // swz(q) = 4 (q & 1) | 2 (q >> 1): with the first form of the kernel (tile ^ 4 q) the q = 0 / 2 and the q = 1 / 3 lanes of a half wave
    // shared their banks (SQ_LDS_BANK_CONFLICT: 4.2 M cycles per launch against 0 for the F(2x2) kernel, profiles/r12_wino_pmc.txt)
#define W36_SWZ(Q) ((((Q) & 1) << 2) | (((Q) >> 1) << 1))
    const unsigned st_base = (unsigned)((((lch >> 2) * W36_TM + (ltile ^ W36_SWZ(lch >> 2))) * 16) + (lch & 3) * 4);
    constexpr unsigned XI_PITCH = 4 * W36_TM * 16;       // bytes of one position's four planes
    const unsigned fr_base0 = (unsigned)((lhalf * W36_TM + (l32 ^ W36_SWZ(lhalf))) * 16) + (unsigned)pg * 9u * XI_PITCH;             // h = 0: q = lhalf
    const unsigned fr_base1 = (unsigned)(((2 + lhalf) * W36_TM + (l32 ^ W36_SWZ(2 + lhalf))) * 16) + (unsigned)pg * 9u * XI_PITCH;  // h = 1: q = 2 + lhalf
    // B: lane's float4 of step g = s * 72 + 2 xi + h sits at wbase + g * BSTEP + lhalf * 1024 + (32 cb + l32) * 16
    const size_t waddr = reinterpret_cast<size_t>(p.w) + ((size_t)n_t * (size_t)nslabs_all + (size_t)s_first) * (W36_STEPS * W36_BSTEP);
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)waddr), whi = __builtin_amdgcn_readfirstlane((unsigned)(waddr >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)whi << 32) | wlo), 0,
                                                                            nslabs * (int)(W36_STEPS * W36_BSTEP), 0x00020000);
    const unsigned b_lane = (unsigned)(lhalf * (W36_TN * 16) + (32 * cb + l32) * 16);
    const int gmax = nslabs * W36_STEPS - 1;

    floatx16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float ld[36];        // patch pixels of the slab being staged (row-major 6 x 6); after the row pass: (d B)[r][j]
    float4 breg[W36_RING];

    // FAST: (the 36 scalar offsets are loop-invariant; left alone the compiler keeps all of them in SGPRs, runs out, and spills to VGPR lanes
    // -- v_readlane is a VALU instruction. The opaque asm makes each row start from its own offset; five s_add per row cost nothing)
#define W36_LOADROW_FAST(R, RS) { unsigned so_ = (unsigned)((R) ? (R) - 1 : 0) * rowpitch_s; asm volatile("" : "+s"(so_));              \
        ld[6 * (R)] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(RS, (R) ? tbase + rowpitch_s : tbase, so_, 0));               \
        _Pragma("unroll") for (int c_ = 1; c_ < 6; ++c_) ld[6 * (R) + c_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(       \
            RS, ((R) ? tbase + rowpitch_s : tbase) + cin4_s, so_ + (unsigned)(c_ - 1) * cin4_s, 0)); }
    // SAFE: (the offsets are loop-invariant -- the slab advances through the descriptor's base -- and the compiler, left alone, hoists all 36
    // of them out of the slab loop into registers that stay live through the transform: the opaque asm makes each slab recompute them)
#define W36_LOADROW_SAFE(R, RS) { unsigned cb_ = cbase; asm volatile("" : "+v"(cb_)); const unsigned rb_ = W36_ROWOFF(R) + cb_; _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) \
        ld[6 * (R) + c_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(RS, (rb_ + (unsigned)c_ * cin4) | ((ncmask << (30 - c_)) & 0x40000000u), 0, 0)); }
    // FAST, border waves only: patch row R, pixels outside the image -> 0 (a real branch, not six selects in every wave)
#define W36_FIXROW(R) if (fix) { asm volatile("" ::: "memory"); const unsigned bad_ = ((nrmask >> (R)) & 1u) ? 0x3fu : ncmask;           \
        _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) ld[6 * (R) + c_] = ((bad_ >> c_) & 1u) ? 0.f : ld[6 * (R) + c_]; }
#define W36_ROWPASS(R) W36_BT(ld[6 * (R) + 0], ld[6 * (R) + 1], ld[6 * (R) + 2], ld[6 * (R) + 3], ld[6 * (R) + 4], ld[6 * (R) + 5])
    // column pass on column J + stash of V[i][J], i = 0..5, into the buffer at byte address SB
#define W36_COLSTASH(J, SB)                                                                                           \
    {                                                                                                                 \
        float c0_ = ld[0 + (J)], c1_ = ld[6 + (J)], c2_ = ld[12 + (J)], c3_ = ld[18 + (J)], c4_ = ld[24 + (J)], c5_ = ld[30 + (J)]; \
        W36_BT(c0_, c1_, c2_, c3_, c4_, c5_)                                                                          \
        *reinterpret_cast<float *>(smem_raw + (SB) + (0 * 6 + (J)) * XI_PITCH) = c0_;                                 \
        *reinterpret_cast<float *>(smem_raw + (SB) + (1 * 6 + (J)) * XI_PITCH) = c1_;                                 \
        *reinterpret_cast<float *>(smem_raw + (SB) + (2 * 6 + (J)) * XI_PITCH) = c2_;                                 \
        *reinterpret_cast<float *>(smem_raw + (SB) + (3 * 6 + (J)) * XI_PITCH) = c3_;                                 \
        *reinterpret_cast<float *>(smem_raw + (SB) + (4 * 6 + (J)) * XI_PITCH) = c4_;                                 \
        *reinterpret_cast<float *>(smem_raw + (SB) + (5 * 6 + (J)) * XI_PITCH) = c5_;                                 \
    }
#define W36_BLOAD(SLOT, G) { const w36_uintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_lane, (unsigned)min((G), gmax) * W36_BSTEP, 0); \
        breg[SLOT] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }

    // ---- prologue: slab 0 into buffer 0, first ring of B fragments
#pragma unroll
    for (int u = 0; u < W36_RING; ++u) W36_BLOAD(u, 18 * pg + u)
    {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbase), 0, xbytes, 0x00020000);
#pragma unroll
        for (int r = 0; r < 6; ++r) { W36_LOADROW_SAFE(r, xr) }
#pragma unroll
        for (int r = 0; r < 6; ++r) { W36_ROWPASS(r) }
#pragma unroll
        for (int j = 0; j < 6; ++j) { W36_COLSTASH(j, st_base) }
    }
    __syncthreads();
    float4 afr = *reinterpret_cast<const float4 *>(smem_raw + fr_base0);     // fragment of this wave's first step

    // the slab loop, instantiated twice (a workgroup-uniform choice between two whole loops: a branch around the LOADS inside one loop makes
    // the compiler's vmcnt bookkeeping conservative -- it then waits for the patch loads in the step that issues them)
    auto slab_loop = [&](auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    for (int s = 0; s < nslabs; ++s) {
        const int sn = min(s + 1, nslabs - 1);                               // next slab (last slab: harmless re-stage of itself)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbase) + (size_t)sn * 64, 0, xbytes - (unsigned)sn * 64u, 0x00020000);
        const unsigned cur = (s & 1) ? W36_VBUF : 0u, nxt = W36_VBUF - cur;
        const unsigned sb = nxt + st_base;
        const int g0 = s * W36_STEPS + 18 * pg;
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            // (1) global loads of the next slab's patch, spread over the first six steps (one patch row each)
            if (u < 6) { if constexpr (FAST) { W36_LOADROW_FAST(u, xr) } else { W36_LOADROW_SAFE(u, xr) } }
            // (2) A fragment of the next step, requested BEFORE this step's stashes: the wait in front of the next step's MFMAs then only has
            // to cover this (older) read, not the stash writes behind it (step 0 of the next slab comes from the other buffer, after the barrier)
            float4 afn;
            if (u < 17) afn = *reinterpret_cast<const float4 *>(smem_raw + cur + (((u + 1) & 1) ? fr_base1 : fr_base0) + (unsigned)((u + 1) >> 1) * XI_PITCH);
            else afn = *reinterpret_cast<const float4 *>(smem_raw + nxt + fr_base0);
            // (3) input transform of the next slab and its stash into the other buffer (done before the barrier of step 16)
            if (u == 6) { if constexpr (FAST) { W36_FIXROW(0) } W36_ROWPASS(0) }
            if (u == 7) { if constexpr (FAST) { W36_FIXROW(1) } W36_ROWPASS(1) }
            if (u == 8) { if constexpr (FAST) { W36_FIXROW(2) } W36_ROWPASS(2) }
            if (u == 9) { if constexpr (FAST) { W36_FIXROW(3) } W36_ROWPASS(3) }
            if (u == 10) { if constexpr (FAST) { W36_FIXROW(4) } W36_ROWPASS(4) }
            if (u == 11) { if constexpr (FAST) { W36_FIXROW(5) } W36_ROWPASS(5) W36_COLSTASH(0, sb) }
            if (u == 12) { W36_COLSTASH(1, sb) }
            if (u == 13) { W36_COLSTASH(2, sb) }
            if (u == 14) { W36_COLSTASH(3, sb) }
            if (u == 15) { W36_COLSTASH(4, sb) }
            if (u == 16) { W36_COLSTASH(5, sb) }
            // (4) the four MFMAs of step (xi = 9 pg + u / 2, h = u % 2): channels 4 (2 h + half) + m of the slab
            const float4 bf = breg[u % W36_RING];
            acc[u >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr.x, bf.x, acc[u >> 1], 0, 0, 0);
            acc[u >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr.y, bf.y, acc[u >> 1], 0, 0, 0);
            acc[u >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr.z, bf.z, acc[u >> 1], 0, 0, 0);
            acc[u >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr.w, bf.w, acc[u >> 1], 0, 0, 0);
            // (5) refill the ring slot just consumed: this wave's step W36_RING ahead (wraps into the next slab)
            W36_BLOAD(u % W36_RING, (u + W36_RING < 18 ? g0 : g0 + W36_STEPS - 18) + u + W36_RING)
            if (u == 16) __syncthreads();   // every read of `cur` is issued, every stash into `nxt` is visible
            afr = afn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    };
    if (risky) slab_loop(std::false_type{}); else slab_loop(std::true_type{});
#undef W36_LOADROW_FAST
#undef W36_LOADROW_SAFE
#undef W36_ROWOFF
#undef W36_FIXROW
#undef W36_ROWPASS
#undef W36_COLSTASH
#undef W36_BLOAD

    // ---- output transform Y = A^T M A through LDS, one column block at a time: [element r][xi][lane] floats
    const unsigned crow = (unsigned)p.Cout * 4u;
    const size_t oaddr_ = reinterpret_cast<size_t>(SK ? p.partial + (size_t)kz * (size_t)p.m_total * (size_t)p.Cout : sg.out);
    const unsigned obytes_ = __builtin_amdgcn_readfirstlane((unsigned)((long)sg.N * sg.OH * sg.OW) * crow);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(oaddr_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)oaddr_)),
        0, (int)obytes_, 0x00020000);
    float *xch = reinterpret_cast<float *>(smem_raw);
#pragma unroll 1
    for (int bb = 0; bb < 2; ++bb) {
        __syncthreads();                    // the V buffers (bb = 0) / the previous block's exchange (bb = 1) are no longer read
        if (cb == bb) {
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(r * W36_PTS + 9 * pg + u) * 64 + lane] = acc[u][r];
        }
        __syncthreads();
        const int co_ch = n_t * W36_TN + 32 * bb + l32;
        const bool co_ok = co_ch < p.Cout;
        const float bv = (p.bias != nullptr && co_ok) ? p.bias[co_ch] : 0.f;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
            const long pp = p0 + 8 * (r >> 2) + 4 * lhalf + (r & 3);     // this lane's tile
            float m[36];
#pragma unroll
            for (int x = 0; x < 36; ++x) m[x] = xch[(r * W36_PTS + x) * 64 + lane];
            if (!(co_ok && pp < sg.M)) continue;
            const int n = (int)(pp / HoWo);
            const int rem = (int)(pp - (long)n * HoWo);
            const int ty = rem / sg.Wo, tx = rem - ty * sg.Wo;
            float t[6][4];
#pragma unroll
            for (int i = 0; i < 6; ++i) W36_AT(m[6 * i + 0], m[6 * i + 1], m[6 * i + 2], m[6 * i + 3], m[6 * i + 4], m[6 * i + 5], t[i][0], t[i][1], t[i][2], t[i][3])
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float y[4];
                W36_AT(t[0][c], t[1][c], t[2][c], t[3][c], t[4][c], t[5][c], y[0], y[1], y[2], y[3])
                const int ox = 4 * tx + c;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oy = 4 * ty + a;
                    float v = y[a];
                    if constexpr (!SK) { v = v + bv; if (p.relu) v = fmaxf(v, 0.f); }
                    const unsigned off = (oy < sg.OH && ox < sg.OW) ? (unsigned)((n * sg.OH + oy) * sg.OW + ox) * crow + 4u * (unsigned)co_ch : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, off, 0, 0);
                }
            }
        }
    }
}

// geometry of the 3x3 / stride 1 / pad 1 convolution first (conv_fill), then the GEMM rows become 4x4 output tiles
static int w36_fill(ConvParams &p, const char *who, int nseg, const float *const x[], float *const out[], const int batch[], const int height[],
                    const int width[], int Cin, int Cout, const float *wpack, int ldw, const float *bias, int relu)
{
    UPS_REQUIRE(Cin > 0 && Cin % 32 == 0 && ldw % W36_TN == 0, "%s: Cin %% 32 == 0 and ldw %% 64 == 0 (got %d, %d)", who, Cin, ldw);
    int rc = conv_fill(p, who, nseg, x, nullptr, nullptr, nullptr, out, batch, height, width, Cin, Cout, wpack, ldw, bias, 3, 3, 1, 1, 1, relu);
    if (rc) return rc;
    UPS_REQUIRE((long)W36_PTS * Cin * ldw < (1L << 29), "%s: packed weight exceeds 2 GiB", who);
    int tiles = 0;
    for (int i = 0; i < nseg; ++i) {
        ConvSeg &s = p.seg[i];
        s.OH = s.Ho; s.OW = s.Wo;
        s.Ho = (s.OH + 3) / 4; s.Wo = (s.OW + 3) / 4;
        s.M = (long)s.N * s.Ho * s.Wo;
        UPS_REQUIRE((long)s.N * s.H * s.W * Cin < (1L << 28), "%s: feature map %d exceeds 1 GiB; split the batch", who, i);
        UPS_REQUIRE((long)s.N * s.OH * s.OW * Cout < (1L << 29), "%s: output %d exceeds 2 GiB; split the batch", who, i);
        s.tile_start = tiles;
        tiles += (int)((s.M + W36_TM - 1) / W36_TM);
    }
    p.m_tiles = tiles;
    p.n_tiles = ldw / W36_TN;
    return 0;
}

/* 3x3 / stride 1 / pad 1 convolution (+ bias, ReLU) of up to 5 NHWC maps sharing weights by Winograd F(4x4, 3x3): the contract of
 * upsnet_conv2d_winograd_nhwc_f32 without a residual. wpack from upsnet_conv_pack_weight_winograd36 (ldw = Cout rounded up to 64). */
extern "C" int upsnet_conv2d_winograd36_nhwc_f32(void *stream, int nseg, const float *const x[], float *const out[], const int batch[],
                                                 const int height[], const int width[], int Cin, const float *wpack, int ldw, const float *bias,
                                                 int Cout, int relu)
{
    ConvParams p;
    int rc = w36_fill(p, "conv2d_winograd36_nhwc_f32", nseg, x, out, batch, height, width, Cin, Cout, wpack, ldw, bias, relu);
    if (rc) return rc;
    static std::atomic<unsigned long long> attr_dev{0};
    UPS_ONCE_PER_DEVICE(attr_dev, UPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino36_f32_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W36_XCH)));
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;
    hipLaunchKernelGGL(conv_wino36_f32_kernel<false>, dim3(grid), dim3(512), W36_XCH, (hipStream_t)stream, p);
    UPS_CHECK_LAUNCH("conv_wino36_f32_kernel");
    ups_set_form("wino36<%d,%d>", W36_TM, W36_TN);
    return 0;
}

/* The same convolution of ONE map with the channel walk of every tile split over `ksplit` (2..8) workgroups (+ the shared reduce / epilogue
 * kernel; fixed summation order: bit-repeatable): for maps with fewer 32-tile x 64-channel workgroups than CUs. workspace:
 * upsnet_conv2d_winograd36_splitk_workspace_bytes bytes, caller-allocated. Cin / 16 >= ksplit. */
extern "C" size_t upsnet_conv2d_winograd36_splitk_workspace_bytes(int batch, int height, int width, int Cout, int ksplit)
{
    return (size_t)ksplit * (size_t)batch * height * width * Cout * sizeof(float);
}

extern "C" int upsnet_conv2d_winograd36_nhwc_f32_splitk(void *stream, const float *x, float *out, int batch, int height, int width, int Cin,
                                                        const float *wpack, int ldw, const float *bias, int Cout, int relu, int ksplit,
                                                        void *workspace)
{
    const float *xs[1] = {x};
    float *os[1] = {out};
    const int nb[1] = {batch}, hh[1] = {height}, ww[1] = {width};
    ConvParams p;
    int rc = w36_fill(p, "conv2d_winograd36_nhwc_f32_splitk", 1, xs, os, nb, hh, ww, Cin, Cout, wpack, ldw, bias, relu);
    if (rc) return rc;
    UPS_REQUIRE(workspace && ksplit >= 2 && ksplit <= 8 && Cin / 16 >= ksplit, "conv2d_winograd36_nhwc_f32_splitk: ksplit 2..8, <= Cin / 16, and a workspace");
    p.ksplit = ksplit;
    p.partial = (float *)workspace;
    p.m_total = (long)batch * height * width;         // rows of one partial: the output pixels
    UPS_REQUIRE(p.m_total * Cout < (1L << 29), "conv2d_winograd36_nhwc_f32_splitk: output exceeds 2 GiB; split the batch");
    static std::atomic<unsigned long long> attr_dev{0};
    UPS_ONCE_PER_DEVICE(attr_dev, UPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino36_f32_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W36_XCH)));
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles * ksplit;
    hipLaunchKernelGGL(conv_wino36_f32_kernel<true>, dim3(grid), dim3(512), W36_XCH, (hipStream_t)stream, p);
    UPS_CHECK_LAUNCH("conv_wino36_f32_kernel<splitk>");
    ups_set_form("wino36<%d,%d> splitk%d", W36_TM, W36_TN, ksplit);
    return conv_splitk_reduce((hipStream_t)stream, p.partial, ksplit, p.m_total, p.m_total, Cout, bias, nullptr, relu, out);
}

// weight [Cout, Cin, 3, 3] -> U = G g G^T (double precision, rounded once), G of the points {0, 3/4, -3/4, 3/2, -3/2, inf} =
// [[64/81,0,0],[-128/243,-32/81,-8/27],[-128/243,32/81,-8/27],[32/243,16/81,8/27],[32/243,-16/81,8/27],[0,0,1]], stored in fragment order [n-tile = co/64][slab = c/16][xi = 6i+j][q = (c%16)/4][co%64][c%4]
__global__ void conv_pack_weight_wino36_kernel(const float *__restrict__ w, int cout, int cin, int ldw, float *__restrict__ wp)
{
    const long total = (long)ldw * cin;
    const int nslabs = cin >> 4;
    const double G[6][3] = {{64.0 / 81, 0.0, 0.0}, {-128.0 / 243, -32.0 / 81, -8.0 / 27}, {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                            {32.0 / 243, 16.0 / 81, 8.0 / 27}, {32.0 / 243, -16.0 / 81, 8.0 / 27}, {0.0, 0.0, 1.0}};
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int co = idx % ldw, c = idx / ldw;
        double g[3][3], t[6][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = co < cout ? (double)w[(((long)co * cin + c) * 3 + a) * 3 + b] : 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
        const long blk = ((long)(co / W36_TN) * nslabs + (c >> 4)) * W36_PTS;
        const int q = (c & 15) >> 2, ci = c & 3, cl = co % W36_TN;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                wp[(((blk + i * 6 + j) * 4 + q) * W36_TN + cl) * 4 + ci] = (float)u;
            }
    }
}

/* weight [Cout, Cin, 3, 3] -> the operand of upsnet_conv2d_winograd36_nhwc_f32: 36 * Cin * ldw floats, ldw = Cout rounded up to 64. */
extern "C" int upsnet_conv_pack_weight_winograd36(void *stream, const float *weight, int cout, int cin, int ldw, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cin > 0 && ldw >= cout, "conv_pack_weight_winograd36: bad args");
    UPS_REQUIRE(cin % 16 == 0 && ldw % W36_TN == 0, "conv_pack_weight_winograd36: Cin %% 16 must be 0 and ldw a multiple of 64");
    const long total = (long)ldw * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv_pack_weight_wino36_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, ldw, wpack);
    UPS_CHECK_LAUNCH("conv_pack_weight_wino36_kernel");
    return 0;
}
