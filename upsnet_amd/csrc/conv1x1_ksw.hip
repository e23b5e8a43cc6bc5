// conv1x1_ksw.hip -- 1x1 convolution (stride 1 / 2) on SMALL tiles of v_mfma_f32_16x16x4_f32 fragments, the K walk split over the four
// WAVES of a workgroup (r13; VERDICT r05 next #2: "the small-tile 16x16x4 fragment family").
//
// Reference: the 1x1 nn.Conv2d layers of the ResNet bottlenecks (conv1 / conv3, upsnet/models/resnet.py:53-100,102-153) followed by
// separate frozen-BN / add / ReLU passes.
//
// Why a third 1x1 kernel. conv1x1_frag_f32_kernel (conv1x1.hip) tiles the GEMM in 64 pixels x 64 / 128 channels, one workgroup keeping a
// CU's matrix pipe busy by itself: a launch lasts (most workgroups on one CU) x (one K walk). Maps that are no multiple of the chip --
// every map of UPSNet-101-DCN at 800x1333: 50 x 84 = 4200 pixels = 66 tiles x 4 = 264 workgroups on 256 CUs -- take TWO walks for 1.03
// walks of work (profiles/r12_layer_table_c3.txt: 1024 -> 256 on that map 35.8 us for 14 us of matrix time, 23 such layers per image).
// The cure is granularity: 16 / 32-pixel tiles give 1052 quarter-size workgroups = 5 quarter walks on the fullest CU (1.25 walks).
// With 32x32x2 fragments a tile cannot be narrower than 32 x 32 and four waves cannot share one; with 16x16x4 fragments
// (tools/ubench/mfma16_valu.hip, profiles/r13_mfma16_valu.txt: 0.99 of the fp32 MFMA rate from two waves per SIMD, 40-cycle dependent
// latency hidden by two independent accumulators) a wave holds the WHOLE 16 RB x 16 CB tile in 4 RB CB registers and the four waves of the
// workgroup each walk a quarter of K:
//   * no LDS and no barrier in the K loop. A lane's A operand of four consecutive MFMAs is four consecutive channels of one pixel = one
//     16-byte buffer load straight from the NHWC map (lane (k, i) = channels 4k..4k+3 of the 16-channel step, pixel i of the row block);
//     its B operand is one 16-byte load from the weights packed in fragment order (L2-resident). A step (16 channels) is RB + CB loads
//     for 4 RB CB MFMAs; steps are prefetched RING deep in registers. Pixels beyond the map and steps beyond a wave's range read 0
//     through the bounds check of the buffer descriptor (out-of-range offset: no memory access).
//   * the four partial tiles meet once, in LDS, after the walk: out = ((wave0 + wave1) + (wave2 + wave3)) + bias + residual, ReLU -- a fixed
//     order (bit-repeatable), and the transposition through LDS makes every output store a full 16-byte channel quad of one pixel.
// Instances (RB, CB): 16 x 64, 32 x 32, 32 x 64, 64 x 64 pixels x channels. The price of a small tile is operand traffic: each weight is
// fetched from L2 once per 16 RB pixels (32 x 32: 32 B / clk / CU at the full MFMA rate, against 24 for conv1x1_frag), so the 64-pixel
// kernel keeps the maps it tiles evenly; models/hipconv.py picks per layer (shape only).
#include "conv_params.h"
#include "upsnet_hip.h"

typedef float kfloatx4 __attribute__((ext_vector_type(4)));
typedef unsigned kuintx4 __attribute__((ext_vector_type(4)));

#define KSW_MAXCB 4   // column blocks (16 channels) the pack is padded to: every n-tile of every instance reads valid memory

// weight [Cout, Cin] -> fragment order [cbk = co / 16][step = c / 16][lane = 16 ((c % 16) / 4) + co % 16][c % 4], column blocks padded
// to a multiple of KSW_MAXCB with zeros
__global__ void conv1x1_ksw_pack_kernel(const float *__restrict__ w, int cout, int cin, int nblk, float *__restrict__ wp)
{
    const long total = (long)nblk * 16 * cin;
    const int nst = cin >> 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int c4 = idx & 3, j = (idx >> 2) & 15, k = (idx >> 6) & 3;
        const long r = idx >> 8;
        const int s = (int)(r % nst), cbk = (int)(r / nst);
        const int co = 16 * cbk + j, c = 16 * s + 4 * k + c4;
        wp[idx] = co < cout ? w[(long)co * cin + c] : 0.f;
    }
}

extern "C" size_t upsnet_conv1x1_ksw_packed_weight_floats(int cout, int cin)
{
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)((cout + 16 * KSW_MAXCB - 1) / (16 * KSW_MAXCB)) * (16 * KSW_MAXCB) * (size_t)cin;
}

extern "C" int upsnet_conv1x1_ksw_pack_weight(void *stream, const float *weight, int cout, int cin, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cin > 0 && cin % 16 == 0, "conv1x1_ksw_pack_weight: bad args (Cin %% 16 must be 0)");
    const int nblk = (cout + 16 * KSW_MAXCB - 1) / (16 * KSW_MAXCB) * KSW_MAXCB;
    const long total = (long)nblk * 16 * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv1x1_ksw_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, nblk, wpack);
    UPS_CHECK_LAUNCH("conv1x1_ksw_pack_kernel");
    return 0;
}

// RB / CB: 16-pixel row blocks / 16-channel column blocks of the workgroup's tile (every wave holds all of them); RING: steps in flight
// per wave; WPE: waves per SIMD the register budget is set for.
// KS true: the four waves split K and hold the same 16 RB x 16 CB tile (partial sums added in LDS); false: the four waves split N -- the
// workgroup's tile is 16 RB x 64 CB, wave w owns column blocks [w CB, (w + 1) CB) and walks all of K (layers with a short K walk and many
// output channels, a bottleneck's conv3: the four-way reduction and the prologue of a 4-step walk would cost more than they spread).
template <int RB, int CB, int RING, int WPE, bool KS>
__global__ void __launch_bounds__(256, WPE) conv1x1_ksw_f32_kernel(const ConvParams p)
{
    constexpr int BM = 16 * RB, BN = 16 * CB;       // BN: channels per WAVE (KS: = per workgroup)
    constexpr int PRB = RB > 2 ? 2 : RB;            // row blocks reduced per pass through LDS
    constexpr int LDP = BN + 4;                     // floats per row of a partial tile in LDS (16-byte aligned rows, banks spread)
    __shared__ __attribute__((aligned(16))) float red[4 * 16 * PRB * LDP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, li = lane & 15;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): contiguous m-tile range per XCD, all n-tiles of an m-tile together
    int m_t, n_t;
    {
        const int nt = p.n_tiles;
        const int per = (p.m_tiles + 7) >> 3;
        const int bid = (int)blockIdx.x;
        const int qq = bid >> 3;
        n_t = qq % nt;
        const int local = qq / nt;
        m_t = (bid & 7) * per + local;
        if (local >= per || m_t >= p.m_tiles) return;
    }
    const ConvSeg sg = p.seg[0];
    const long p0 = (long)m_t * BM;
    const int nst = p.Cin >> 4;                     // 16-channel steps of the whole K walk
    const int s_begin = KS ? (wave * nst) >> 2 : 0, s_end = KS ? ((wave + 1) * nst) >> 2 : nst;   // this wave's share (wave-uniform; may be empty)
    const int wn_t = KS ? n_t : 4 * n_t + wave;     // the wave's BN-wide column tile
    const long HoWo = (long)sg.Ho * sg.Wo;

    // ---- A: byte offset of (pixel 16 rb + li, channels 4 lk ..) in the NHWC map; bit 31 beyond the map (the load then returns 0)
    unsigned po[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const long pp = p0 + 16 * rb + li;
        po[rb] = 0x80000000u;
        if (pp < sg.M) {
            const int n = (int)(pp / HoWo);
            const int rem = (int)(pp - (long)n * HoWo);
            const int ho = rem / sg.Wo, wo = rem - ho * sg.Wo;
            po[rb] = (unsigned)((n * sg.H + ho * p.stride) * sg.W + wo * p.stride) * 4u * (unsigned)p.Cin + 16u * (unsigned)lk;
        }
    }
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * 4u * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)xhi << 32) | xlo), 0, (int)xbytes, 0x00020000);
    // ---- B: lane's float4 of (column block cb of this n-tile, step s) at wbase + (cb nst + s) * 1024 + lane * 16
    // (p.ldw / 16 = column blocks in the pack. N split: a wave of the last workgroup may lie beyond them -- its descriptor then covers
    // nothing, every load reads 0, and its stores are outside Cout)
    const int nblk = p.ldw >> 4;
    const int cb_first = min(wn_t * CB, nblk), cb_have = min(CB, nblk - cb_first);
    const size_t waddr = reinterpret_cast<size_t>(p.w) + (size_t)cb_first * (size_t)nst * 1024u;
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)waddr), whi = __builtin_amdgcn_readfirstlane((unsigned)(waddr >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)whi << 32) | wlo), 0,
                                                                            __builtin_amdgcn_readfirstlane(cb_have * nst * 1024), 0x00020000);
    const unsigned b_lane = (unsigned)lane * 16u;

    kfloatx4 acc[RB][CB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = (kfloatx4){0.f, 0.f, 0.f, 0.f};
    kfloatx4 ra[RING][RB], rw[RING][CB];

    // step S into ring slot U: a step beyond this wave's range reads zeros for A (offset out of range) and the last valid weights for B
    // (clamped), i.e. it adds exact zeros -- for FINITE weights (include/upsnet_hip.h states the assumption, as for conv1x1_frag)
#define KSW_LOAD(U, S)                                                                                                                 \
    {                                                                                                                                  \
        const int s_ = (S);                                                                                                            \
        const unsigned ao_ = s_ < s_end ? (unsigned)s_ * 64u : 0x80000000u;                                                            \
        const unsigned bo_ = (unsigned)min(s_, nst - 1) * 1024u;                                                                       \
        _Pragma("unroll") for (int rb_ = 0; rb_ < RB; ++rb_) {                                                                         \
            const kuintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, po[rb_] + ao_, 0, 0);                                      \
            ra[U][rb_] = (kfloatx4){__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)};       \
        }                                                                                                                              \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_) {                                                                         \
            const kuintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_lane, bo_ + (unsigned)(cb_ * nst) * 1024u, 0);           \
            rw[U][cb_] = (kfloatx4){__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)};       \
        }                                                                                                                              \
    }
#pragma unroll
    for (int u = 0; u < RING; ++u) {
        KSW_LOAD(u, s_begin + u)
        // (the prologue must issue the slots in ring order: the wait count the compiler derives for the loop is the merge of this entry state
        // and the back edge -- with the slots reordered here, slot 0 was the YOUNGEST load on entry and every iteration began with vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int s = s_begin; s < s_end; s += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][rb][e], rw[u][cb][e], acc[rb][cb], 0, 0, 0);
            KSW_LOAD(u, s + u + RING)
            __builtin_amdgcn_sched_barrier(0);      // (left alone the compiler sinks every load to the end of the iteration and waits vmcnt(0) at its top)
        }
    }
#undef KSW_LOAD

    // ---- the four partial tiles meet in LDS (PRB row blocks per pass): accumulator register r of lane (g = lane / 16, j = lane % 16) is
    // row 4 g + r, column j of its 16 x 16 block. Then thread t owns channel quads: out = ((w0 + w1) + (w2 + w3)) + bias + residual, ReLU.
    const int cw = p.Cout;
    const bool has_res = sg.res != nullptr;
#pragma unroll
    for (int rb0 = 0; rb0 < RB; rb0 += PRB) {
        if (rb0) __syncthreads();                    // (N split: a wave only ever touches its own region, but the barrier is uniform and cheap)
#pragma unroll
        for (int rb = 0; rb < PRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((wave * PRB + rb) * 16 + 4 * lk + r) * LDP + 16 * cb + li] = acc[rb0 + rb][cb][r];
        __syncthreads();
        constexpr int QPR = BN / 4;                 // channel quads per row
        constexpr int QUADS = 16 * PRB * QPR;
#pragma unroll
        for (int gi = KS ? tid : lane; gi < QUADS; gi += KS ? 256 : 64) {
            const int row = gi / QPR, c4 = gi - row * QPR;
            float4 v;
            if constexpr (KS) {
                const float4 v0 = *reinterpret_cast<const float4 *>(&red[(0 * PRB * 16 + row) * LDP + 4 * c4]);
                const float4 v1 = *reinterpret_cast<const float4 *>(&red[(1 * PRB * 16 + row) * LDP + 4 * c4]);
                const float4 v2 = *reinterpret_cast<const float4 *>(&red[(2 * PRB * 16 + row) * LDP + 4 * c4]);
                const float4 v3 = *reinterpret_cast<const float4 *>(&red[(3 * PRB * 16 + row) * LDP + 4 * c4]);
                v = make_float4((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w));
            } else {
                v = *reinterpret_cast<const float4 *>(&red[(wave * PRB * 16 + row) * LDP + 4 * c4]);   // the wave's own tile, transposed for the store
            }
            const long pp = p0 + 16 * rb0 + row;
            const int co = wn_t * BN + 4 * c4;
            if (pp < sg.M && co < cw) {             // (Cout % 4 == 0: a quad is inside or outside)
                if (p.bias != nullptr) {
                    const float4 b = *reinterpret_cast<const float4 *>(p.bias + co);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (has_res) {
                    const float4 rr = *reinterpret_cast<const float4 *>(sg.res + pp * cw + co);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4 *>(sg.out + pp * cw + co) = v;
            }
        }
    }
}

// development knob: 0 = the caller's / automatic choice, else 1000 split + 100 RB + CB of the WAVE's tile (split 0: K over the waves, 1: N over the waves)
static int g_ksw_tile = 0;
extern "C" void upsnet_conv1x1_ksw_tuning(int tile) { g_ksw_tile = tile; }

/* out = relu?(conv1x1(x, w; stride) + bias + residual) on the small-tile kernel. x [N,H,W,Cin] NHWC, out [N,Ho,Wo,Cout] NHWC, residual like
 * out or NULL. wpack: upsnet_conv1x1_ksw_pack_weight. split_n == 0: the four waves of a workgroup split K; tile_pixels x tile_channels (of the
 * workgroup) 16 x 64, 32 x 32, 32 x 64 or 64 x 64. split_n == 1: the waves split N; 16 x 256, 32 x 128 or 32 x 256. Any other pair: error.
 * Cin % 16 == 0, Cout % 4 == 0, all pointers 16-byte aligned. Finite weights assumed (a K step beyond a wave's range multiplies zeros by the
 * last valid weights). Fixed summation order: bit-repeatable, independent of the batch size for a given (tile, split). */
extern "C" int upsnet_conv1x1_ksw_nhwc_f32(void *stream, const float *x, const float *residual, float *out, int batch, int height, int width,
                                           int Cin, const float *wpack, const float *bias, int Cout, int stride, int relu, int tile_pixels,
                                           int tile_channels, int split_n)
{
    UPS_REQUIRE(x && out && wpack && batch > 0 && height > 0 && width > 0, "conv1x1_ksw_nhwc_f32: bad args");
    UPS_REQUIRE(Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout % 4 == 0, "conv1x1_ksw_nhwc_f32: Cin %% 16 and Cout %% 4 must be 0 (got %d, %d)", Cin, Cout);
    UPS_REQUIRE(stride >= 1, "conv1x1_ksw_nhwc_f32: bad stride");
    UPS_REQUIRE(((reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(residual) | reinterpret_cast<size_t>(bias) | reinterpret_cast<size_t>(x) |
                  reinterpret_cast<size_t>(wpack)) & 15) == 0, "conv1x1_ksw_nhwc_f32: pointers must be 16-byte aligned");
    ConvParams p;
    p.w = wpack; p.bias = bias; p.nseg = 1; p.Cin = Cin; p.Cout = Cout; p.ldw = (Cout + 16 * KSW_MAXCB - 1) / (16 * KSW_MAXCB) * (16 * KSW_MAXCB);   /* padded columns of the pack */ p.KH = p.KW = 1;
    p.stride = stride; p.pad = 0; p.dil = 1; p.relu = relu; p.res_up = 0;
    p.ksplit = 1; p.partial = nullptr; p.m_total = 0; p.io = 0; p.sib_split = 0; p.sib_relu = 0; p.sib_out = nullptr;
    for (int i = 0; i < CV_MAXSEG; ++i) {
        ConvSeg &s = p.seg[i];
        s.x = s.res = s.off = s.mask = nullptr; s.w = nullptr; s.out = nullptr;
        s.N = s.H = s.W = s.Ho = s.Wo = s.OH = s.OW = 0; s.M = 0; s.tile_start = 0x7fffffff;
    }
    ConvSeg &s = p.seg[0];
    s.x = x; s.res = residual; s.out = out; s.N = batch; s.H = height; s.W = width;
    s.Ho = (height - 1) / stride + 1; s.Wo = (width - 1) / stride + 1;
    s.M = (long)batch * s.Ho * s.Wo; s.tile_start = 0;
    UPS_REQUIRE((long)batch * height * width * Cin < (1L << 29), "conv1x1_ksw_nhwc_f32: feature map exceeds 2 GiB; split the batch");
    if (g_ksw_tile) { split_n = g_ksw_tile / 1000; tile_pixels = 16 * ((g_ksw_tile % 1000) / 100); tile_channels = 16 * (g_ksw_tile % 100) * (split_n ? 4 : 1); }
    const int rb = tile_pixels / 16, cb = tile_channels / (split_n ? 64 : 16);       // blocks of a WAVE's tile
    UPS_REQUIRE(tile_pixels == 16 * rb && tile_channels == (split_n ? 64 : 16) * cb &&
                    ((rb == 1 && cb == 4) || (rb == 2 && cb == 2) || (rb == 2 && cb == 4) || (rb == 4 && cb == 4 && !split_n)),
                "conv1x1_ksw_nhwc_f32: tile %d x %d (split %s) is not one of 16x64, 32x32, 32x64, 64x64 (K) / 16x256, 32x128, 32x256 (N)",
                tile_pixels, tile_channels, split_n ? "N" : "K");
    p.m_tiles = (int)((s.M + tile_pixels - 1) / tile_pixels);
    p.n_tiles = (Cout + tile_channels - 1) / tile_channels;
    // (N split: the last workgroup's waves may start beyond the padded pack -- their loads are out of range of the descriptor and read 0,
    // their stores are outside Cout and dropped)
    UPS_REQUIRE((long)(p.n_tiles * (split_n ? 4 : 1) + 1) * cb * (Cin / 16) * 1024 < (1L << 31), "conv1x1_ksw_nhwc_f32: packed weight exceeds 2 GiB");
    const int grid = 8 * ((p.m_tiles + 7) / 8) * p.n_tiles;
    hipStream_t st = (hipStream_t)stream;
#define KSW_GO(RB, CB, RING, WPE) { if (split_n) hipLaunchKernelGGL((conv1x1_ksw_f32_kernel<RB, CB, RING, WPE, false>), dim3(grid), dim3(256), 0, st, p); \
                                    else hipLaunchKernelGGL((conv1x1_ksw_f32_kernel<RB, CB, RING, WPE, true>), dim3(grid), dim3(256), 0, st, p); }
    // RING = 2 everywhere (measured r13, tools/bench_conv1x1_ksw.py, RING 1 / 2 / 3 / 4 / 6 at the occupancy the registers allow: a deeper ring
    // lengthens the prologue and rounds every wave's walk up to more wasted steps; occupancy hides the latency instead)
    if (rb == 1) KSW_GO(1, 4, 2, 6)
    else if (rb == 2 && cb == 2) KSW_GO(2, 2, 2, 6)
    else if (rb == 2) KSW_GO(2, 4, 2, 4)
    else KSW_GO(4, 4, 2, 2)
#undef KSW_GO
    UPS_CHECK_LAUNCH("conv1x1_ksw_f32_kernel");
    ups_set_form("conv1x1_ksw<%d,%d,%c>", tile_pixels, tile_channels, split_n ? 'n' : 'k');
    return 0;
}

// =====================================================================================================================================
// 3x3 / stride 1 / pad 1 convolution into FEW channels (<= 32) on the same scheme: the 18-channel offset predictors of the deformable
// bottlenecks (conv2_offset, upsnet/models/resnet.py:102-153; UPSNet-101-DCN has 30, on maps of 16 800 / 4200 / 1050 pixels at 800x1333).
// On the general kernel such a layer was 33 workgroups of 128 pixels walking 72 slabs, split 8 ways over K + a reduce launch: 22 us for
// 4 us of matrix work, 23 times per image (profiles/r12_layer_table_c3.txt). Here a workgroup is 16 pixels x 32 channels, its four waves
// split the 9 Cin / 16 steps of the (tap, channel) walk; a lane's A operand is 16 bytes of the tap's shifted pixel straight from the map
// (a tap outside the image: out-of-range offset, reads the 0 of the padding), B from the fragment-order pack; the partial tiles meet in LDS.
// weight [Cout, Cin, 3, 3] -> [cbk = co / 16 (2 blocks)][step = tap * (Cin / 16) + c / 16][lane = 16 ((c % 16) / 4) + co % 16][c % 4]
__global__ void conv3x3_ksw_pack_kernel(const float *__restrict__ w, int cout, int cin, float *__restrict__ wp)
{
    const long total = (long)2 * 16 * cin * 9;
    const int nch = cin >> 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)blockDim.x * gridDim.x) {
        const int c4 = idx & 3, j = (idx >> 2) & 15, k = (idx >> 6) & 3;
        const long r = idx >> 8;
        const int s = (int)(r % (9 * nch)), cbk = (int)(r / (9 * nch));
        const int tap = s / nch, ch = s - tap * nch;
        const int co = 16 * cbk + j, c = 16 * ch + 4 * k + c4;
        wp[idx] = co < cout ? w[((long)co * cin + c) * 9 + tap] : 0.f;
    }
}

extern "C" size_t upsnet_conv3x3_ksw_packed_weight_floats(int cin) { return cin > 0 ? (size_t)32 * 9 * (size_t)cin : 0; }

extern "C" int upsnet_conv3x3_ksw_pack_weight(void *stream, const float *weight, int cout, int cin, float *wpack)
{
    UPS_REQUIRE(weight && wpack && cout > 0 && cout <= 32 && cin > 0 && cin % 16 == 0, "conv3x3_ksw_pack_weight: Cout <= 32 and Cin %% 16 == 0 (got %d, %d)", cout, cin);
    const long total = (long)32 * 9 * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(conv3x3_ksw_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, wpack);
    UPS_CHECK_LAUNCH("conv3x3_ksw_pack_kernel");
    return 0;
}

// NW: waves per workgroup = shares of the (tap, channel) walk (4, 8 or 16); RB: 16-pixel row blocks per workgroup
template <int RING, int NW, int RB>
__global__ void __launch_bounds__(64 * NW, 1) conv3x3_ksw_f32_kernel(const ConvParams p)
{
    constexpr int LDP = 36;
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * RB * LDP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, li = lane & 15;
    const ConvSeg sg = p.seg[0];
    const long p0 = (long)blockIdx.x * (16 * RB);
    const int nch = p.Cin >> 4, nst = 9 * nch;
    const int s_begin = (wave * nst) / NW, s_end = ((wave + 1) * nst) / NW;
    const long HW = (long)sg.H * sg.W;
    // this lane's pixels: byte offset of the channel vector (+ 16 lk), and the 9-bit mask of taps that fall inside the image
    unsigned base[RB], tapmask[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        base[rb] = 0; tapmask[rb] = 0;
        const long pp = p0 + 16 * rb + li;
        if (pp < sg.M) {
            const int n = (int)(pp / HW);
            const int rem = (int)(pp - (long)n * HW);
            const int h = rem / sg.W, w = rem - h * sg.W;
            base[rb] = (unsigned)pp * 4u * (unsigned)p.Cin + 16u * (unsigned)lk;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
                tapmask[rb] |= (hh >= 0 && hh < sg.H && ww >= 0 && ww < sg.W) ? (1u << t) : 0u;
            }
        }
    }
    const size_t xaddr = reinterpret_cast<size_t>(sg.x);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const unsigned xbytes = __builtin_amdgcn_readfirstlane((unsigned)(sg.N * sg.H * sg.W) * 4u * (unsigned)p.Cin);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)xhi << 32) | xlo), 0, (int)xbytes, 0x00020000);
    const size_t waddr = reinterpret_cast<size_t>(p.w);
    const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)waddr), whi = __builtin_amdgcn_readfirstlane((unsigned)(waddr >> 32));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((size_t)whi << 32) | wlo), 0, 2 * nst * 1024, 0x00020000);
    const unsigned b_lane = (unsigned)lane * 16u;
    const int pixb = 4 * p.Cin;                     // bytes of one pixel's channel vector

    kfloatx4 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc[rb][0] = (kfloatx4){0.f, 0.f, 0.f, 0.f}; acc[rb][1] = (kfloatx4){0.f, 0.f, 0.f, 0.f}; }
    kfloatx4 ra[RING][RB], rw0[RING], rw1[RING];
    // step S = tap * nch + chunk (wave-uniform): pixel shifted by the tap, channels 16 chunk + 4 lk ..; a tap outside the image or a step
    // beyond this wave's range: offset out of range, reads 0 (B then multiplies zeros: finite weights assumed, as above)
#define K3_LOAD(U, S)                                                                                                                  \
    {                                                                                                                                  \
        const int s_ = (S);                                                                                                            \
        const int sc_ = min(s_, nst - 1);                                                                                              \
        const int tap_ = sc_ / nch, ch_ = sc_ - tap_ * nch;                                                                            \
        const int ty_ = tap_ / 3, tx_ = tap_ - 3 * ty_;                                                                                \
        const int delta_ = ((ty_ - 1) * sg.W + (tx_ - 1)) * pixb + ch_ * 64;                                                           \
        _Pragma("unroll") for (int rb_ = 0; rb_ < RB; ++rb_) {                                                                         \
            const bool ok_ = s_ < s_end && ((tapmask[rb_] >> tap_) & 1u);                                                              \
            const kuintx4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok_ ? base[rb_] + (unsigned)delta_ : 0x80000000u, 0, 0);   \
            ra[U][rb_] = (kfloatx4){__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)};       \
        }                                                                                                                              \
        const kuintx4 b0_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_lane, (unsigned)sc_ * 1024u, 0);                            \
        const kuintx4 b1_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_lane, (unsigned)(nst + sc_) * 1024u, 0);                    \
        rw0[U] = (kfloatx4){__uint_as_float(b0_.x), __uint_as_float(b0_.y), __uint_as_float(b0_.z), __uint_as_float(b0_.w)};           \
        rw1[U] = (kfloatx4){__uint_as_float(b1_.x), __uint_as_float(b1_.y), __uint_as_float(b1_.z), __uint_as_float(b1_.w)};           \
    }
#pragma unroll
    for (int u = 0; u < RING; ++u) {
        K3_LOAD(u, s_begin + u)
        __builtin_amdgcn_sched_barrier(0);          // (ring order: see conv1x1_ksw_f32_kernel)
    }
    for (int s = s_begin; s < s_end; s += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][rb][e], rw0[u][e], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[u][rb][e], rw1[u][e], acc[rb][1], 0, 0, 0);
                }
            K3_LOAD(u, s + u + RING)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef K3_LOAD
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[((wave * RB + rb) * 16 + 4 * lk + r) * LDP + li] = acc[rb][0][r];
            red[((wave * RB + rb) * 16 + 4 * lk + r) * LDP + 16 + li] = acc[rb][1][r];
        }
    __syncthreads();
    // 16 RB pixels x 32 channels over the workgroup's threads; the NW partial sums are added as a fixed binary tree
    for (int e = tid; e < 512 * RB; e += 64 * NW) {
        const int row = e >> 5, co = e & 31;
        const long pp = p0 + row;
        if (pp < sg.M && co < p.Cout) {
            float t[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) t[w] = red[(w * RB * 16 + row) * LDP + co];
#pragma unroll
            for (int span = 1; span < NW; span *= 2)
#pragma unroll
                for (int w = 0; w < NW; w += 2 * span) t[w] = t[w] + t[w + span];
            float v = t[0];
            if (p.bias != nullptr) v += p.bias[co];
            if (p.relu) v = fmaxf(v, 0.f);
            sg.out[pp * p.Cout + co] = v;
        }
    }
}

/* out = relu?(conv3x3(x; stride 1, pad 1) + bias) for Cout <= 32 (the 18-channel offset predictors of the deformable bottlenecks,
 * upsnet/models/resnet.py:102-153) on small maps: 16-pixel x 32-channel workgroups, K split over the waves. x [N,H,W,Cin] NHWC,
 * out [N,H,W,Cout] NHWC; Cin % 16 == 0; wpack from upsnet_conv3x3_ksw_pack_weight (weight [Cout, Cin, 3, 3]). Finite weights assumed.
 * Fixed summation order for a given pixel count: bit-repeatable (the wave count follows the tile count: not for ROI batches). */
extern "C" int upsnet_conv3x3_ksw_nhwc_f32(void *stream, const float *x, float *out, int batch, int height, int width, int Cin,
                                           const float *wpack, const float *bias, int Cout, int relu)
{
    UPS_REQUIRE(x && out && wpack && batch > 0 && height > 0 && width > 0, "conv3x3_ksw_nhwc_f32: bad args");
    UPS_REQUIRE(Cin > 0 && Cin % 16 == 0 && Cout > 0 && Cout <= 32, "conv3x3_ksw_nhwc_f32: Cin %% 16 == 0 and Cout <= 32 (got %d, %d)", Cin, Cout);
    UPS_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(wpack)) & 15) == 0, "conv3x3_ksw_nhwc_f32: x / wpack must be 16-byte aligned");
    UPS_REQUIRE((long)batch * height * width * Cin < (1L << 29), "conv3x3_ksw_nhwc_f32: feature map exceeds 2 GiB; split the batch");
    ConvParams p;
    p.w = wpack; p.bias = bias; p.nseg = 1; p.Cin = Cin; p.Cout = Cout; p.ldw = 32; p.KH = p.KW = 3;
    p.stride = 1; p.pad = 1; p.dil = 1; p.relu = relu; p.res_up = 0;
    p.ksplit = 1; p.partial = nullptr; p.m_total = 0; p.io = 0; p.sib_split = 0; p.sib_relu = 0; p.sib_out = nullptr;
    for (int i = 0; i < CV_MAXSEG; ++i) {
        ConvSeg &s = p.seg[i];
        s.x = s.res = s.off = s.mask = nullptr; s.w = nullptr; s.out = nullptr;
        s.N = s.H = s.W = s.Ho = s.Wo = s.OH = s.OW = 0; s.M = 0; s.tile_start = 0x7fffffff;
    }
    ConvSeg &s = p.seg[0];
    s.x = x; s.out = out; s.N = batch; s.H = s.Ho = height; s.W = s.Wo = width;
    s.M = (long)batch * height * width; s.tile_start = 0;
    // (measured r13, tools/bench_conv3x3_ksw.py: 16-pixel tiles; 32-pixel ones are no faster on 4200-pixel maps and slower on 1050-pixel ones.
    // 4 / 8 / 16 waves per workgroup differ by < 1 us; more waves for fewer tiles keeps the SIMDs supplied)
    const int rb = 1;
    p.m_tiles = (int)((s.M + 16 * rb - 1) / (16 * rb)); p.n_tiles = 1;
    const int nw = p.m_tiles >= 192 ? 4 : (p.m_tiles >= 96 ? 8 : 16);
    hipStream_t st = (hipStream_t)stream;
#define K3_GO(NW, RB) hipLaunchKernelGGL((conv3x3_ksw_f32_kernel<2, NW, RB>), dim3(p.m_tiles), dim3(64 * NW), 0, st, p)
    if (nw == 16) K3_GO(16, 1); else if (nw == 8) K3_GO(8, 1); else K3_GO(4, 1);
#undef K3_GO
    UPS_CHECK_LAUNCH("conv3x3_ksw_f32_kernel");
    ups_set_form("conv3x3_ksw<%d,32,%d>", 16 * rb, nw);
    return 0;
}
