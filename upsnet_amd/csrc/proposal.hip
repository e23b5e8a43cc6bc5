// proposal.hip -- RPN proposal generation entirely on the device (gfx950).
//
// Reference: upsnet/operators/functions/pyramid_proposal.py:62-222 copies every level's scores and
// deltas to the host (10.5 MB / image), runs numpy top-k / decode / clip there and calls gpu_nms five
// times with a malloc + H2D + D2H each; modules/pyramid_proposal.py:61-67 then re-sorts on the GPU.
//
// Here: (1) prop_key_kernel turns every anchor score into a unique sortable 64-bit key in the
// reference's (h, w, a) enumeration order; (2) an exact radix SELECT finds each level's k-th largest key
// (k = pre_nms_top_n): three grid-wide digit passes (3 x 11 bits: the score and the top index bit) of many small
// workgroups -- 8 KiB LDS histogram each, merged with global atomics -- with a one-workgroup-per-level pick kernel
// in between, and one launch (prop_tail_select_kernel) for the remaining 31 index bits, which has work to do only
// when several keys share the boundary score (r03-r10: six pass pairs, the last three no-ops in the common case);
// the k survivors are compacted; (3) prop_sort_decode_kernel sorts them (one 1024-key LDS bitonic sort per level:
// rule (ii) of the oracle, score desc / anchor index asc), applies bbox_transform + clip_boxes + the size filter to
// the survivors only and writes the NMS's visiting order straight into its workspace; (4) the mask + scan half of the
// batched NMS of nms.hip handles all levels at once; (5) prop_merge_kernel concatenates the kept boxes per level
// and ranks them (score desc, concatenation index asc). 13 launches (r10: 21). No host synchronisation anywhere.
#include "common.h"
#include "roi_order.h"
#include "sort.h"
#include "upsnet_hip.h"

#define PROP_CH 8192  // keys per tournament chunk (64 KiB of LDS)
#define PROP_MAXLEV 8

int ups_nms_batched_impl(hipStream_t st, const float *boxes, const float *scores, const int *counts,
                         const uint8_t *pre_removed, int P, int nmax, float thresh, int tie_mode, int *keep_idx,
                         int *keep_cnt, void *workspace, int ge);
void ups_nms_ws_views(void *workspace, int P, int nmax, float4 **sorted_boxes, int **order);
int ups_nms_batched_presorted_impl(hipStream_t st, const int *counts, const uint8_t *pre_removed, int P, int nmax, float thresh,
                                   int *keep_idx, int *keep_cnt, void *workspace, int ge);

struct PropLevels {
    const float *cls[PROP_MAXLEV];
    const float *box[PROP_MAXLEV];
    long cls_cs[PROP_MAXLEV], cls_ps[PROP_MAXLEV];   // element strides of the score map: per anchor channel, per pixel
    long box_cs[PROP_MAXLEV], box_ps[PROP_MAXLEV];   // ... of the delta map
    int H[PROP_MAXLEV], W[PROP_MAXLEV], stride[PROP_MAXLEV];
    int n[PROP_MAXLEV];        // anchors per level
    long key_off[PROP_MAXLEV]; // offset of the level's key segment
    float anchors[PROP_MAXLEV][4][4]; // up to 4 anchors per cell
    int nlev, A;
};

// selection state of one level
struct PropSel {
    ups_u64 prefix;      // decided high bits of the k-th largest key (lower bits zero)
    unsigned need;       // how many keys are still to be taken from the keys that match `prefix`
    unsigned done;       // threshold final: every key >= prefix is selected
    unsigned cnt;        // compaction counter
    unsigned pad;
};

#define PROP_BINS 2048

// scores (anchor a, pixel) at cls[a * cls_cs + pixel * cls_ps] (NCHW: cs = H*W, ps = 1; a channel slice of an NHWC map: cs = 1,
// ps = channels of the map) -> keys at (h*W + w)*A + a
__global__ void __launch_bounds__(256)
prop_key_kernel(const PropLevels lv, ups_u64 *__restrict__ keys, PropSel *__restrict__ sel, const int k)
{
    const int l = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PropSel st;
        st.prefix = 0; st.need = (unsigned)k; st.done = 0; st.cnt = 0; st.pad = 0;
        sel[l] = st;
    }
    const int n = lv.n[l], A = lv.A;
    const long hw = (long)lv.H[l] * lv.W[l];
    const float *__restrict__ s = lv.cls[l];
    ups_u64 *__restrict__ out = keys + lv.key_off[l];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)blockDim.x * gridDim.x) {
        const int a = (int)(i / hw);
        const long pix = i % hw;
        const unsigned idx = (unsigned)(pix * A + a);
        out[idx] = ups_make_key(s[a * lv.cls_cs[l] + pix * lv.cls_ps[l]], idx, 1);
    }
}

// digit histogram of the keys that still match the decided prefix; hi = shift + bits of this pass (64 in pass 0)
__global__ void __launch_bounds__(256)
prop_hist_kernel(const PropLevels lv, const ups_u64 *__restrict__ keys, const PropSel *__restrict__ sel, unsigned *__restrict__ hist,
                 const int shift, const int hi)
{
    __shared__ unsigned s_hist[PROP_BINS];
    const int l = blockIdx.y;
    const PropSel st = sel[l];
    if (st.done) return;
    for (int i = threadIdx.x; i < PROP_BINS; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const int n = lv.n[l];
    const ups_u64 *__restrict__ src = keys + lv.key_off[l];
    const unsigned mask = (1u << (hi - shift)) - 1u;
    // (r11: a thread's keys -- 8 for the largest level at the launch's ~8 keys per thread -- are LOADED first, all in flight together, then
    // counted: with the LDS atomic between two loads the loop paid one L2 round trip per key)
    const long stride = (long)blockDim.x * gridDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += 8 * stride) {
        ups_u64 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = (i + u * stride < n) ? src[i + u * stride] : 0ULL;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i + u * stride < n && (hi >= 64 || ((kk[u] ^ st.prefix) >> hi) == 0)) atomicAdd(&s_hist[(unsigned)(kk[u] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PROP_BINS; i += blockDim.x) {
        const unsigned v = s_hist[i];
        if (v) atomicAdd(&hist[(long)l * PROP_BINS + i], v);
    }
}

// one workgroup per level: the bin that holds the k-th largest key among the matching ones; clears the histogram
__global__ void __launch_bounds__(256)
prop_pick_kernel(PropSel *__restrict__ sel, unsigned *__restrict__ hist, const int shift, const int last)
{
    __shared__ unsigned s_hist[PROP_BINS], s_part[256];
    const int l = blockIdx.x, tid = threadIdx.x;
    unsigned *h = hist + (long)l * PROP_BINS;
    PropSel st = sel[l];
    unsigned part = 0;
    for (int q = 0; q < PROP_BINS / 256; ++q) {
        const unsigned v = h[tid * (PROP_BINS / 256) + q];
        s_hist[tid * (PROP_BINS / 256) + q] = v;
        part += v;
        h[tid * (PROP_BINS / 256) + q] = 0;
    }
    s_part[tid] = part;
    __syncthreads();
    if (st.done) return;
    unsigned above = 0;   // keys in the bins of higher threads
    for (int u = tid + 1; u < 256; ++u) above += s_part[u];
    if (tid == 0 && above + part < st.need) {   // fewer matching keys than needed (level smaller than k): take everything
        st.done = 1;
        sel[l] = st;
    }
    if (above < st.need && st.need <= above + part) {   // the wanted key is in one of this thread's bins
        unsigned cum = above;
        for (int q = PROP_BINS / 256 - 1; q >= 0; --q) {
            const int bin = tid * (PROP_BINS / 256) + q;
            const unsigned v = s_hist[bin];
            if (cum + v >= st.need) {
                st.prefix |= (ups_u64)bin << shift;
                st.need -= cum;
                if (v == st.need || last) st.done = 1;   // the whole bin is wanted: all lower bits are free
                sel[l] = st;
                break;
            }
            cum += v;
        }
    }
}

// r11: the last three digit passes (bits 30..0: the low index bits of the key) as ONE launch, one workgroup per level. After the first
// three passes the 33 leading bits -- the whole score and the top index bit -- of the k-th largest key are decided, and unless several
// keys share the boundary SCORE the level is already `done` (the bin holds exactly the keys still needed): this kernel then returns at
// once, where r03-r10 issued six more launches that did the same. With ties at the boundary it finishes the selection itself -- each
// pass one sweep of the level's keys by this one workgroup (histogram in LDS, the same pick rule): slower than the grid-wide passes
// (~10 us per pass on the 393 216 keys of P2), exact all the same, and rare (a saturated sigmoid).
__global__ void __launch_bounds__(1024)
prop_tail_select_kernel(const PropLevels lv, const ups_u64 *__restrict__ keys, PropSel *__restrict__ sel)
{
    __shared__ unsigned s_hist[PROP_BINS], s_part[256];
    __shared__ PropSel s_st;
    const int l = blockIdx.x, tid = threadIdx.x;
    PropSel st = sel[l];
    if (st.done) return;
    const int n = lv.n[l];
    const ups_u64 *__restrict__ src = keys + lv.key_off[l];
    const int shifts[3] = {20, 9, 0}, his[3] = {31, 20, 9};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], hi = his[pass];
        for (int i = tid; i < PROP_BINS; i += 1024) s_hist[i] = 0;
        __syncthreads();
        const unsigned mask = (1u << (hi - shift)) - 1u;
        for (int i = tid; i < n; i += 1024) {
            const ups_u64 key = src[i];
            if (((key ^ st.prefix) >> hi) == 0) atomicAdd(&s_hist[(unsigned)(key >> shift) & mask], 1u);
        }
        __syncthreads();
        // the pick rule of prop_pick_kernel on the LDS histogram (threads 0..255: 8 bins each)
        unsigned part = 0;
        if (tid < 256) {
            for (int q = 0; q < PROP_BINS / 256; ++q) part += s_hist[tid * (PROP_BINS / 256) + q];
            s_part[tid] = part;
        }
        if (tid == 0) s_st = st;
        __syncthreads();
        if (tid < 256) {
            unsigned above = 0;
            for (int u = tid + 1; u < 256; ++u) above += s_part[u];
            if (tid == 0 && above + part < st.need) {
                PropSel t = st;
                t.done = 1;
                s_st = t;
            }
            if (above < st.need && st.need <= above + part) {
                unsigned cum = above;
                for (int q = PROP_BINS / 256 - 1; q >= 0; --q) {
                    const int bin = tid * (PROP_BINS / 256) + q;
                    const unsigned v = s_hist[bin];
                    if (cum + v >= st.need) {
                        PropSel t = st;
                        t.prefix |= (ups_u64)bin << shift;
                        t.need -= cum;
                        if (v == t.need || pass == 2) t.done = 1;
                        s_st = t;
                        break;
                    }
                    cum += v;
                }
            }
        }
        __syncthreads();
        st = s_st;
        if (st.done) break;
        __syncthreads();     // (s_st / s_hist are rewritten by the next pass)
    }
    if (tid == 0) sel[l] = st;
}

// survivors (key >= threshold) -> out[l][0..k), unordered. r11: ONE global atomic per workgroup that has survivors (r03-r10: one per wave
// and key round -- ~1000 atomics on the same word per level, serialised in L2: 20 us for a kernel that reads 4 MB): a thread's 8 keys are
// loaded first, its takes counted, the four waves' counts combined through LDS, thread 0 reserves the workgroup's range, and every
// thread writes its takes at (base + takes of lower waves + takes of lower lanes + own earlier takes).
__global__ void __launch_bounds__(256)
prop_compact_kernel(const PropLevels lv, const ups_u64 *__restrict__ keys, PropSel *__restrict__ sel, ups_u64 *__restrict__ out,
                    const int k)
{
    __shared__ unsigned s_wave[4], s_base;
    const int l = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lv.n[l];
    const ups_u64 thr = sel[l].prefix;
    const ups_u64 *__restrict__ src = keys + lv.key_off[l];
    ups_u64 *__restrict__ dst = out + lv.key_off[l];
    const long stride = (long)blockDim.x * gridDim.x;
    for (long i0 = (long)blockIdx.x * blockDim.x; i0 < n; i0 += 8 * stride) {
        ups_u64 kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long i = i0 + u * stride + threadIdx.x; kk[u] = i < n ? src[i] : 0ULL; }
        unsigned mine = 0;                                   // bit u: key u of this thread is a survivor
#pragma unroll
        for (int u = 0; u < 8; ++u) mine |= (kk[u] != 0ULL && kk[u] >= thr) ? (1u << u) : 0u;
        const unsigned cnt = (unsigned)__builtin_popcount(mine);
        // exclusive prefix of the per-thread counts inside the wave (counts <= 8: 6 shuffle steps), wave totals through LDS
        unsigned incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        const unsigned w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
        const unsigned total = w0 + w1 + w2 + w3;
        if (total) {                                         // (uniform over the workgroup)
            if (threadIdx.x == 0) s_base = atomicAdd(&sel[l].cnt, total);
            __syncthreads();
            unsigned pos = s_base + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + (incl - cnt);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (mine & (1u << u)) { if (pos < (unsigned)k) dst[pos] = kk[u]; ++pos; }
        }
        __syncthreads();                                     // (s_wave / s_base are rewritten by the next round)
    }
}

// one workgroup per level: sort the (<= k) survivors descending, zero-pad to k
__global__ void __launch_bounds__(1024)
prop_sortk_kernel(const PropLevels lv, const PropSel *__restrict__ sel, ups_u64 *__restrict__ out, const int k, const int M)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ups_u64 *keys = reinterpret_cast<ups_u64 *>(smem_raw);
    const int l = blockIdx.x;
    const int c = (int)min(sel[l].cnt, (unsigned)k);
    ups_u64 *__restrict__ buf = out + lv.key_off[l];
    for (int i = threadIdx.x; i < M; i += blockDim.x) keys[i] = i < c ? buf[i] : 0ULL;
    ups_block_sort_desc(keys, M);     // (one key per thread at M = 1024: 34 -> ~16 us vs the 256-thread launch of r07)
    for (int i = threadIdx.x; i < k; i += blockDim.x) buf[i] = keys[i];
}

// bbox_transform + clip_boxes of anchor `idx` ((h, w, a) enumeration) of level l (functions/pyramid_proposal.py:83-97,132-135)
__device__ static inline void prop_decode_anchor(const PropLevels &lv, const int l, const unsigned idx, const float im_h, const float im_w,
                                                 float o[4])
{
    const int A = lv.A, W = lv.W[l];
    const int a = idx % A;
    const int pix = idx / A;
    const int h = pix / W, w = pix % W;
    const float sx = (float)(w * lv.stride[l]), sy = (float)(h * lv.stride[l]);
    const float ax1 = lv.anchors[l][a][0] + sx, ay1 = lv.anchors[l][a][1] + sy;
    const float ax2 = lv.anchors[l][a][2] + sx, ay2 = lv.anchors[l][a][3] + sy;
    const long bcs = lv.box_cs[l];
    const float *__restrict__ d = lv.box[l] + (long)(a * 4) * bcs + (long)pix * lv.box_ps[l];
    ups_decode_clip(ax1, ay1, ax2, ay2, d[0], d[bcs], d[2 * bcs], d[3 * bcs], 1.f, 1.f, 1.f, 1.f, im_h, im_w, true, o);
}

// r11: sort + decode + NMS visiting order in ONE launch per level set (r03-r10: prop_sortk_kernel, prop_decode_kernel and the NMS's own
// nms_sort_kernel -- which sorted by score a list that IS sorted by score). One workgroup per level: (1) LDS bitonic sort of the <= k
// survivors (descending keys = score descending, anchor index ascending: rule (ii)); (2) thread i decodes entry i (bbox_transform + clip +
// size filter, the arithmetic of prop_decode_kernel) into boxes / scores / pre_removed; (3) the order gpu_nms would visit them in --
// `scores.argsort()[::-1]`, rule (i): equal scores HIGHER list index first -- is the list itself with every run of equal scores
// reversed: entry i of the run [a, b] goes to position a + b - i (two binary searches in the sorted LDS keys: no scan, no barrier);
// order[position] = i and sorted_boxes[position] = box i are written straight into the NMS workspace.
__global__ void __launch_bounds__(1024)
prop_sort_decode_kernel(const PropLevels lv, const PropSel *__restrict__ sel, const ups_u64 *__restrict__ cand, const int k, const int M,
                        const float *__restrict__ im_info, const float min_size, float *__restrict__ boxes, float *__restrict__ scores,
                        uint8_t *__restrict__ pre_removed, int *__restrict__ counts, float4 *__restrict__ sorted_boxes,
                        int *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ups_u64 *keys = reinterpret_cast<ups_u64 *>(smem_raw);
    const int l = blockIdx.x;
    const int c = (int)min(sel[l].cnt, (unsigned)k);      // = min(k, anchors of the level)
    const ups_u64 *__restrict__ buf = cand + lv.key_off[l];
    for (int i = threadIdx.x; i < M; i += blockDim.x) keys[i] = i < c ? buf[i] : 0ULL;
    ups_block_sort_desc(keys, M);
    if (threadIdx.x == 0) counts[l] = c;
    const float im_h = im_info[0], im_w = im_info[1], ms = min_size * im_info[2];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const long row = (long)l * k + i;
        float *b = boxes + row * 4;
        if (i >= c) {
            b[0] = b[1] = b[2] = b[3] = 0.f;
            scores[row] = 0.f;
            pre_removed[row] = 1;
            continue;
        }
        const ups_u64 key = keys[i];
        float o[4];
        prop_decode_anchor(lv, l, ups_key_index(key, 1), im_h, im_w, o);
        b[0] = o[0]; b[1] = o[1]; b[2] = o[2]; b[3] = o[3];
        scores[row] = ups_key_score(key);
        const float ws = o[2] - o[0] + 1.0f, hs = o[3] - o[1] + 1.0f;
        pre_removed[row] = !((ws >= ms) && (hs >= ms));
        // run [a, bnd) of entries with this entry's score bits in the descending list
        const unsigned sb = (unsigned)(key >> 32);
        int lo = 0, hi = i;                                 // first position whose score bits are <= sb (they are == sb there)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned)(keys[mid] >> 32) > sb) lo = mid + 1; else hi = mid; }
        const int a = lo;
        lo = i; hi = c;                                     // first position whose score bits are < sb
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned)(keys[mid] >> 32) >= sb) lo = mid + 1; else hi = mid; }
        const int pos = a + (lo - 1) - i;
        order[(long)l * k + pos] = i;
        sorted_boxes[(long)l * k + pos] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// single workgroup: concatenate per-level kept boxes (<= post_n each), rank by (score desc, concat idx asc), keep the first post_n.
// Every level's kept list is already in that order (the NMS keeps in visiting order = score descending, the concatenation index grows
// along the list), so the ranking is a MERGE of nlev sorted runs, not a sort: the rank of an element is its position in its own run
// plus, for every other run, the number of keys there that precede it -- one binary search per other run (nlev - 1 searches of ~10
// LDS reads), all elements in parallel, no barrier after the key gather. (r01-r07: an 8192-key bitonic sort, 91 barrier-separated
// compare-exchange steps = 59-65 us; this is ~10.) Unique keys (score bits << 32 | ~index): same order as the sort, bit for bit.
__global__ void __launch_bounds__(1024)
prop_merge_kernel(const int nlev, const int pre_n, const int post_n, const float *__restrict__ boxes,
                  const float *__restrict__ scores, const int *__restrict__ keep_idx, const int *__restrict__ keep_cnt,
                  float *__restrict__ rois_out, float *__restrict__ scores_out, int *__restrict__ num_out, int *__restrict__ roi_order,
                  const float *__restrict__ im_info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ups_u64 *keys = reinterpret_cast<ups_u64 *>(smem_raw);
    __shared__ int s_start[PROP_MAXLEV + 1];
    if (threadIdx.x == 0) {
        int t = 0;
        for (int l = 0; l < nlev; ++l) { s_start[l] = t; t += min(keep_cnt[l], post_n); }
        s_start[nlev] = t;
    }
    __syncthreads();
    const int total = s_start[nlev];
    const int nout = min(total, post_n);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int l = 0;
        for (int q = 1; q < nlev; ++q) if (i >= s_start[q]) l = q;
        const int src = keep_idx[(long)l * pre_n + (i - s_start[l])];
        keys[i] = ups_make_key(scores[(long)l * pre_n + src], (unsigned)i, 1);
    }
    // rows beyond the merged count: defined output (zeros)
    for (int i = nout + threadIdx.x; i < post_n; i += blockDim.x) {
        float *r = rois_out + (long)i * 5;
        r[0] = r[1] = r[2] = r[3] = r[4] = 0.f;
        scores_out[i] = 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int l = 0;
        for (int q = 1; q < nlev; ++q) if (i >= s_start[q]) l = q;
        const ups_u64 key = keys[i];
        int rank = i - s_start[l];
        for (int q = 0; q < nlev; ++q) {
            if (q == l) continue;
            int lo = s_start[q], hi = s_start[q + 1];         // first position of run q whose key does NOT precede `key`
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[mid] > key) lo = mid + 1; else hi = mid;
            }
            rank += lo - s_start[q];
        }
        if (rank < post_n) {
            const int src = keep_idx[(long)l * pre_n + (i - s_start[l])];
            const float *b = boxes + ((long)l * pre_n + src) * 4;
            float *r = rois_out + (long)rank * 5;
            r[0] = 0.f; r[1] = b[0]; r[2] = b[1]; r[3] = b[2]; r[4] = b[3];
            scores_out[rank] = scores[(long)l * pre_n + src];
        }
    }
    if (threadIdx.x == 0) *num_out = nout;
    // r13: the ROI -> XCD dealing table of the box head's ROIAlign launch (roi_order.h), here because the ranked rois exist here: no launch of its own
    if (roi_order != nullptr && post_n <= ROI_ORDER_MAX) {
        __syncthreads();                     // rois_out complete and visible to the workgroup; the key buffer is free: its first ints become the counters
        ups_roi_order_block(rois_out, post_n, nout, 16.0f / im_info[0], 8.0f / im_info[1], roi_order, reinterpret_cast<int *>(smem_raw));
    }
}


// ---------------------------------------------------------------------------------------------
// individual_proposals = False (functions/pyramid_proposal.py:181-208; the constructor's default, :26): no per-level top-k and
// no per-level NMS -- EVERY anchor of every level is decoded, clipped and size-filtered (:132-141), the survivors of all levels are
// concatenated (:176-177), ranked jointly (`scores.argsort()[::-1]`: rule (i), equal scores -> higher concatenation index first),
// the first pre_nms_top_n go through ONE NMS and the first post_nms_top_n kept boxes are the result. Device form: the key of an
// anchor that fails the size filter is 0 (= absent), all keys lie in ONE segment in concatenation order (level-major, (h, w, a)
// inside a level; the filter removes rows but keeps their order, so the index before the filter ranks ties like the index after
// it), and the selection / sort machinery above runs on that segment as a single "level".
struct PropJoint {
    long goff[PROP_MAXLEV + 1];   // first key of each level in the concatenation
};

__global__ void __launch_bounds__(256)
prop_key_joint_kernel(const PropLevels lv, const PropJoint jt, ups_u64 *__restrict__ keys, PropSel *__restrict__ sel, const int k,
                      const float *__restrict__ im_info, const float min_size)
{
    const int l = blockIdx.y;
    if (blockIdx.x == 0 && l == 0 && threadIdx.x == 0) {
        PropSel st;
        st.prefix = 0; st.need = (unsigned)k; st.done = 0; st.cnt = 0; st.pad = 0;
        sel[0] = st;
    }
    const int n = lv.n[l], A = lv.A;
    const long hw = (long)lv.H[l] * lv.W[l];
    const float *__restrict__ s = lv.cls[l];
    const float im_h = im_info[0], im_w = im_info[1], ms = min_size * im_info[2];
    ups_u64 *__restrict__ out = keys + jt.goff[l];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)blockDim.x * gridDim.x) {
        const int a = (int)(i / hw);
        const long pix = i % hw;
        const unsigned idx = (unsigned)(pix * A + a);
        float o[4];
        prop_decode_anchor(lv, l, idx, im_h, im_w, o);
        const float ws = o[2] - o[0] + 1.0f, hs = o[3] - o[1] + 1.0f;
        const bool ok = (ws >= ms) && (hs >= ms);
        out[idx] = ok ? ups_make_key(s[a * lv.cls_cs[l] + pix * lv.cls_ps[l]], (unsigned)(jt.goff[l] + idx), 0) : 0ULL;
    }
}

// the sorted top-pre_n keys of the joint segment -> boxes / scores of the one NMS problem
__global__ void __launch_bounds__(256)
prop_decode_joint_kernel(const PropLevels lv, const PropJoint jt, const ups_u64 *__restrict__ keys, const PropSel *__restrict__ sel,
                         const int pre_n, const float *__restrict__ im_info, float *__restrict__ boxes, float *__restrict__ scores,
                         int *__restrict__ counts)
{
    const int cnt = (int)min(sel[0].cnt, (unsigned)pre_n);
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = cnt;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pre_n) return;
    float *b = boxes + (long)i * 4;
    const ups_u64 key = i < cnt ? keys[i] : 0ULL;
    if (key == 0ULL) {
        b[0] = b[1] = b[2] = b[3] = 0.f;
        scores[i] = 0.f;
        return;
    }
    const long g = (long)ups_key_index(key, 0);
    int l = 0;
    for (int q = 1; q < lv.nlev; ++q) if (g >= jt.goff[q]) l = q;
    float o[4];
    prop_decode_anchor(lv, l, (unsigned)(g - jt.goff[l]), im_info[0], im_info[1], o);
    b[0] = o[0]; b[1] = o[1]; b[2] = o[2]; b[3] = o[3];
    scores[i] = ups_key_score(key);
}

// rows [0, min(kept, post_n)) = the kept boxes in NMS visiting order; zero rows behind; the count
__global__ void __launch_bounds__(256)
prop_emit_joint_kernel(const int post_n, const float *__restrict__ boxes, const float *__restrict__ scores,
                       const int *__restrict__ keep_idx, const int *__restrict__ keep_cnt, float *__restrict__ rois_out,
                       float *__restrict__ scores_out, int *__restrict__ num_out)
{
    const int nout = min(keep_cnt[0], post_n);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *num_out = nout;
    if (i >= post_n) return;
    float *r = rois_out + (long)i * 5;
    if (i < nout) {
        const int src = keep_idx[i];
        const float *b = boxes + (long)src * 4;
        r[0] = 0.f; r[1] = b[0]; r[2] = b[1]; r[3] = b[2]; r[4] = b[3];
        scores_out[i] = scores[src];
    } else {
        r[0] = r[1] = r[2] = r[3] = r[4] = 0.f;
        scores_out[i] = 0.f;
    }
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct PropPlan {
    long key_total;      // keys per ping-pong buffer
    size_t off_keys0, off_keys1, off_boxes, off_scores, off_pre, off_counts, off_keep, off_keepcnt, off_hist, off_sel, off_nms, total;
};

static PropPlan prop_plan(int nlev, const int *H, const int *W, int A, int pre_n)
{
    PropPlan p;
    long tot = 0;
    for (int l = 0; l < nlev; ++l) {
        long n = (long)H[l] * W[l] * A;
        long chunks = (n + PROP_CH - 1) / PROP_CH;
        long need = n > chunks * pre_n ? n : chunks * pre_n;
        tot += need + pre_n;
    }
    p.key_total = tot;
    size_t o = 0;
    p.off_keys0 = o; o += al256((size_t)tot * 8);
    p.off_keys1 = o; o += al256((size_t)tot * 8);
    p.off_boxes = o; o += al256((size_t)nlev * pre_n * 16);
    p.off_scores = o; o += al256((size_t)nlev * pre_n * 4);
    p.off_pre = o; o += al256((size_t)nlev * pre_n);
    p.off_counts = o; o += 256;
    p.off_keep = o; o += al256((size_t)nlev * pre_n * 4);
    p.off_keepcnt = o; o += 256;
    p.off_hist = o; o += al256((size_t)nlev * PROP_BINS * 4);
    p.off_sel = o; o += al256((size_t)nlev * sizeof(PropSel));
    p.off_nms = o; o += upsnet_nms_workspace_bytes(nlev, pre_n);
    p.total = o + 256;
    return p;
}

extern "C" size_t upsnet_proposal_workspace_bytes(int nlev, const int *heights, const int *widths, int num_anchors,
                                                  int pre_nms_top_n, int post_nms_top_n)
{
    (void)post_nms_top_n;
    if (nlev <= 0 || nlev > PROP_MAXLEV || !heights || !widths) return 0;
    return prop_plan(nlev, heights, widths, num_anchors, pre_nms_top_n).total;
}

// fills the per-level descriptors (pointers, strides, sizes, anchors, key segments); returns the largest level
static int prop_fill_levels(PropLevels &lv, int nlev, const float *const cls_prob[], const float *const bbox_pred[], const long *cls_cs,
                            const long *cls_ps, const long *box_cs, const long *box_ps, const int *heights, const int *widths,
                            const int *strides, const float *anchors, int num_anchors, int pre_n)
{
    lv.nlev = nlev; lv.A = num_anchors;
    long off = 0;
    int maxn = 0;
    for (int l = 0; l < nlev; ++l) {
        if (!(cls_prob[l] && bbox_pred[l] && heights[l] > 0 && widths[l] > 0)) return -1 - l;
        lv.cls[l] = cls_prob[l]; lv.box[l] = bbox_pred[l];
        const long hw_ = (long)heights[l] * widths[l];
        lv.cls_cs[l] = cls_cs ? cls_cs[l] : hw_; lv.cls_ps[l] = cls_ps ? cls_ps[l] : 1;
        lv.box_cs[l] = box_cs ? box_cs[l] : hw_; lv.box_ps[l] = box_ps ? box_ps[l] : 1;
        lv.H[l] = heights[l]; lv.W[l] = widths[l]; lv.stride[l] = strides[l];
        lv.n[l] = heights[l] * widths[l] * num_anchors;
        for (int a = 0; a < num_anchors; ++a)
            for (int q = 0; q < 4; ++q) lv.anchors[l][a][q] = anchors[((long)l * num_anchors + a) * 4 + q];
        long n = lv.n[l];
        long chunks = (n + PROP_CH - 1) / PROP_CH;
        long need = n > chunks * pre_n ? n : chunks * pre_n;
        lv.key_off[l] = off;
        off += need + pre_n;
        if (lv.n[l] > maxn) maxn = lv.n[l];
    }
    return maxn;
}

// exact radix select of every level's pre_n-th largest key (64 bits: digits of 11, 11, 11 bits grid-wide, the remaining 31 in
// prop_tail_select_kernel) and compaction of the survivors into `out` (unordered, sel[l].cnt of them)
static int prop_select_compact(hipStream_t st, const PropLevels &lv, int nlev, int maxn, int pre_n, ups_u64 *keys, ups_u64 *out, unsigned *hist,
                               PropSel *sel)
{
    int gh = (maxn + 256 * 8 - 1) / (256 * 8);   // ~8 keys per thread
    if (gh > 512) gh = 512;
    if (gh < 1) gh = 1;
    static const int digit_shift[3] = {53, 42, 31}, digit_hi[3] = {64, 53, 42};
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(prop_hist_kernel, dim3(gh, nlev), dim3(256), 0, st, lv, keys, sel, hist, digit_shift[pass], digit_hi[pass]);
        UPS_CHECK_LAUNCH("prop_hist_kernel");
        hipLaunchKernelGGL(prop_pick_kernel, dim3(nlev), dim3(256), 0, st, sel, hist, digit_shift[pass], 0);
        UPS_CHECK_LAUNCH("prop_pick_kernel");
    }
    hipLaunchKernelGGL(prop_tail_select_kernel, dim3(nlev), dim3(1024), 0, st, lv, keys, sel);   // bits 30..0: a no-op unless scores tie at the boundary
    UPS_CHECK_LAUNCH("prop_tail_select_kernel");
    hipLaunchKernelGGL(prop_compact_kernel, dim3(gh, nlev), dim3(256), 0, st, lv, keys, sel, out, pre_n);
    UPS_CHECK_LAUNCH("prop_compact_kernel");
    return 0;
}

// dynamic LDS of the per-level sort kernels: M2 = pre_n rounded up to a power of two (>= 64) keys of 8 bytes; > 64 KiB is opted into
// (templated on the kernel's VALUE: each kernel gets its own once-per-device flag, whatever its signature)
template <auto KERNEL>
static int prop_sort_lds(int pre_n, int *M2_out)
{
    constexpr auto kernel = KERNEL;
    const int M2 = ups_next_pow2(pre_n < 64 ? 64 : pre_n);
    *M2_out = M2;
    if ((size_t)M2 * 8 > 64 * 1024) {
        static std::atomic<unsigned long long> attr_dev{0};
        UPS_ONCE_PER_DEVICE(attr_dev, UPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                        PROP_CH * 8)));
    }
    return 0;
}

extern "C" int upsnet_pyramid_proposals_strided_ordered(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                        const long *cls_cs, const long *cls_ps, const long *box_cs, const long *box_ps, const int *heights, const int *widths, const int *strides, const float *anchors,
                                        int num_anchors, const float *im_info, int pre_n, int post_n, float nms_thresh,
                                        float min_size, float *rois_out, float *scores_out, int *num_out, void *workspace, int *roi_order_out);

extern "C" int upsnet_pyramid_proposals_strided(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                        const long *cls_cs, const long *cls_ps, const long *box_cs, const long *box_ps, const int *heights, const int *widths, const int *strides, const float *anchors,
                                        int num_anchors, const float *im_info, int pre_n, int post_n, float nms_thresh,
                                        float min_size, float *rois_out, float *scores_out, int *num_out, void *workspace)
{
    return upsnet_pyramid_proposals_strided_ordered(stream, nlev, cls_prob, bbox_pred, cls_cs, cls_ps, box_cs, box_ps, heights, widths, strides, anchors, num_anchors,
                                                    im_info, pre_n, post_n, nms_thresh, min_size, rois_out, scores_out, num_out, workspace, nullptr);
}

/* upsnet_pyramid_proposals_strided + (r13) roi_order_out: int[post_n] or NULL -- the workgroup -> ROI table of upsnet_fpn_roi_align_forward_ordered for
 * these rois (see upsnet_fpn_roi_order), written by the launch that ranks them; post_n <= 2048, else the table is not written (pass NULL). */
extern "C" int upsnet_pyramid_proposals_strided_ordered(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                        const long *cls_cs, const long *cls_ps, const long *box_cs, const long *box_ps, const int *heights, const int *widths, const int *strides, const float *anchors,
                                        int num_anchors, const float *im_info, int pre_n, int post_n, float nms_thresh,
                                        float min_size, float *rois_out, float *scores_out, int *num_out, void *workspace, int *roi_order_out)
{
    UPS_REQUIRE(roi_order_out == nullptr || post_n <= ROI_ORDER_MAX, "pyramid_proposals: the ROI order table needs post_nms_top_n <= %d", ROI_ORDER_MAX);
    UPS_REQUIRE(nlev >= 1 && nlev <= PROP_MAXLEV, "pyramid_proposals: nlev must be 1..%d", PROP_MAXLEV);
    UPS_REQUIRE(cls_prob && bbox_pred && heights && widths && strides && anchors && im_info && rois_out && scores_out &&
                    num_out && workspace, "pyramid_proposals: null pointer");
    UPS_REQUIRE(num_anchors >= 1 && num_anchors <= 4, "pyramid_proposals: 1..4 anchors per cell supported (got %d)", num_anchors);
    UPS_REQUIRE(pre_n >= 1 && pre_n <= PROP_CH, "pyramid_proposals: pre_nms_top_n must be 1..%d", PROP_CH);
    UPS_REQUIRE(post_n >= 1 && (long)nlev * (post_n < pre_n ? post_n : pre_n) <= PROP_CH,
                "pyramid_proposals: nlev*post_nms_top_n must be <= %d", PROP_CH);
    hipStream_t st = (hipStream_t)stream;
    const PropPlan plan = prop_plan(nlev, heights, widths, num_anchors, pre_n);
    unsigned char *ws = (unsigned char *)workspace;
    ups_u64 *kbuf[2] = {(ups_u64 *)(ws + plan.off_keys0), (ups_u64 *)(ws + plan.off_keys1)};
    float *boxes = (float *)(ws + plan.off_boxes);
    float *scores = (float *)(ws + plan.off_scores);
    uint8_t *pre_removed = ws + plan.off_pre;
    int *counts = (int *)(ws + plan.off_counts);
    int *keep_idx = (int *)(ws + plan.off_keep);
    int *keep_cnt = (int *)(ws + plan.off_keepcnt);
    void *nms_ws = ws + plan.off_nms;

    PropLevels lv;
    const int maxn = prop_fill_levels(lv, nlev, cls_prob, bbox_pred, cls_cs, cls_ps, box_cs, box_ps, heights, widths, strides, anchors,
                                      num_anchors, pre_n);
    UPS_REQUIRE(maxn > 0, "pyramid_proposals: bad level %d", -1 - maxn);
    int gx = (maxn + 255) / 256;
    if (gx > 1024) gx = 1024;
    unsigned *hist = (unsigned *)(ws + plan.off_hist);
    PropSel *sel = (PropSel *)(ws + plan.off_sel);
    if (ups_zero_async(hist, (size_t)nlev * PROP_BINS * 4, st)) return 1;
    hipLaunchKernelGGL(prop_key_kernel, dim3(gx, nlev), dim3(256), 0, st, lv, kbuf[0], sel, pre_n);
    UPS_CHECK_LAUNCH("prop_key_kernel");

    if (int rc = prop_select_compact(st, lv, nlev, maxn, pre_n, kbuf[0], kbuf[1], hist, sel)) return rc;
    // sort + decode + the NMS's visiting order in one launch (the survivors of every level at key_off[l] of kbuf[1]); then mask + scan
    int M2;
    if (int rc = prop_sort_lds<&prop_sort_decode_kernel>(pre_n, &M2)) return rc;
    float4 *nms_sorted_boxes;
    int *nms_order;
    ups_nms_ws_views(nms_ws, nlev, pre_n, &nms_sorted_boxes, &nms_order);
    hipLaunchKernelGGL(prop_sort_decode_kernel, dim3(nlev), dim3(M2 < 1024 ? M2 : 1024), (size_t)M2 * 8, st, lv, sel, kbuf[1], pre_n, M2, im_info,
                       min_size, boxes, scores, pre_removed, counts, nms_sorted_boxes, nms_order);
    UPS_CHECK_LAUNCH("prop_sort_decode_kernel");
    int rc = ups_nms_batched_presorted_impl(st, counts, pre_removed, nlev, pre_n, nms_thresh, keep_idx, keep_cnt, nms_ws, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(prop_merge_kernel, dim3(1), dim3(1024), (size_t)PROP_CH * 8, st, nlev, pre_n, post_n, boxes, scores,
                       keep_idx, keep_cnt, rois_out, scores_out, num_out, roi_order_out, im_info);
    UPS_CHECK_LAUNCH("prop_merge_kernel");
    return 0;
}

extern "C" int upsnet_pyramid_proposals_joint_strided(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                        const long *cls_cs, const long *cls_ps, const long *box_cs, const long *box_ps, const int *heights, const int *widths, const int *strides, const float *anchors,
                                        int num_anchors, const float *im_info, int pre_n, int post_n, float nms_thresh,
                                        float min_size, float *rois_out, float *scores_out, int *num_out, void *workspace)
{
    UPS_REQUIRE(nlev >= 1 && nlev <= PROP_MAXLEV, "pyramid_proposals_joint: nlev must be 1..%d", PROP_MAXLEV);
    UPS_REQUIRE(cls_prob && bbox_pred && heights && widths && strides && anchors && im_info && rois_out && scores_out &&
                    num_out && workspace, "pyramid_proposals_joint: null pointer");
    UPS_REQUIRE(num_anchors >= 1 && num_anchors <= 4, "pyramid_proposals_joint: 1..4 anchors per cell supported (got %d)", num_anchors);
    UPS_REQUIRE(pre_n >= 1 && pre_n <= PROP_CH, "pyramid_proposals_joint: pre_nms_top_n must be 1..%d (the reference's <= 0 = unlimited is not supported)", PROP_CH);
    UPS_REQUIRE(post_n >= 1, "pyramid_proposals_joint: post_nms_top_n must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const PropPlan plan = prop_plan(nlev, heights, widths, num_anchors, pre_n);
    unsigned char *ws = (unsigned char *)workspace;
    ups_u64 *kbuf[2] = {(ups_u64 *)(ws + plan.off_keys0), (ups_u64 *)(ws + plan.off_keys1)};
    float *boxes = (float *)(ws + plan.off_boxes);
    float *scores = (float *)(ws + plan.off_scores);
    int *counts = (int *)(ws + plan.off_counts);
    int *keep_idx = (int *)(ws + plan.off_keep);
    int *keep_cnt = (int *)(ws + plan.off_keepcnt);
    unsigned *hist = (unsigned *)(ws + plan.off_hist);
    PropSel *sel = (PropSel *)(ws + plan.off_sel);

    PropLevels lv;
    const int maxn = prop_fill_levels(lv, nlev, cls_prob, bbox_pred, cls_cs, cls_ps, box_cs, box_ps, heights, widths, strides, anchors,
                                      num_anchors, pre_n);
    UPS_REQUIRE(maxn > 0, "pyramid_proposals_joint: bad level %d", -1 - maxn);
    PropJoint jt;
    long total = 0;
    for (int l = 0; l < nlev; ++l) { jt.goff[l] = total; total += lv.n[l]; }
    jt.goff[nlev] = total;
    UPS_REQUIRE(total < (1L << 31), "pyramid_proposals_joint: %ld anchors exceed the 32-bit key index", total);
    PropLevels one = lv;            // the concatenation as ONE selection problem
    one.nlev = 1; one.n[0] = (int)total; one.key_off[0] = 0;

    if (ups_zero_async(hist, (size_t)PROP_BINS * 4, st)) return 1;
    int gx = (maxn + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(prop_key_joint_kernel, dim3(gx, nlev), dim3(256), 0, st, lv, jt, kbuf[0], sel, pre_n, im_info, min_size);
    UPS_CHECK_LAUNCH("prop_key_joint_kernel");
    if (int rc = prop_select_compact(st, one, 1, (int)total, pre_n, kbuf[0], kbuf[1], hist, sel)) return rc;
    int M2;
    if (int rc = prop_sort_lds<&prop_sortk_kernel>(pre_n, &M2)) return rc;
    hipLaunchKernelGGL(prop_sortk_kernel, dim3(1), dim3(M2 < 1024 ? M2 : 1024), (size_t)M2 * 8, st, one, sel, kbuf[1], pre_n, M2);
    UPS_CHECK_LAUNCH("prop_sortk_kernel");
    hipLaunchKernelGGL(prop_decode_joint_kernel, dim3((pre_n + 255) / 256), dim3(256), 0, st, lv, jt, kbuf[1], sel, pre_n, im_info,
                       boxes, scores, counts);
    UPS_CHECK_LAUNCH("prop_decode_joint_kernel");
    // gpu_nms re-sorts its input (`scores.argsort()[::-1]`, gpu_nms.pyx:33): equal scores are VISITED higher list position first
    int rc = ups_nms_batched_impl(st, boxes, scores, counts, nullptr, 1, pre_n, nms_thresh, 0, keep_idx, keep_cnt, ws + plan.off_nms, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(prop_emit_joint_kernel, dim3((post_n + 255) / 256), dim3(256), 0, st, post_n, boxes, scores, keep_idx, keep_cnt,
                       rois_out, scores_out, num_out);
    UPS_CHECK_LAUNCH("prop_emit_joint_kernel");
    return 0;
}

// NCHW-contiguous inputs ([A,H,W] scores, [4A,H,W] deltas per level): the strided entry with the default strides
extern "C" int upsnet_pyramid_proposals(void *stream, int nlev, const float *const cls_prob[], const float *const bbox_pred[],
                                        const int *heights, const int *widths, const int *strides, const float *anchors,
                                        int num_anchors, const float *im_info, int pre_n, int post_n, float nms_thresh,
                                        float min_size, float *rois_out, float *scores_out, int *num_out, void *workspace)
{
    return upsnet_pyramid_proposals_strided(stream, nlev, cls_prob, bbox_pred, nullptr, nullptr, nullptr, nullptr, heights, widths, strides,
                                            anchors, num_anchors, im_info, pre_n, post_n, nms_thresh, min_size, rois_out, scores_out,
                                            num_out, workspace);
}
