// postprocess.hip -- the step after the hot path: BaseDataset.get_unified_pan_result
// (upsnet/dataset/base_dataset.py:332-371), which turns the network's panoptic label map + semantic argmax map into the
// 2-channel (category, instance) image that panoptic evaluation consumes.
//
// Reference (numpy, one image at a time on the host, after a 2 x 16 MB device-to-host copy): for every instance id in the
// panoptic map a boolean region mask is built and np.unique(seg[region], return_counts=True) votes the region's majority
// semantic class; the region keeps its thing class, or is re-labelled as stuff when a stuff class holds >= 50 % of it;
// void (255) is passed through; finally stuff classes covering less than stuff_area_limit pixels become void.
// Here, all on the device, integers only (bit-exact by construction):
//   unipan_hist_kernel   one pass over (pan, seg): run-length-compressed counts in an LDS-privatised table hist[id][seg class]
//   unipan_decide_kernel one workgroup: majority vote (first maximum, like np.argmax), the three-way decision, rank of each
//                        present id (the reference's enumerate index), stuff-area filter -> a 256-entry LUT id -> (cat, ins)
//   unipan_apply_kernel  one pass: LUT lookup, uint8 [H,W,3] written (3rd channel zero)
#include "common.h"
#include "resize.h"
#include "upsnet_hip.h"

#define UP_IDS 256
#define UP_RUN 16

struct UniPanWs {
    int hist[UP_IDS][UP_IDS];   // [panoptic id][semantic class]
    unsigned char lut_cat[UP_IDS], lut_ins[UP_IDS];
};

// LDS-privatised histogram: a workgroup owns 4096 consecutive pixels (thread -> UP_RUN consecutive ones, run-length compressed),
// counts them in an LDS table [row][class] (row = panoptic id, void in the last row) and flushes the non-zero entries with one
// global atomic each. Works for smooth label maps (one run per thread) and for noisy ones (LDS atomics, no global contention).
__global__ void __launch_bounds__(256)
unipan_hist_kernel(const int64_t *__restrict__ pan, const int64_t *__restrict__ seg, const long npix, const int nrow, const int ncls,
                   UniPanWs *__restrict__ ws)
{
    extern __shared__ int s_hist[];   // [nrow][ncls]
    const int n = nrow * ncls;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const long start = ((long)blockIdx.x * blockDim.x + threadIdx.x) * UP_RUN;
    const long end = start + UP_RUN < npix ? start + UP_RUN : npix;
    int key = -1, run = 0;
#define UP_FLUSH()                                                                                             \
    if (run) {                                                                                                 \
        const int id = key >> 8, cls = key & 255;                                                              \
        const int row = id == 255 ? nrow - 1 : id;                                                             \
        if (row < nrow && cls < ncls && (id == 255 || id < nrow - 1)) atomicAdd(&s_hist[row * ncls + cls], run); \
        else atomicAdd(&ws->hist[id][cls], run);   /* id / class outside the expected table */                \
    }
    for (long i = start; i < end; ++i) {
        const int k = (((int)pan[i] & 255) << 8) | ((int)seg[i] & 255);
        if (k != key) { UP_FLUSH() key = k; run = 0; }
        ++run;
    }
    UP_FLUSH()
#undef UP_FLUSH
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = s_hist[i];
        if (v) {
            const int row = i / ncls, cls = i - row * ncls;
            atomicAdd(&ws->hist[row == nrow - 1 ? 255 : row][cls], v);
        }
    }
}

__global__ void __launch_bounds__(UP_IDS)
unipan_decide_kernel(UniPanWs *__restrict__ ws, const int64_t *__restrict__ cls_inds, const int num_inst, const int id_last_stuff,
                     const int num_seg, const int stuff_area_limit)
{
    __shared__ int s_present[UP_IDS], s_to_stuff[UP_IDS], s_total[UP_IDS], s_stuff_area[UP_IDS];
    const int t = threadIdx.x;
    int total = 0;
    for (int c = 0; c < UP_IDS; ++c) total += ws->hist[t][c];
    const bool is_ins = t > id_last_stuff;            // ids_ins = ids[ids > id_last_stuff]  (:342)
    const bool present = is_ins && total > 0;
    int to_stuff = -1;                                // stuff class this region is re-labelled to, or -1
    int cat = t;                                      // category written for pixels of this id
    if (present && t != 255) {
        int best = 0, best_c = 0;                     // np.unique sorts classes; np.argmax takes the first maximum (:350-351)
        for (int c = 0; c < num_seg; ++c) {
            const int n = ws->hist[t][c];
            if (n > best) { best = n; best_c = c; }
        }
        const int j = t - id_last_stuff - 1;
        const int inst_cat = (j < num_inst ? (int)cls_inds[j] : 0) + id_last_stuff;
        if (best_c == inst_cat) cat = inst_cat;
        else if (2L * best >= (long)total && best_c <= id_last_stuff) { cat = best_c; to_stuff = best_c; }   // max/sum >= 0.5 (:355)
        else cat = inst_cat;
    }
    s_present[t] = present ? 1 : 0;
    s_to_stuff[t] = to_stuff;
    s_total[t] = total;
    __syncthreads();
    // enumerate index of this id among the present instance ids, ascending (:345)
    int rank = 0;
    for (int q = id_last_stuff + 1; q < t; ++q) rank += s_present[q];
    // stuff areas after re-labelling (:362-367): own pixels + regions voted into this class
    int sa = 0;
    if (t <= id_last_stuff) {
        sa = total;
        for (int q = id_last_stuff + 1; q < UP_IDS; ++q) if (s_to_stuff[q] == t) sa += s_total[q];
    }
    s_stuff_area[t] = sa;
    __syncthreads();
    unsigned char ins = 0;
    if (t <= id_last_stuff) {
        if (sa < stuff_area_limit) cat = 255;
    } else if (t == 255) {
        cat = 255;
    } else if (present) {
        if (to_stuff >= 0) { if (s_stuff_area[to_stuff] < stuff_area_limit) cat = 255; }
        else ins = (unsigned char)(rank + 1);
    }
    ws->lut_cat[t] = (unsigned char)cat;
    ws->lut_ins[t] = ins;
}

__global__ void __launch_bounds__(256)
unipan_apply_kernel(const int64_t *__restrict__ pan, const long npix, const UniPanWs *__restrict__ ws, unsigned char *__restrict__ out)
{
    __shared__ unsigned char s_cat[UP_IDS], s_ins[UP_IDS];
    s_cat[threadIdx.x] = ws->lut_cat[threadIdx.x];
    s_ins[threadIdx.x] = ws->lut_ins[threadIdx.x];
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int id = (int)pan[i] & 255;
        out[3 * i] = s_cat[id];
        out[3 * i + 1] = s_ins[id];
        out[3 * i + 2] = 0;
    }
}

extern "C" size_t upsnet_unified_pan_workspace_bytes(void) { return sizeof(UniPanWs); }

extern "C" int upsnet_unified_pan_result(void *stream, const int64_t *pan, const int64_t *seg, const int64_t *cls_inds, int num_inst,
                                         int height, int width, int id_last_stuff, int num_seg_classes, int stuff_area_limit,
                                         void *workspace, unsigned char *pan_2ch)
{
    UPS_REQUIRE(pan && seg && workspace && pan_2ch && (cls_inds || num_inst == 0), "unified_pan_result: null pointer");
    UPS_REQUIRE(height > 0 && width > 0 && id_last_stuff >= 0 && num_seg_classes > id_last_stuff && num_seg_classes <= UP_IDS,
                "unified_pan_result: bad shape / class counts");
    UPS_REQUIRE(num_inst >= 0 && id_last_stuff + num_inst < 255, "unified_pan_result: instance ids collide with the void label (%d instances)", num_inst);
    hipStream_t st = (hipStream_t)stream;
    UniPanWs *ws = (UniPanWs *)workspace;
    const long npix = (long)height * width;
    if (ups_zero_async(ws, sizeof(UniPanWs), st)) return 1;
    const long threads = (npix + UP_RUN - 1) / UP_RUN;
    int nrow = id_last_stuff + 1 + num_inst + 1;   // stuff ids, instance ids, void
    if ((size_t)nrow * num_seg_classes * sizeof(int) > 60 * 1024) nrow = (int)(60 * 1024 / sizeof(int) / num_seg_classes);  // rest: global atomics
    if (nrow < 1) nrow = 1;
    hipLaunchKernelGGL(unipan_hist_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), (size_t)nrow * num_seg_classes * sizeof(int), st,
                       pan, seg, npix, nrow, num_seg_classes, ws);
    UPS_CHECK_LAUNCH("unipan_hist_kernel");
    hipLaunchKernelGGL(unipan_decide_kernel, dim3(1), dim3(UP_IDS), 0, st, ws, cls_inds, num_inst, id_last_stuff, num_seg_classes, stuff_area_limit);
    UPS_CHECK_LAUNCH("unipan_decide_kernel");
    int blocks = (int)((npix + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unipan_apply_kernel, dim3(blocks), dim3(256), 0, st, pan, npix, ws, pan_2ch);
    UPS_CHECK_LAUNCH("unipan_apply_kernel");
    return 0;
}


// ---------------------------------------------------------------------------------------------
// im_post (upsnet/upsnet_end2end_test.py:95-152): per detection, the 28x28 mask probability of its class is zero-padded to
// 30x30, resized (cv2 INTER_LINEAR) to the box expanded by 30/28 and truncated to integers, thresholded at 0.5, pasted into a
// full-image uint8 mask and run-length encoded column-major by pycocotools. The reference does this on the host, one
// detection at a time, over the whole image (2 M pixels per detection at Cityscapes size).
// Here: one workgroup per detection evaluates only the box region and emits the RLE directly as the sorted list of
// column-major pixel indices at which the mask value changes (first entry = start of the first run of ones); the run lengths
// are the differences of that list (host: upsnet_amd/dataset/rle.py, which also does pycocotools' string compression).
// Pass 1 counts the transitions per column, a workgroup prefix sum gives the offsets, pass 2 writes the positions.
#define IMP_T 256
#define IMP_MAXMS 34   // padded mask side (28 + 2 in every config)

struct ImpBox { int bx0, by0, w, h, x_0, x_1, y_0, y_1; };

__device__ static inline int imp_column(const float *__restrict__ s_pad, const int ps, const ImpBox &b, const int x, const int H,
                                        unsigned prev, unsigned *__restrict__ out, const long col_base)
{
    // transitions of column x: rows y_0..y_1-1 from the resized mask, everything else zero; `prev` = value of the pixel before
    // the column's first row in column-major order; returns the count, writes the positions when out != nullptr
    int n = 0;
    unsigned cur = prev;
    if (b.y_0 > 0 && cur) { if (out) out[n] = (unsigned)col_base; ++n; cur = 0; }   // previous column ended set at row H-1, this one starts with zero rows
    for (int y = b.y_0; y < b.y_1; ++y) {
        const unsigned v = pan_resize_at(s_pad, ps, b.w, b.h, x - b.bx0, y - b.by0) > 0.5f ? 1u : 0u;
        if (v != cur) { if (out) out[n] = (unsigned)(col_base + y); ++n; cur = v; }
    }
    if (b.y_1 < H && cur) { if (out) out[n] = (unsigned)(col_base + b.y_1); ++n; }
    return n;
}

__global__ void __launch_bounds__(IMP_T)
im_post_rle_kernel(const float *__restrict__ boxes, const float *__restrict__ mask_prob, const int64_t *__restrict__ cls_inds,
                   const int num_ch, const int ms, const int H, const int W, const int cap, unsigned *__restrict__ trans,
                   int *__restrict__ trans_cnt)
{
    __shared__ float s_pad[IMP_MAXMS * IMP_MAXMS];
    __shared__ int s_scan[IMP_T];
    __shared__ int s_carry;
    const int d = blockIdx.x, tid = threadIdx.x;
    const int ps = ms + 2;
    const int ch = num_ch > 1 ? (int)cls_inds[d] : 0;
    const float *src = mask_prob + ((long)d * num_ch + ch) * ms * ms;
    for (int i = tid; i < ps * ps; i += IMP_T) {
        const int py = i / ps, px = i - py * ps;
        s_pad[i] = (py >= 1 && py <= ms && px >= 1 && px <= ms) ? src[(py - 1) * ms + (px - 1)] : 0.f;
    }
    if (tid == 0) s_carry = 0;
    // expand_boxes (bbox_transform.py:365-381) in float32, then astype(int32) (truncation)
    const float scale = (float)(((double)ms + 2.0) / (double)ms);
    const float bx1 = boxes[d * 4 + 0], by1 = boxes[d * 4 + 1], bx2 = boxes[d * 4 + 2], by2 = boxes[d * 4 + 3];
    float w_half = (bx2 - bx1) * .5f, h_half = (by2 - by1) * .5f;
    const float x_c = (bx2 + bx1) * .5f, y_c = (by2 + by1) * .5f;
    w_half = w_half * scale; h_half = h_half * scale;
    const int rx1 = (int)(x_c - w_half), rx2 = (int)(x_c + w_half), ry1 = (int)(y_c - h_half), ry2 = (int)(y_c + h_half);
    ImpBox b;
    b.w = max(rx2 - rx1 + 1, 1); b.h = max(ry2 - ry1 + 1, 1);
    b.bx0 = rx1; b.by0 = ry1;
    b.x_0 = max(rx1, 0); b.x_1 = min(rx2 + 1, W);
    b.y_0 = max(ry1, 0); b.y_1 = min(ry2 + 1, H);
    __syncthreads();
    unsigned *out = trans + (long)d * cap;
    const bool touch = b.y_1 == H;   // only then the last pixel of a column (row H-1) can be set and precede the next column
    if (b.x_1 <= b.x_0 || b.y_1 <= b.y_0) { if (tid == 0) trans_cnt[d] = 0; return; }
    for (int x0 = b.x_0; x0 < b.x_1; x0 += IMP_T) {
        const int x = x0 + tid;
        const bool act = x < b.x_1;
        unsigned prev = 0;
        if (act && touch && x > b.x_0) prev = pan_resize_at(s_pad, ps, b.w, b.h, x - 1 - b.bx0, H - 1 - b.by0) > 0.5f ? 1u : 0u;
        const int n = act ? imp_column(s_pad, ps, b, x, H, prev, nullptr, (long)x * H) : 0;
        // exclusive prefix sum over the chunk of columns
        s_scan[tid] = n;
        __syncthreads();
        for (int o = 1; o < IMP_T; o <<= 1) {
            const int v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int base = s_carry + s_scan[tid] - n;
        if (act && n && base + n <= cap) imp_column(s_pad, ps, b, x, H, prev, out + base, (long)x * H);
        __syncthreads();
        if (tid == IMP_T - 1) s_carry += s_scan[tid];
        __syncthreads();
    }
    // a run of ones reaching row H-1 of the last box column ends where column x_1 (all zeros) starts
    if (tid == 0) {
        int total = s_carry;
        if (touch && b.x_1 < W) {
            const unsigned last = pan_resize_at(s_pad, ps, b.w, b.h, b.x_1 - 1 - b.bx0, H - 1 - b.by0) > 0.5f ? 1u : 0u;
            if (last) { if (total < cap) out[total] = (unsigned)((long)b.x_1 * H); ++total; }
        }
        trans_cnt[d] = total;   // > cap: overflow, the caller must retry with a larger capacity
    }
}

extern "C" int upsnet_im_post_rle(void *stream, const float *pred_boxes, const float *mask_prob, const int64_t *cls_inds, int num_det,
                                  int num_mask_channels, int mask_size, int im_height, int im_width, int cap, unsigned *transitions,
                                  int *transition_count)
{
    UPS_REQUIRE(pred_boxes && mask_prob && cls_inds && transitions && transition_count, "im_post_rle: null pointer");
    UPS_REQUIRE(num_det >= 0 && num_mask_channels >= 1 && mask_size >= 2 && mask_size + 2 <= IMP_MAXMS, "im_post_rle: bad mask shape");
    UPS_REQUIRE(im_height > 0 && im_width > 0 && (long)im_height * im_width < (1L << 32) && cap > 0, "im_post_rle: bad image size / capacity");
    if (num_det == 0) return 0;
    hipLaunchKernelGGL(im_post_rle_kernel, dim3(num_det), dim3(IMP_T), 0, (hipStream_t)stream, pred_boxes, mask_prob, cls_inds,
                       num_mask_channels, mask_size, im_height, im_width, cap, transitions, transition_count);
    UPS_CHECK_LAUNCH("im_post_rle_kernel");
    return 0;
}
