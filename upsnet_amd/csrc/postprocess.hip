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
    UPS_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(UniPanWs), st));
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
