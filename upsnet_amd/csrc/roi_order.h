// roi_order.h -- ROI -> XCD dealing table of the FPN ROIAlign launch (r13), as a workgroup-wide device function: used by fpn_roi_order_kernel
// (roi_align.hip, any ROI set) and at the end of prop_merge_kernel (proposal.hip: the box head's proposals get their table in the launch
// that ranks them -- no extra launch on the image's serial chain).
#pragma once
#include "common.h"

#define ROI_ORDER_MAX 2048
#define ROI_ORDER_NB (4 * 16 * 8 + 1)     // (level, 16 image stripes, 8 column cells) + one bucket for the rows beyond the valid count
#define ROI_ORDER_LDS (ROI_ORDER_NB + 63) // ints of LDS the caller provides

// FPN level of a ROI (upsnet/operators/modules/fpn_roi_align.py:35-51): floor(2 + log2(sqrt(w h) / 224 + 1e-6)) clamped to [0, 3]
__device__ static inline int fpn_level_of(float x1, float y1, float x2, float y2)
{
    float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
    float s = sqrtf(w * h) / 224.0f + 1e-6f;
    float l = floorf(2.0f + ups_log2_f32(s));
    l = fminf(fmaxf(l, 0.f), 3.f);
    return (int)l;
}

// order[b] = the ROI workgroup b of the ROIAlign launch takes (workgroup b runs on XCD b % 8; every XCD has its own L2). A counting sort, not a
// comparison sort (a 1024-key bitonic network in one workgroup takes ~20 us -- more than the dealing saves): bucket = (pyramid level, stripe of
// the ROI centre, column cell; boustrophedon), position = bucket base + arrival slot, and XCD j's workgroups (b = j, j + 8, ...) take one contiguous
// range of positions. The order INSIDE a bucket is the order in which the LDS atomics are served -- it only decides which of two neighbouring
// workgroups takes which of two neighbouring ROIs; the ROIAlign OUTPUT does not depend on the table at all. All 1024 threads of the workgroup
// must call; num_rois <= ROI_ORDER_MAX; rows [nvalid, num_rois) (zero-filled outputs) are ordered last. rois must be visible to the workgroup.
__device__ static inline void ups_roi_order_block(const float *rois, const int num_rois, const int nvalid, const float inv_stripe_h,
                                                  const float inv_cell_w, int *__restrict__ order, int *cnt)
{
    for (int i = threadIdx.x; i < ROI_ORDER_LDS; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    int bkt[2], slot[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = threadIdx.x + q * 1024;
        bkt[q] = -1;
        slot[q] = 0;
        if (i < num_rois) {
            int b = ROI_ORDER_NB - 1;
            if (i < nvalid) {
                const float *r = rois + (long)i * 5;
                const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
                const int lvl = fpn_level_of(x1, y1, x2, y2);
                const int st = min(max((int)((y1 + y2) * 0.5f * inv_stripe_h), 0), 15), cell = min(max((int)((x1 + x2) * 0.5f * inv_cell_w), 0), 7);
                b = (lvl * 16 + st) * 8 + ((st & 1) ? 7 - cell : cell);     // (boustrophedon: consecutive buckets are neighbours in the image)
            }
            bkt[q] = b;
            slot[q] = atomicAdd(&cnt[b], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                  // exclusive scan of the bucket counts by one wave: 9 consecutive buckets per lane
        constexpr int PER = ROI_ORDER_LDS / 64;
        int loc[PER], sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) { loc[u] = cnt[threadIdx.x * PER + u]; sum += loc[u]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += v; }
        int run = incl - sum;
#pragma unroll
        for (int u = 0; u < PER; ++u) { cnt[threadIdx.x * PER + u] = run; run += loc[u]; }
    }
    __syncthreads();
    // XCD j owns workgroups j, j + 8, ...: n_j = ceil((N - j) / 8) of them; it takes sorted positions [start_j, start_j + n_j)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (bkt[q] < 0) continue;
        const int k = cnt[bkt[q]] + slot[q];
        int j = 0, start = 0;
        bool found = false;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int nj = (num_rois - x + 7) >> 3;
            if (!found) { if (k < start + nj) { j = x; found = true; } else start += nj; }
        }
        order[(k - start) * 8 + j] = threadIdx.x + q * 1024;
    }
}
