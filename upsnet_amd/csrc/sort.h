// sort.h -- workgroup-wide bitonic sort of unique 64-bit keys held in LDS (descending).
// Every selection in the path (top-k, NMS visiting order, final proposal ranking) is expressed as a
// sort of UNIQUE composite keys (orderable score bits << 32 | index or ~index), so no stability
// argument is needed and tie order is exactly the one pinned in the oracle.
#pragma once
#include "common.h"

typedef unsigned long long ups_u64;

// keys[0..M) in LDS, M a power of two >= 2; all threads of the workgroup must call.
__device__ static inline void ups_block_sort_desc(ups_u64 *keys, const int M)
{
    const int tid = threadIdx.x, bd = blockDim.x;
    __syncthreads();
    for (int k = 2; k <= M; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M; i += bd) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const ups_u64 a = keys[i], b = keys[ixj];
                    const bool sw = ((i & k) == 0) ? (a < b) : (a > b);
                    if (sw) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__host__ __device__ static inline int ups_next_pow2(int v) { int m = 2; while (m < v) m <<= 1; return m; }

// key builders: score descending, then index descending (tie 0) or ascending (tie 1)
__device__ static inline ups_u64 ups_make_key(float score, unsigned idx, int tie_lower_first)
{
    return ((ups_u64)ups_float_key(score) << 32) | (tie_lower_first ? ~idx : idx);
}
__device__ static inline unsigned ups_key_index(ups_u64 k, int tie_lower_first)
{
    const unsigned lo = (unsigned)k;
    return tie_lower_first ? ~lo : lo;
}
__device__ static inline float ups_key_score(ups_u64 k) { return ups_key_float((unsigned)(k >> 32)); }
