// sort.h -- workgroup-wide bitonic sort of unique 64-bit keys held in LDS (descending).
// Every selection in the path (top-k, NMS visiting order, final proposal ranking) is expressed as a
// sort of UNIQUE composite keys (orderable score bits << 32 | index or ~index), so no stability
// argument is needed and tie order is exactly the one pinned in the oracle.
#pragma once
#include "common.h"

typedef unsigned long long ups_u64;

// keys[0..M) in LDS, M a power of two >= 2; all threads of the workgroup must call; blockDim.x a multiple of 64.
// A compare-exchange step with partner distance j < 64 pairs elements of one aligned 64-block, and thread t visits the elements
// t, t + blockDim, ...: every 64-block belongs to ONE wavefront per visit, so the whole tail j = 32, 16, ..., 1 of a merge stage
// (and every stage with k <= 64) stays inside a wavefront -- LDS operations of a wave execute in order, no workgroup barrier is
// needed between those steps. Only the steps with j >= 64 cross waves. For 8192 keys that is 41 barriers instead of 91, for 1024
// keys 20 instead of 55 (each ~0.7 us with 16 waves); same network, same result.
__device__ static inline void ups_sort_step(ups_u64 *keys, const int M, const int k, const int j, const int tid, const int bd)
{
    for (int i = tid; i < M; i += bd) {
        const int ixj = i ^ j;
        if (ixj > i) {
            const ups_u64 a = keys[i], b = keys[ixj];
            const bool sw = ((i & k) == 0) ? (a < b) : (a > b);
            if (sw) { keys[i] = b; keys[ixj] = a; }
        }
    }
}

__device__ static inline void ups_block_sort_desc(ups_u64 *keys, const int M)
{
    const int tid = threadIdx.x, bd = blockDim.x;
    __syncthreads();
    for (int k = 2; k <= M; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64; j >>= 1) {
            ups_sort_step(keys, M, k, j, tid, bd);
            __syncthreads();
        }
        for (; j > 0; j >>= 1) {
            ups_sort_step(keys, M, k, j, tid, bd);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (compiler ordering of the LDS accesses; no instruction)
            __builtin_amdgcn_wave_barrier();
        }
        if (k >= 64) __syncthreads();   // the next stage starts with a cross-wave step (j = k >= 64)
    }
    __syncthreads();
}

// Rank ("counting") sort for small problems: keys[0..n) UNIQUE and non-zero, keys[n..M) padding (n <= M, M a power of two, M * 8
// bytes of LDS; at most UPS_RANK_R * blockDim.x >= M elements). Every thread counts, for each of its elements, how many keys
// precede it -- all lanes read the same LDS address (broadcast), two keys per ds_read_b128 -- and scatters the element to that
// rank: two barriers in all instead of the ~log^2 compare-exchange steps of the bitonic network. It pays for SMALL sets only (n^2 / 2
// broadcast LDS reads per wave: ~100 keys of mask removal; at 1000 keys the bitonic network with one key per thread is faster --
// measured 66 vs 16 us). Result identical: descending keys in [0, n), zeros behind.
#define UPS_RANK_R 4
__device__ static inline void ups_block_rank_sort_desc(ups_u64 *keys, const int n, const int M)
{
    const int tid = threadIdx.x, bd = blockDim.x;
    __syncthreads();
    ups_u64 mine[UPS_RANK_R];
    int rank[UPS_RANK_R];
#pragma unroll
    for (int r = 0; r < UPS_RANK_R; ++r) {
        const int i = tid + r * bd;
        mine[r] = i < n ? keys[i] : 0ULL;
        rank[r] = 0;
    }
    const int n2 = (n + 1) & ~1;                       // (keys[n] is padding (0) when n is odd: it precedes nothing)
#pragma unroll 8
    for (int j = 0; j < n2; j += 2) {
        const ups_u64 k0 = keys[j], k1 = (j + 1 < n) ? keys[j + 1] : 0ULL;
#pragma unroll
        for (int r = 0; r < UPS_RANK_R; ++r) rank[r] += (int)(k0 > mine[r]) + (int)(k1 > mine[r]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < UPS_RANK_R; ++r) {
        const int i = tid + r * bd;
        if (i < n) keys[rank[r]] = mine[r];
        else if (i < M) keys[i] = 0ULL;
    }
    __syncthreads();
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
