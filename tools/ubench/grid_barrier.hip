// grid_barrier.hip -- what does a grid-wide barrier cost on this chip with one workgroup per CU? (development aid, r13; VERDICT r05 next #3 (ii):
// "one cooperative launch per res-stage (grid barrier between layers) -- measure the barrier on 256 CUs first, keep only if < 4 us".)
// Two software barriers, both with BOUNDED spins (a barrier that cannot complete sets an error flag and every workgroup leaves: a ubench must not be
// able to hang the box):
//   flat: one monotonic counter; arrive = lane-0 release fence + relaxed agent-scope atomic add; wait = relaxed sc1 polls with s_sleep, then an
//         agent-scope acquire fence, __syncthreads().
//   xcd:  hierarchical -- a counter per XCD (workgroup b counts on XCD b % 8's counter), the last arriver of an XCD arrives on the top counter, the
//         last of those publishes a generation word everybody polls.
// Between two barriers every workgroup optionally WRITES `kb` KiB (so the release fence has dirty lines to write back -- a layer's output) and reads
// a neighbour's first word (so the acquire matters). Reported: microseconds per barrier = (kernel time with N barriers - kernel time of the same
// work without barriers) / N.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/grid_barrier tools/ubench/grid_barrier.hip && tools/ubench/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

#define SPIN_MAX 2000000

struct Bar {
    unsigned flat;            // monotonic arrivals
    unsigned pad0[31];
    unsigned xcd[8][32];      // per-XCD monotonic arrivals (one cache line each)
    unsigned top;             // XCD leaders' arrivals
    unsigned pad1[31];
    unsigned gen;             // published generation
    unsigned pad2[31];
    unsigned error;
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: no barrier; 1: flat; 2: xcd-hierarchical
template <int MODE>
__global__ void __launch_bounds__(256, 1) bar_kernel(Bar *bar, float *scratch, int iters, int kb, unsigned base_gen)
{
    const int b = blockIdx.x, nb = gridDim.x;
    float *mine = scratch + (size_t)b * (kb > 0 ? kb * 256 : 256);
    float acc = 0.f;
    __shared__ int s_abort;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (s_abort) break;                  // (workgroup-uniform: written by thread 0 before the barrier below)
        // a phase: write kb KiB of own output, read a word of the neighbour's previous output
        for (int i = threadIdx.x; i < kb * 256; i += 256) mine[i] = (float)(it + i);
        if (threadIdx.x == 0) acc += scratch[(size_t)((b + 1) % nb) * (kb > 0 ? kb * 256 : 256)];
        if (MODE == 0) continue;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned target = base_gen + (unsigned)(it + 1);
            if (MODE == 1) {
                __hip_atomic_fetch_add(&bar->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = target * (unsigned)nb;
                int spin = 0;
                while ((int)(ld_relaxed(&bar->flat) - want) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spin > SPIN_MAX) { bar->error = 1; break; }
                }
            } else {
                const int x = b & 7;
                const unsigned per = (unsigned)((nb - x + 7) >> 3);        // workgroups of this XCD
                const unsigned old = __hip_atomic_fetch_add(&bar->xcd[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == target * per) {                              // last of the XCD
                    const unsigned nx = (unsigned)(nb < 8 ? nb : 8);
                    const unsigned t = __hip_atomic_fetch_add(&bar->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t + 1 == target * nx) __hip_atomic_store(&bar->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int spin = 0;
                while ((int)(ld_relaxed(&bar->gen) - target) < 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spin > SPIN_MAX) { bar->error = 1; break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (ld_relaxed(&bar->error)) s_abort = 1;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) scratch[0] = acc;
}

template <int MODE> static float run(Bar *bar, float *scratch, int grid, int iters, int kb, unsigned &gen)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bar_kernel<MODE><<<grid, 256>>>(bar, scratch, 20, kb, gen);      // warm-up
    if (MODE) gen += 20;
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bar_kernel<MODE><<<grid, 256>>>(bar, scratch, iters, kb, gen);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    if (MODE) gen += (unsigned)iters;
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; one 256-thread workgroup per CU, %d barriers per launch\n", prop.gcnArchName, cus, 200);
    Bar *bar;
    float *scratch;
    CHECK(hipMalloc(&bar, sizeof(Bar)));
    CHECK(hipMalloc(&scratch, (size_t)cus * 64 * 256 * sizeof(float)));
    CHECK(hipMemset(scratch, 0, (size_t)cus * 64 * 256 * sizeof(float)));
    const int iters = 200;
    for (int kb : {0, 4, 32, 64}) {
        // each mode gets fresh counters (flat and xcd count independently)
        CHECK(hipMemset(bar, 0, sizeof(Bar)));
        unsigned g0 = 0, g1 = 0, g2 = 0;
        const float t0 = run<0>(bar, scratch, cus, iters, kb, g0);
        CHECK(hipMemset(bar, 0, sizeof(Bar)));
        const float t1 = run<1>(bar, scratch, cus, iters, kb, g1);
        CHECK(hipMemset(bar, 0, sizeof(Bar)));
        const float t2 = run<2>(bar, scratch, cus, iters, kb, g2);
        Bar h;
        CHECK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
        printf("%2d KiB written per workgroup and phase: no barrier %7.2f us / phase | flat counter +%6.2f us / barrier | XCD-hierarchical +%6.2f us / barrier%s\n",
               kb, t0 / iters, (t1 - t0) / iters, (t2 - t0) / iters, h.error ? "   (a spin hit its bound: numbers invalid)" : "");
        fflush(stdout);
    }
    return 0;
}
