// Microbenchmark (development aid): out = relu(acc + res) for a [M, 256] fp32 map with the two access patterns an MFMA
// epilogue can use. A: accumulator layout (lane = 1 channel x 16 rows, 4-byte accesses, 128 contiguous bytes per half-wave).
// B: row layout after a transpose (lane = float4 of one row, 1 KiB contiguous per wave-instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) pat_a(const float *__restrict__ acc, const float *__restrict__ res, float *__restrict__ out, long M, int C)
{
    // workgroup = 64 rows x 64 channels (4 waves of 32x32), like the conv tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, lhalf = lane >> 5;
    const int nt = C / 64;
    const long m_t = blockIdx.x / nt; const int n_t = blockIdx.x % nt;
    const long row0 = m_t * 64 + (wave & 1) * 32 + 4 * lhalf;
    const int co = n_t * 64 + (wave >> 1) * 32 + l32;
    float r[16], a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const long row = row0 + (i & 3) + 8 * (i >> 2); r[i] = res[row * C + co]; a[i] = acc[row * C + co]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const long row = row0 + (i & 3) + 8 * (i >> 2); out[row * C + co] = fmaxf(a[i] + r[i], 0.f); }
}
__global__ void __launch_bounds__(256) pat_b(const float4 *__restrict__ acc, const float4 *__restrict__ res, float4 *__restrict__ out, long M, int C)
{
    // same 64 x 64 tile: thread t handles float4 (row t/16 + 16 i, channels 4 (t%16)), i = 0..3
    const int nt = C / 64;
    const long m_t = blockIdx.x / nt; const int n_t = blockIdx.x % nt;
    const int t = threadIdx.x;
    float4 r[4], a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const long idx = ((m_t * 64 + t / 16 + 16 * i) * C + n_t * 64) / 4 + t % 16; r[i] = res[idx]; a[i] = acc[idx]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long idx = ((m_t * 64 + t / 16 + 16 * i) * C + n_t * 64) / 4 + t % 16;
        out[idx] = make_float4(fmaxf(a[i].x + r[i].x, 0.f), fmaxf(a[i].y + r[i].y, 0.f), fmaxf(a[i].z + r[i].z, 0.f), fmaxf(a[i].w + r[i].w, 0.f));
    }
}
int main()
{
    const long M = 131072; const int C = 256; const size_t n = (size_t)M * C;
    float *a, *r, *o;
    hipMalloc(&a, n * 4); hipMalloc(&r, n * 4); hipMalloc(&o, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(r, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)(M / 64) * (C / 64);
    for (int pat = 0; pat < 2; ++pat) {
        for (int it = 0; it < 3; ++it) { if (pat == 0) pat_a<<<grid, 256>>>(a, r, o, M, C); else pat_b<<<grid, 256>>>((float4 *)a, (float4 *)r, (float4 *)o, M, C); }
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) { if (pat == 0) pat_a<<<grid, 256>>>(a, r, o, M, C); else pat_b<<<grid, 256>>>((float4 *)a, (float4 *)r, (float4 *)o, M, C); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pattern %c: %.1f us per pass, %.2f TB/s (3 x %.0f MB)\n", pat ? 'B' : 'A', ms / 20 * 1e3, 3.0 * n * 4 / (ms / 20 * 1e-3) / 1e12, n * 4 / 1e6);
    }
    return 0;
}
