// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) operand / result lane layout, (2) issue rate with 20 independent accumulators,
// (3) rate of v_fmac_f32 for comparison. Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/ubench/mfma4x4_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ void rate_kernel(float *out, int iters)
{
    floatx4 acc[NACC];
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void fma_rate_kernel(float *out, int iters)
{
    float acc[40];
    const float a = threadIdx.x * 0.001f;
    float b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 40; ++i) acc[i] = __builtin_fmaf(a, b, acc[i]);
        b += 1e-9f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 40; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f * (1 + l); }   // A lane l = 1 + l, B lane l = 100 (1 + l)
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: block = l / 4; A row i = l % 4; B col j = l % 4; D: lane 4 blk + j, vgpr i  ->  d[(4 blk + j) * 4 + i] = A[4 blk + i] * B[4 blk + j]
    int ok = 1;
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        if (hd[(4 * blk + j) * 4 + i] != ha[4 * blk + i] * hb[4 * blk + j]) ok = 0;
    printf("layout hypothesis (D[lane 4b+j][vgpr i] = A[lane 4b+i] * B[lane 4b+j]): %s\n", ok ? "CONFIRMED" : "WRONG");
    if (!ok) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3]);
    float *o; hipMalloc(&o, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int waves = 1; waves <= 4; ++waves) {           // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate_kernel<20>, dim3(256 * waves), dim3(256), 0, 0, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 512.0 * 20 * iters * 4.0 * 256 * waves;
        printf("4x4x1 MFMA, 20 accumulators, %d wave(s)/SIMD: %.1f TFLOP/s\n", waves, flops / ms / 1e9);
    }
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_rate_kernel, dim3(256 * waves), dim3(256), 0, 0, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 64 * 40 * iters * 4.0 * 256 * waves;
        printf("v_fmac_f32, 40 accumulators, %d wave(s)/SIMD: %.1f TFLOP/s\n", waves, flops / ms / 1e9);
    }
    return 0;
}
