// mfma16_valu.hip -- v_mfma_f32_16x16x4_f32 on gfx950 (development aid, r13; the question VERDICT r05 next #2 asks before any small-tile
// kernel is written): (1) operand / result lane layout, (2) issue rate with 1..8 independent accumulators at one and at two waves per SIMD
// -- the `4x4x1` probe (mfma4x4_probe.hip) issued at 0.71-0.76 of the fp32 peak, which would rule the small-tile family out --, (3) the
// dependent-accumulator latency, (4) does VALU overlap it? (the mfma_valu.hip method: NV VALU instructions of one kind next to a chain of
// MFMAs; for `32x32x2` the times ADD: 2.8-5 cycles per VALU instruction per wave on top of the MFMA time).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma16_valu tools/ubench/mfma16_valu.hip && tools/ubench/mfma16_valu
// Units: cycles per SIMD at the reported clock. One `16x16x4` = 2048 flop = 32 cycles at 64 flop / clk / SIMD; one `32x32x2` = 64 cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// ---- (1) layout: D[16 x 16] = A[16 x 4] B[4 x 16]; which lane holds A[i][k], B[k][j], D[i][j]?
__global__ void layout_kernel(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// ---- (2, 3) rate: NACC independent accumulators, WAVES waves per workgroup (one workgroup per CU)
template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) rate16_kernel(float *out, int iters)
{
    floatx4 acc[NACC];
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) rate32_kernel(float *out, int iters)
{
    floatx16 acc[NACC];
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// ---- (4) VALU next to a block of 8 MFMAs on 4 accumulators (2 rounds: each accumulator is re-used after 4 x 32 = 128 cycles > the 40-cycle latency)
// KIND 0: v_fma_f32, 1: v_add_u32, 2: v_pk_fma_f32, 3: v_mov_b32, 4: ds_write_b32, 5: s_add_u32, 6: ds_read_b128
template <int KIND> __device__ __forceinline__ void valu(float &x, floatx4 &x4, unsigned &k, float a, float b)
{
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(k) : "v"(a));
    if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<float __attribute__((ext_vector_type(2))) *>(&x4)) : "v"(*reinterpret_cast<float __attribute__((ext_vector_type(2))) *>(&x4)));
    if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
    if (KIND == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(k), "v"(x) : "memory");
    if (KIND == 5) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
    if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(x4) : "v"(k) : "memory");
}

template <int KIND, int NV, int PLACE, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) probe16(float *sink, int iters, float a, float b)
{
    __shared__ float lds_[16384];
    if (iters < 0) sink[0] = lds_[threadIdx.x];
    floatx4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float x[8];
    floatx4 x4[8];
    unsigned k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = a * (float)i; x4[i] = floatx4{a, b, a, b}; k[i] = (unsigned)(16 * (threadIdx.x & 63) + 2048 * i); }
    const float fa = a + (float)threadIdx.x, fb = b;
    for (int it = 0; it < iters; ++it) {
        if (PLACE == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) valu<KIND>(x[v & 7], x4[v & 7], k[v & 7], a, b);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[m & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[m & 3], 0, 0, 0);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int v = m * (NV / 8); v < (m + 1) * (NV / 8); ++v) valu<KIND>(x[v & 7], x4[v & 7], k[v & 7], a, b);
            }
        }
        if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + x4[i].x + x4[i].y + x4[i].z + x4[i].w + (float)k[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

static float clock_ghz;
static int cus;

template <typename F> static double cycles_per_iter(F launch, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(200);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e-3 * clock_ghz * 1e9 / iters;
}

template <int NACC, int THREADS> static void rate_row(float *sink)
{
    const int iters = 40000;
    const int wps = THREADS / 256;                                            // waves per SIMD
    const double c16 = cycles_per_iter([&](int it) { rate16_kernel<NACC, THREADS><<<cus, THREADS>>>(sink, it); }, iters) / (NACC * wps);
    const double c32 = cycles_per_iter([&](int it) { rate32_kernel<(NACC > 4 ? 4 : NACC), THREADS><<<cus, THREADS>>>(sink, it); }, iters) / ((NACC > 4 ? 4 : NACC) * wps);
    printf("%d wave(s) / SIMD, %d independent accumulators: 16x16x4 %6.1f cycles / MFMA / SIMD (%.3f of the 32-cycle rate)   32x32x2 (%d acc) %6.1f (%.3f of 64)\n",
           wps, NACC, c16, 32.0 / c16, NACC > 4 ? 4 : NACC, c32, 64.0 / c32);
    fflush(stdout);
}

template <int KIND, int THREADS> static void valu_row(const char *name, float *sink)
{
    const int iters = 20000;
#define R(NV, PL) cycles_per_iter([&](int it) { probe16<KIND, NV, PL, THREADS><<<cus, THREADS>>>(sink, it, 1.0f, 0.5f); }, iters)
    printf("%d w/SIMD  8 x 16x16x4 + %-13s in front: NV=0 %6.0f  8 %6.0f  16 %6.0f  32 %6.0f  64 %6.0f | in the gaps: 8 %6.0f  16 %6.0f  32 %6.0f  64 %6.0f   cycles / iteration / SIMD (MFMA alone: %d)\n",
           THREADS / 256, name, R(0, 0), R(8, 0), R(16, 0), R(32, 0), R(64, 0), R(8, 1), R(16, 1), R(32, 1), R(64, 1), 8 * 32 * (THREADS / 256));
#undef R
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    clock_ghz = prop.clockRate * 1e-6f;
    cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %.2f GHz (cycles below assume this clock)\n", prop.gcnArchName, cus, clock_ghz);
    float *sink, *da, *db, *dd;
    CHECK(hipMalloc(&sink, 8192));
    // (1) layout
    float ha[64], hb[64], hd[256];
    CHECK(hipMalloc(&da, 256)); CHECK(hipMalloc(&db, 256)); CHECK(hipMalloc(&dd, 1024));
    // A[i][k] = 1 only at (i0, k0), B = all ones in row k: D[i0][*] = B[k0][*] ...: find the lane of A[i][k] by probing one lane at a time
    int a_row[64], a_k[64], b_col[64], b_k[64];
    for (int probe = 0; probe < 64; ++probe) {
        for (int l = 0; l < 64; ++l) { ha[l] = l == probe ? 1.f : 0.f; hb[l] = 1.f + (float)(l / 16); }   // B lane l: assume k = l / 16 -> value 1 + k
        CHECK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
        layout_kernel<<<1, 64>>>(da, db, dd);
        CHECK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        // the nonzero outputs form one row i of D (16 values, all equal 1 + k)
        a_row[probe] = -1; a_k[probe] = -1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != 0.f) { a_row[probe] = 4 * (l / 16) + r; a_k[probe] = (int)hd[l * 4 + r] - 1; }
    }
    for (int probe = 0; probe < 64; ++probe) {
        for (int l = 0; l < 64; ++l) { hb[l] = l == probe ? 1.f : 0.f; ha[l] = 1.f + (float)(l / 16); }
        CHECK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
        layout_kernel<<<1, 64>>>(da, db, dd);
        CHECK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        b_col[probe] = -1; b_k[probe] = -1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != 0.f) { b_col[probe] = l % 16; b_k[probe] = (int)hd[l * 4 + r] - 1; }
    }
    int ok = 1;
    for (int l = 0; l < 64; ++l) ok &= (a_row[l] == l % 16 && a_k[l] == l / 16 && b_col[l] == l % 16 && b_k[l] == l / 16);
    printf("layout (assuming D[i][j] in lane 16 (i / 4) + j, register i %% 4): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j: %s\n", ok ? "confirmed" : "NOT confirmed");
    if (!ok) for (int l = 0; l < 64; l += 5) printf("  lane %d: A row %d k %d | B col %d k %d\n", l, a_row[l], a_k[l], b_col[l], b_k[l]);
    // (2, 3) issue rate / dependent latency
    rate_row<1, 256>(sink); rate_row<2, 256>(sink); rate_row<4, 256>(sink); rate_row<8, 256>(sink);
    rate_row<1, 512>(sink); rate_row<2, 512>(sink); rate_row<4, 512>(sink); rate_row<8, 512>(sink);
    // (4) VALU / LDS next to the MFMAs
    valu_row<0, 256>("v_fma_f32", sink); valu_row<1, 256>("v_add_u32", sink); valu_row<2, 256>("v_pk_fma_f32", sink); valu_row<3, 256>("v_mov_b32", sink);
    valu_row<4, 256>("ds_write_b32", sink); valu_row<6, 256>("ds_read_b128", sink); valu_row<5, 256>("s_add_u32", sink);
    valu_row<0, 512>("v_fma_f32", sink); valu_row<1, 512>("v_add_u32", sink); valu_row<4, 512>("ds_write_b32", sink); valu_row<6, 512>("ds_read_b128", sink);
    return 0;
}
