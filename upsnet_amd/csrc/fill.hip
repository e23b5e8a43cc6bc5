// fill.hip -- zero-fill of small scratch arrays as a plain kernel (graph-capture friendly: a captured forward holds kernel nodes only).
#include <stdlib.h>

#include "common.h"

__global__ void __launch_bounds__(256) ups_zero_kernel(uint32_t *__restrict__ p, const size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)blockDim.x * gridDim.x) p[i] = 0u;
}

int ups_zero_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return 0;
    static const bool use_memset = getenv("UPSNET_HIP_MEMSET") != nullptr && getenv("UPSNET_HIP_MEMSET")[0] == '1';
    if (use_memset || (bytes & 3) || ((size_t)ptr & 3)) {
        UPS_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, st));
        return 0;
    }
    const size_t n4 = bytes >> 2;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ups_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)ptr, n4);
    UPS_CHECK_LAUNCH("ups_zero_kernel");
    return 0;
}
