// fill.hip -- zero-fill of small scratch arrays as a plain kernel (graph-capture friendly: a captured forward holds kernel nodes only).
#include <stdlib.h>

#include "common.h"

// [p, p + head) and [p + head + 4 n4, p + bytes) are the unaligned byte head / tail (at most 3 bytes each), the middle is dwords
__global__ void __launch_bounds__(256) ups_zero_kernel(unsigned char *__restrict__ p, const size_t head, const size_t n4, const size_t bytes)
{
    uint32_t *__restrict__ q = (uint32_t *)(p + head);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)blockDim.x * gridDim.x) q[i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const size_t t = threadIdx.x;
        if (t < 4) { if (t < head) p[t] = 0; }
        else { const size_t at = head + 4 * n4 + (t - 4); if (at < bytes) p[at] = 0; }
    }
}

int ups_zero_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return 0;
    static const bool use_memset = getenv("UPSNET_HIP_MEMSET") != nullptr && getenv("UPSNET_HIP_MEMSET")[0] == '1';
    if (use_memset) {   // (diagnostic only: memset NODES in a linearly captured graph fault at replay on this stack)
        UPS_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, st));
        return 0;
    }
    // any address / size: unaligned head and tail bytes are written by the same kernel (never a memset node)
    size_t head = (4 - ((size_t)ptr & 3)) & 3;
    if (head > bytes) head = bytes;
    const size_t n4 = (bytes - head) >> 2;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(ups_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char *)ptr, head, n4, bytes);
    UPS_CHECK_LAUNCH("ups_zero_kernel");
    return 0;
}

extern "C" int upsnet_zero_fill(void *stream, void *ptr, size_t bytes)
{
    UPS_REQUIRE(ptr != nullptr || bytes == 0, "upsnet_zero_fill: null pointer");
    return ups_zero_async(ptr, bytes, (hipStream_t)stream);
}
