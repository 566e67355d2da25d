// Measurement utility: the streaming-read ceiling of the device, measured with the same access
// pattern the fused scan kernels use (one 256-thread workgroup per CU, 16-byte non-temporal loads,
// several loads in flight per lane, nothing written but one word per workgroup). bench.py reports
// it next to the 8 TB/s nominal peak as `roofline.peak_measured` ("% of nominal" and
// "% of achievable").
#include "common.h"

namespace ph {

typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void read_reduce_kernel(const v2u64 *__restrict__ p, int64_t nvec,
                                                          unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // 4 loads of 16 bytes in flight per lane (what a Q6 tile keeps in flight is 6)
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        v2u64 a = __builtin_nontemporal_load(p + i);
        v2u64 b = __builtin_nontemporal_load(p + i + stride);
        v2u64 c = __builtin_nontemporal_load(p + i + 2 * stride);
        v2u64 d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
    }
    for (; i < nvec; i += stride) {
        v2u64 a = __builtin_nontemporal_load(p + i);
        acc += a.x ^ a.y;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ unsigned long long ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

}  // namespace ph

extern "C" int ph_dev_read_reduce(ph_ctx *ctx, const void *dev, int64_t bytes, uint64_t *out_words_dev, int32_t grid) {
    PH_REQUIRE(ctx && dev && out_words_dev && bytes >= 16 && ((uintptr_t)dev % 16) == 0 && grid >= 1 && grid <= 65536,
               "ph_dev_read_reduce: bad arguments (16-byte aligned buffer, 1 <= grid <= 65536)");
    ph::read_reduce_kernel<<<grid, 256, 0, ctx->stream>>>((const ph::v2u64 *)dev, bytes / 16, (unsigned long long *)out_words_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}
