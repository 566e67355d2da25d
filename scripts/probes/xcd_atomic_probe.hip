// Where do integer global atomics execute, and does XCD-private placement + narrower scope make them L2-local?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/xcd_atomic_probe scripts/probes/xcd_atomic_probe.hip
// Variants over n random keys in [0, card): table[key] += value (u64), timed with HIP events.
//   A: one shared table, agent scope            (today's global path)
//   B: table per XCD (HW_REG_XCC_ID), agent scope
//   C: table per XCD, workgroup scope            (no sc1: does the L2 of the XCD execute it?)
//   D: table per XCD, wavefront scope
// The XCD copies are summed by a second kernel and compared with variant A's table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7; }

template <int MODE>
__global__ __launch_bounds__(256) void add_kernel(const int64_t *__restrict__ keys, const int64_t *__restrict__ vals, int64_t n,
                                                 unsigned long long *table, unsigned *cnt, int64_t card) {
    const int x = MODE == 0 ? 0 : xcc_id();
    unsigned long long *t = table + (int64_t)x * card;
    unsigned *c = cnt + (int64_t)x * card;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t k = keys[i];
        const unsigned long long v = (unsigned long long)vals[i];
        if (MODE <= 1) {
            __hip_atomic_fetch_add(&t[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&c[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {
            __hip_atomic_fetch_add(&t[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&c[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(&t[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&c[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}

__global__ void merge_kernel(const unsigned long long *table, const unsigned *cnt, int64_t card, int copies, unsigned long long *out, unsigned *outc) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < card; k += (int64_t)gridDim.x * 256) {
        unsigned long long s = 0; unsigned c = 0;
        for (int x = 0; x < copies; x++) { s += table[(int64_t)x * card + k]; c += cnt[(int64_t)x * card + k]; }
        out[k] = s; outc[k] = c;
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 32000000, card = argc > 2 ? atoll(argv[2]) : 65536;
    std::vector<int64_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (int64_t)(s % (uint64_t)card); hv[i] = (int64_t)((s >> 20) % 1000000); }
    int64_t *keys, *vals; unsigned long long *table, *out, *ref; unsigned *cnt, *outc, *refc;
    CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&vals, n * 8));
    CK(hipMalloc(&table, card * 8 * 8)); CK(hipMalloc(&cnt, card * 4 * 8));
    CK(hipMalloc(&out, card * 8)); CK(hipMalloc(&outc, card * 4)); CK(hipMalloc(&ref, card * 8)); CK(hipMalloc(&refc, card * 4));
    CK(hipMemcpy(keys, hk.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(vals, hv.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"A shared table, agent scope", "B per-XCD table, agent scope", "C per-XCD table, workgroup scope", "D per-XCD table, wavefront scope"};
    std::vector<unsigned long long> href(card), hout(card); std::vector<unsigned> hrefc(card), houtc(card);
    for (int grid : {2048, 8192}) for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(table, 0, card * 8 * 8)); CK(hipMemset(cnt, 0, card * 4 * 8));
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: add_kernel<0><<<grid, 256>>>(keys, vals, n, table, cnt, card); break;
            case 1: add_kernel<1><<<grid, 256>>>(keys, vals, n, table, cnt, card); break;
            case 2: add_kernel<2><<<grid, 256>>>(keys, vals, n, table, cnt, card); break;
            default: add_kernel<3><<<grid, 256>>>(keys, vals, n, table, cnt, card); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        merge_kernel<<<256, 256>>>(table, cnt, card, mode == 0 ? 1 : 8, out, outc);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), out, card * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(houtc.data(), outc, card * 4, hipMemcpyDeviceToHost));
        if (mode == 0) { href = hout; hrefc = houtc; }
        int64_t bad = 0;
        for (int64_t k = 0; k < card; k++) bad += hout[k] != href[k] || houtc[k] != hrefc[k];
        printf("grid %5d  %-36s %8.3f ms  %7.1f Grows/s  %s\n", grid, names[mode], best, n / best / 1e6, bad ? "MISMATCH" : "sums equal");
    }
    // host check of variant A itself
    std::vector<unsigned long long> want(card, 0);
    for (int64_t i = 0; i < n; i++) want[hk[i]] += (unsigned long long)hv[i];
    int64_t bad = 0; for (int64_t k = 0; k < card; k++) bad += want[k] != href[k];
    printf("variant A against the host: %s\n", bad ? "MISMATCH" : "equal");
    return 0;
}
