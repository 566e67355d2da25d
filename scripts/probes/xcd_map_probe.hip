// Which XCD does workgroup b of a 1-D grid land on? hipcc --offload-arch=gfx950 -O3 -o scripts/probes/xcd_map_probe scripts/probes/xcd_map_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(int *out) {
    __shared__ int pad[12288];
    if (threadIdx.x == 0) { pad[0] = 1; out[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15; }
}
int main() {
    const int n = 4096;
    int *d; hipMalloc(&d, n * 4);
    k<<<n, 1024>>>(d);
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int agree = 0; for (int b = 0; b < n; b++) agree += (h[b] == h[b & 7]);
    printf("first 32:"); for (int b = 0; b < 32; b++) printf(" %d", h[b]); printf("\nblocks whose XCD equals that of block (b %% 8): %d of %d\n", agree, n);
    return 0;
}
