// microbenchmark: cost of s_barrier in an 8-wave workgroup (standalone; not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void bar_kernel(int n, int *out) {
    int acc = 0;
    for (int i = 0; i < n; ++i) {
        __builtin_amdgcn_s_barrier();
        acc += i;
    }
    if (acc == -1) out[0] = acc;
}
__global__ __launch_bounds__(512) void bar_lds_kernel(int n, int *out) {
    __shared__ int big[28672];  // 112 KiB: one workgroup per CU
    int acc = 0;
    for (int i = 0; i < n; ++i) {
        __builtin_amdgcn_s_barrier();
        acc += big[(threadIdx.x + i) & 28671];
    }
    if (acc == -1) out[0] = acc;
}
int main() {
    int *d; hipMalloc(&d, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {256, 384, 512}) for (int variant = 0; variant < 2; ++variant) {
        const int n = 1000;
        for (int w = 0; w < 3; ++w) { if (variant) hipLaunchKernelGGL(bar_lds_kernel, dim3(blocks), dim3(512), 0, 0, n, d); else hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(512), 0, 0, n, d); }
        hipEventRecord(a);
        for (int r = 0; r < 10; ++r) { if (variant) hipLaunchKernelGGL(bar_lds_kernel, dim3(blocks), dim3(512), 0, 0, n, d); else hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(512), 0, 0, n, d); }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("blocks %d variant %d: %.2f us per launch, %.1f ns per barrier-iteration\n", blocks, variant, ms * 100, ms * 1e6 / 10 / n);
    }
    return 0;
}
