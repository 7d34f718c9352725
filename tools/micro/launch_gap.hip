// What a dependent launch costs outside its kernel, and whether the store policy of the previous kernel changes it.
//   hipcc -O3 --offload-arch=gfx950 -o launch_gap launch_gap.hip && ./launch_gap
// A chain of kernels on one stream, each reading what the previous one wrote (ping-pong between two buffers): 256 workgroups of
// 512 threads move `bytes` of fp32 (+1.0) per launch.  Modes of the stores: 0 default, 1 nt, 2 sc0 sc1 (write-through to memory),
// 3 sc1.  Per mode and size: us per launch over the chain, and the kernel's own span (s_memrealtime, first entry -> last exit).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void chain_kernel(const float4 *in, float4 *out, long n4, unsigned long long *stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n4; i += (long)gridDim.x * 512) {
        float4 v = in[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        float4 *p = out + i;
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4 vv = {v.x, v.y, v.z, v.w};
        if (MODE == 0) *p = v;
        else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(vv) : "memory");
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(vv) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(vv) : "memory");
    }
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE>
static void run(long bytes, float4 *a, float4 *b, unsigned long long *stamps, int chain) {
    const long n4 = bytes / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < chain; ++k)
            hipLaunchKernelGGL(chain_kernel<MODE>, dim3(256), dim3(512), 0, 0, (k & 1) ? b : a, (k & 1) ? a : b, n4, (unsigned long long *)nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1e3f / chain);
    }
    std::sort(ts.begin(), ts.end());
    hipLaunchKernelGGL(chain_kernel<MODE>, dim3(256), dim3(512), 0, 0, a, b, n4, stamps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < 256; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
    printf("  mode %d  %8.1f KB per launch: %6.2f us per launch in a chain of %d (median of 7), kernel span %6.2f us -> outside %5.2f us\n",
           MODE, bytes / 1024.0, ts[3], chain, (hi - lo) / 100.0, ts[3] - (hi - lo) / 100.0);
}

int main() {
    float4 *a, *b; unsigned long long *stamps;
    const long maxb = 64L << 20;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&stamps, 512 * 8));
    CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
    const long sizes[] = {256L << 10, 2L << 20, 12L << 20, 48L << 20};
    for (long s : sizes) {
        printf("%ld KB read + written per launch\n", s >> 10);
        run<0>(s, a, b, stamps, 200);
        run<1>(s, a, b, stamps, 200);
        run<2>(s, a, b, stamps, 200);
        run<3>(s, a, b, stamps, 200);
    }
    return 0;
}
