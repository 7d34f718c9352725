// microbenchmark: do the MATRIX phase of one wave and the VECTOR phase of another wave on the same SIMD overlap?
// Every wave runs ITER rounds of { NM back-to-back MFMAs ; NV vector instructions (3/4 plain, 1/4 transcendental) }, i.e. the
// phase-separated shape of density_grid_l3_kernel; 1, 2, 3 or 4 waves per SIMD; waves start with different phase offsets
// (wave w skips the first w/W-th of the vector phase) or all together.  Reported: shader cycles per round and SIMD
// divided by the waves per SIMD -- perfect overlap gives max(32 NM, 8 NM + issue(NV)), none gives the sum.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_phase mfma_phase.hip && ./mfma_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITER 512

template <int NM, int NV, int PRIO, bool INTERLEAVE>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink, int stagger) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 7); b[j] = (__bf16)(0.25f * j); }
    f32x16 c0 = {}, c1 = {};
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = 1.0f + 0.001f * (threadIdx.x + j);
    const float z = 1.0001f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ unsigned long long tstart;
    if (threadIdx.x == 0) tstart = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stagger) {  // waves of one SIMD (w, w+4, w+8, w+12) start a quarter round apart
        for (int i = 0; i < (wave >> 2) * (NM * 8 + NV) / 4; ++i) asm volatile("s_nop 7");
    }
#define VPLAIN(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[j]) : "v"(z));
#define VTRANS(j) asm volatile("v_exp_f32 %0, %0" : "+v"(r[j]));
    for (int it = 0; it < ITER; ++it) {
        if (!INTERLEAVE) {
            if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int m = 0; m < NM; m += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int v = 0; v < NV; v += 8) {
                VPLAIN(0) VPLAIN(1) VPLAIN(2) VTRANS(3) VPLAIN(4) VPLAIN(5) VPLAIN(6) VTRANS(7)
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {  // the same work with the vector instructions spread behind the MFMAs
            constexpr int PER = NV / NM;  // vector instructions per MFMA (rounded down; remainder after the loop)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < PER; ++v) {
                    if ((v & 3) == 3) { VTRANS(3) } else { VPLAIN((v & 7) == 3 ? 4 : (v & 7)) }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = NM * PER; v < NV; ++v) VPLAIN(v & 7)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = c0[0] + c1[1];
    for (int j = 0; j < 8; ++j) acc += r[j];
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - tstart);
}

template <int NM, int NV, int PRIO, bool IL>
static double run(int w, int stagger, unsigned long long *d, float *s) {
    hipLaunchKernelGGL((k<NM, NV, PRIO, IL>), dim3(256), dim3(256 * w), 0, 0, d, s, stagger);
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((k<NM, NV, PRIO, IL>), dim3(256), dim3(256 * w), 0, 0, d, s, stagger);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    return (double)h / ITER / w;
}

int main() {
    unsigned long long *d; float *s;
    (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4);
    printf("cycles per round {48 MFMA 32x32x16 bf16 ; 304 vector (228 v_add_f32 + 76 v_exp_f32)} per SIMD and wave-round\n");
    printf("bounds: MFMA pipe 1536; issue = 48*8 + 228*4 + 76*8 = 1904; no overlap = 1536 + 1520 = 3056\n");
    for (int w : {1, 2, 3, 4}) {
        printf("waves/SIMD %d: phases, together %7.1f | phases, staggered %7.1f | phases + setprio(3) in the matrix phase %7.1f (staggered %7.1f) | interleaved %7.1f\n", w,
               run<48, 304, 0, false>(w, 0, d, s), run<48, 304, 0, false>(w, 1, d, s), run<48, 304, 1, false>(w, 0, d, s),
               run<48, 304, 1, false>(w, 1, d, s), run<48, 304, 0, true>(w, 0, d, s));
    }
    return 0;
}
