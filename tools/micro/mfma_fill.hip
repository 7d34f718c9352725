// microbenchmark: what does a vector instruction cost when it is issued BY THE SAME WAVE between two bf16 MFMAs?
// A wave runs ITER iterations of { one v_mfma_f32_32x32x16_bf16 ; N fillers of one kind on 8 independent registers } and
// reports shader cycles per iteration and SIMD (s_memtime, clock-independent; first start to last end over the waves of a
// workgroup, divided by the waves per SIMD).  1, 2 or 4 waves per SIMD run the same stream.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_fill mfma_fill.hip && ./mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITER 4096

#define OPS(X)                                                                                          \
    X(0, "none", "")                                                                                    \
    X(1, "v_add_f32", "v_add_f32 %0, %0, %1")                                                           \
    X(2, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                           \
    X(3, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1")                                                       \
    X(4, "v_and_b32", "v_and_b32 %0, %0, %1")                                                           \
    X(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0")                                                    \
    X(6, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %0, %1")                                           \
    X(7, "v_exp_f32", "v_exp_f32 %0, %0")                                                               \
    X(8, "v_rcp_f32", "v_rcp_f32 %0, %0")                                                               \
    X(9, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1")                                                     \
    X(10, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1")                                                      \
    X(11, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0")                                                \
    X(12, "v_max_f32", "v_max_f32 %0, %0, %1")                                                          \
    X(13, "v_add_u32", "v_add_u32 %0, %0, %1")                                                          \
    X(14, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1")                                                      \
    X(15, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1")                                                    \
    X(16, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1")                                                    \
    X(17, "v_add_f16", "v_add_f16 %0, %0, %1")                                                          \
    X(18, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %1")                                                    \
    X(19, "v_dot2c_f32_bf16", "v_dot2c_f32_bf16 %0, %1, %1")                                            \
    X(20, "v_sub_f32", "v_sub_f32 %0, %0, %1")                                                          \
    X(21, "v_mov_b32", "v_mov_b32 %0, %1")                                                              \
    X(22, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0")                                                      \
    X(23, "v_med3_f32", "v_med3_f32 %0, %0, %1, %1")                                                    \
    X(24, "v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %1")                                            \
    X(25, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1")                                              \
    X(26, "v_fmac_f32", "v_fmac_f32 %0, %1, %1")

template <int OP, int N, bool MFMA>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x + j) & 7); b[j] = (__bf16)(0.25f * j); }
    f32x16 c0 = {}, c1 = {};
    // pk ops take register PAIRS: keep 8 independent 64-bit values
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[8];
    for (int j = 0; j < 8; ++j) r[j] = f2{1.0f + 0.001f * (threadIdx.x + j), 0.5f};
    f2 z = {1.0001f, 1.0f};
    __shared__ unsigned long long tstart;
    if (threadIdx.x == 0) tstart = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t00 = tstart;
    for (int i = 0; i < ITER; i += 2) {
#define FILL(j) if (N > j) {                                                                                       \
        OPS(SEL)                                                                                                   \
    }
#define SEL(id, name, text) if constexpr (OP == id && id != 0) { if (id == 15 || id == 16) asm volatile(text : "+v"(r[jj]) : "v"(z)); else asm volatile(text : "+v"(r[jj].x) : "v"(z.x)); }
        if (MFMA) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        { constexpr int jj = 0; FILL(0) } { constexpr int jj = 1; FILL(1) } { constexpr int jj = 2; FILL(2) } { constexpr int jj = 3; FILL(3) }
        { constexpr int jj = 4; FILL(4) } { constexpr int jj = 5; FILL(5) } { constexpr int jj = 6; FILL(6) } { constexpr int jj = 7; FILL(7) }
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        { constexpr int jj = 0; FILL(0) } { constexpr int jj = 1; FILL(1) } { constexpr int jj = 2; FILL(2) } { constexpr int jj = 3; FILL(3) }
        { constexpr int jj = 4; FILL(4) } { constexpr int jj = 5; FILL(5) } { constexpr int jj = 6; FILL(6) } { constexpr int jj = 7; FILL(7) }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = c0[0] + c1[1];
    for (int j = 0; j < 8; ++j) acc += r[j].x + r[j].y;
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t00);  // first start to last end over the block's waves
}

template <int OP, int N, bool MFMA>
static double run(int waves_per_simd, unsigned long long *d, float *s) {
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<OP, N, MFMA>), dim3(256), dim3(threads), 0, 0, d, s);
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL((k<OP, N, MFMA>), dim3(256), dim3(threads), 0, 0, d, s);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    return (double)h / ITER / waves_per_simd;  // cycles per (MFMA + N fillers) on one SIMD
}

template <int OP>
static void row(const char *name, unsigned long long *d, float *s) {
    printf("%-20s", name);
    for (int w : {1, 2, 4}) {
        printf(" | w=%d:", w);
        printf(" %5.1f", run<OP, 2, true>(w, d, s));
        printf(" %5.1f", run<OP, 4, true>(w, d, s));
        printf(" %5.1f", run<OP, 6, true>(w, d, s));
        printf(" %5.1f", run<OP, 8, true>(w, d, s));
        printf(" (no mfma, 8: %5.1f)", run<OP, 8, false>(w, d, s));
    }
    printf("\n");
}

int main() {
    unsigned long long *d; float *s;
    hipMalloc(&d, 8); hipMalloc(&s, 4);
    printf("cycles per iteration { 1 MFMA 32x32x16 bf16 + N fillers }, N = 2 4 6 8, and 8 fillers without the MFMA; per wave\n");
    printf("%-20s | w=1: %5.1f | w=2: %5.1f | w=4: %5.1f\n", "mfma only", run<0, 0, true>(1, d, s), run<0, 0, true>(2, d, s), run<0, 0, true>(4, d, s));
#define ROW(id, name, text) if (id) row<id>(name, d, s);
    OPS(ROW)
    return 0;
}
