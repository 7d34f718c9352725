// microbenchmark: what does one SiLU + conversion to a 16-bit MFMA operand cost per value in pass A of the filtered density grid
// (csrc/density_filter.hip), for the candidate instruction sequences?  Every wave runs ITER rounds of { 2 MFMAs 32x32x16 ; SiLU of 8
// accumulator values } on register data -- the cadence of density_coarse_kernel -- and reports shader cycles per VALUE and SIMD
// (s_memtime; first start to last end over a workgroup's waves, divided by waves per SIMD).
//   V0  fp32:  v_exp_f32, v_add_f32, v_rcp_f32, v_mul_f32 per value + v_cvt_pk_f16_f32 per pair        (shipped first)
//   V1  fp16:  v_cvt_pk_f16_f32 per pair, then v_exp_f16 x2 (low / high half by SDWA), v_pk_add_f16, v_rcp_f16 x2, v_pk_mul_f16
//   V2  fp16, transcendentals on the low half only of two registers + v_pack / v_perm to rebuild the pair (no SDWA)
//   V3  fp32 like V0 with the add and the multiply on register pairs: v_pk_add_f32 / v_pk_mul_f32
// and the bare instruction costs the variants are made of.
//   hipcc -O3 --offload-arch=gfx950 -o silu_cost silu_cost.hip && ./silu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define ITER 2048

template <int V>
__device__ __forceinline__ void silu8(float *x, unsigned *out) {
    if constexpr (V == 0) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_exp2f(-x[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = 1.0f + t[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_rcpf(t[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = x[i] * t[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 v = {t[2 * q], t[2 * q + 1]};
            out[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
        }
    } else if constexpr (V == 3) {
        // fp32 with the add and the multiply as packed pairs (v_pk_add_f32 / v_pk_mul_f32: two values per 4-cycle issue)
        f2 t[4], xx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { t[q][0] = __builtin_amdgcn_exp2f(-x[2 * q]); t[q][1] = __builtin_amdgcn_exp2f(-x[2 * q + 1]); xx[q][0] = x[2 * q]; xx[q][1] = x[2 * q + 1]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = t[q] + 1.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { t[q][0] = __builtin_amdgcn_rcpf(t[q][0]); t[q][1] = __builtin_amdgcn_rcpf(t[q][1]); }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = t[q] * xx[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(t[q], h2));
    } else if constexpr (V == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned y, e, d, r, o;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(y) : "v"(x[2 * q]), "v"(x[2 * q + 1]));
            asm volatile("v_exp_f16_e64 %0, -%1" : "=v"(e) : "v"(y));
            asm volatile("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e) : "v"(y));
            asm volatile("v_pk_add_f16 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(e));
            asm volatile("v_rcp_f16_e32 %0, %1" : "=v"(r) : "v"(d));
            asm volatile("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(d));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(o) : "v"(y), "v"(r));
            out[q] = o;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned ya, yb, ea, eb, d, da, db, ra, rb, r, y, o;
            asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(ya) : "v"(x[2 * q]));
            asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(yb) : "v"(x[2 * q + 1]));
            asm volatile("v_exp_f16_e64 %0, -%1" : "=v"(ea) : "v"(ya));
            asm volatile("v_exp_f16_e64 %0, -%1" : "=v"(eb) : "v"(yb));
            asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(d) : "v"(ea), "v"(eb));
            asm volatile("v_pk_add_f16 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(d));
            asm volatile("v_rcp_f16_e32 %0, %1" : "=v"(ra) : "v"(d));
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(db) : "v"(d));
            asm volatile("v_rcp_f16_e32 %0, %1" : "=v"(rb) : "v"(db));
            asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(r) : "v"(ra), "v"(rb));
            asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(y) : "v"(ya), "v"(yb));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(o) : "v"(y), "v"(r));
            (void)da;
            out[q] = o;
        }
    }
}

template <int V, bool MFMA>
__global__ __launch_bounds__(1024) void k(unsigned long long *cyc, float *sink, float seed) {
    h8 a;
    for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.01f * (float)((threadIdx.x + j) & 7));
    f32x16 c0, c1;
    for (int j = 0; j < 16; ++j) { c0[j] = seed * (j + 1) + 0.001f * threadIdx.x; c1[j] = -seed * (j + 2); }
    __shared__ unsigned long long tstart;
    if (threadIdx.x == 0) tstart = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t00 = tstart;
    for (int i = 0; i < ITER; i += 2) {
        // 8 values of one accumulator half -> operand -> 2 MFMAs, like a k-step of density_coarse_kernel
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float x[8];
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = half ? c1[8 + j] : c0[j];
            silu8<V>(x, o);
            const u4 w = {o[0], o[1], o[2], o[3]};
            const h8 b = __builtin_bit_cast(h8, w);
            if (MFMA) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            } else {
                c0[half] += (float)b[0] + (float)b[3];
                c1[8 + half] += (float)b[5] + (float)b[6];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int j = 0; j < 16; ++j) acc += c0[j] + c1[j];
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t00);
}

template <int V, bool MFMA>
static double run(int waves_per_simd, unsigned long long *d, float *s) {
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<V, MFMA>), dim3(256), dim3(threads), 0, 0, d, s, 0.37f);
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL((k<V, MFMA>), dim3(256), dim3(threads), 0, 0, d, s, 0.37f);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at a fixed 100 MHz; report time per value instead of guessing the shader clock
    return (double)h / ITER / 8.0 / waves_per_simd;
}

// accuracy of the fp16 sequence against the fp32 one (values in the accumulator range of the decoder)
__global__ void acc_kernel(float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[8];
    unsigned a[4], b[4];
    for (int j = 0; j < 8; ++j) x[j] = -20.0f + 40.0f * (float)(i * 8 + j) / (float)(n * 8);
    silu8<0>(x, a);
    silu8<1>(x, b);
    float worst = 0.f;
    for (int q = 0; q < 4; ++q) {
        const h2 u = __builtin_bit_cast(h2, a[q]), v = __builtin_bit_cast(h2, b[q]);
        for (int e = 0; e < 2; ++e) {
            const float d = fabsf((float)u[e] - (float)v[e]) / fmaxf(fabsf((float)u[e]), 1e-2f);
            worst = fmaxf(worst, d);
        }
    }
    out[i] = worst;
}

int main() {
    unsigned long long *d; float *s;
    hipMalloc(&d, 8); hipMalloc(&s, 4);
    printf("s_memtime ticks (10 ns) per SiLU value and SIMD x 1000; w = waves per SIMD\n");
    const char *names[3] = {"V0 fp32 exp/add/rcp/mul + cvt_pk", "V1 fp16 packed, SDWA halves", "V2 fp16, pack instead of SDWA"};
    for (int w : {1, 2, 4}) {
        printf("w=%d with 2 MFMAs per 8 values: V0 %.3f  V1 %.3f  V2 %.3f  V3 %.3f | no MFMA: V0 %.3f  V1 %.3f  V2 %.3f  V3 %.3f\n", w,
               1e3 * run<0, true>(w, d, s), 1e3 * run<1, true>(w, d, s), 1e3 * run<2, true>(w, d, s), 1e3 * run<3, true>(w, d, s),
               1e3 * run<0, false>(w, d, s), 1e3 * run<1, false>(w, d, s), 1e3 * run<2, false>(w, d, s), 1e3 * run<3, false>(w, d, s));
    }
    (void)names;
    float *o; hipMalloc(&o, 4096 * 4);
    hipLaunchKernelGGL(acc_kernel, dim3(16), dim3(256), 0, 0, o, 4096);
    float h[4096]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    float worst = 0.f; for (float v : h) worst = worst > v ? worst : v;
    printf("fp16 sequence vs fp32 sequence rounded to fp16, x in [-20, 20]: max |diff| / max(|silu|, 1e-2) = %.3e (fp16 ulp = 9.8e-4)\n", worst);
    return 0;
}
