// microbenchmark: do bf16 MFMAs of one wave overlap VALU work of the partner wave on the same SIMD?
// 8-wave workgroups, one per CU; waves 0-3 run `nm` MFMAs, waves 4-7 run `nv` VALU ops (wave w and w+4 share SIMD w%4).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void k(int nm, int nv, int trans, float *out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r = 0.f;
    if (wave < 4) {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)j; }
        f32x16 c0 = {}, c1 = {};
        for (int i = 0; i < nm; i += 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        r = c0[0] + c1[3];
    } else {
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        if (trans == 2) {  // packed fp32 fma
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {x0, x1}, b = {x2, x3}, c = {x0 + 4.f, x1 + 4.f}, d = {x2 + 4.f, x3 + 4.f};
            const f2 m = {1.0001f, 1.0001f}, ad = {0.5f, 0.5f};
            for (int i = 0; i < nv; i += 4) { a = a * m + ad; b = b * m + ad; c = c * m + ad; d = d * m + ad; }
            x0 = a[0] + a[1]; x1 = b[0] + b[1]; x2 = c[0] + c[1]; x3 = d[0] + d[1];
        } else if (trans == 3) {  // v_max3_f32
            for (int i = 0; i < nv; i += 4) {
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(x2), "v"(x3));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(x3), "v"(x0));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x0), "v"(x1));
            }
        } else if (trans == 4) {  // v_cvt_pk_bf16_f32
            unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
            for (int i = 0; i < nv; i += 4) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(x1), "v"(x2));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u2) : "v"(x2), "v"(x3));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u3) : "v"(x3), "v"(x0));
            }
            x0 += __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
        } else if (trans == 5) {  // integer add (address-like VALU)
            int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3;
            for (int i = 0; i < nv; i += 4) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i2));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i3));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(i0));
            }
            x0 = (float)(i0 + i1 + i2 + i3);
        } else if (trans) {
            for (int i = 0; i < nv; i += 4) {
                x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1);
                x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
            }
        } else {
            for (int i = 0; i < nv; i += 4) {
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f);
                x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
            }
        }
        r = x0 + x1 + x2 + x3;
    }
    if (r == 12345.678f) out[0] = r;
}
static float run(int nm, int nv, int trans, float *d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nm, nv, trans, d);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nm, nv, trans, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 200.f;  // us per launch
}
int main() {
    float *d; hipMalloc(&d, 4);
    const int nm = 20000, nv = 160000;   // 20000 MFMAs x 32 cyc = 640k cycles; 160000 VALU x 4 cyc = 640k cycles
    printf("MFMA only           : %.1f us\n", run(nm, 0, 0, d));
    printf("VALU(fma) only      : %.1f us\n", run(0, nv, 0, d));
    printf("MFMA + VALU(fma)    : %.1f us\n", run(nm, nv, 0, d));
    printf("VALU(exp) only      : %.1f us\n", run(0, nv / 4, 1, d));
    printf("MFMA + VALU(exp)    : %.1f us\n", run(nm, nv / 4, 1, d));
    const char *names[] = {"", "", "pk_fma", "max3", "cvt_pk_bf16", "add_u32"};
    for (int kind = 2; kind <= 5; ++kind) {
        printf("VALU(%s) only : %.1f us\n", names[kind], run(0, nv, kind, d));
        printf("MFMA + VALU(%s): %.1f us\n", names[kind], run(nm, nv, kind, d));
    }
    return 0;
}
