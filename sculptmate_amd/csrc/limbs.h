// The limb-tiled form of an fp32 matrix (the operands of gemm_l3p.hip) and the splits that produce it; shared by the kernels that
// WRITE activations as limbs (gemm_l3p.hip epilogues, norms.hip LayerNorm, attention_l3.hip).
//
// Limb-tiled X [R][K] (K % 32 == 0), rows in blocks of 32, k in chunks of 8, NL limbs per element:
//     byte offset of limb l (0 = leading) of X[r][k] = (((r / 32) * (K / 8) + k / 8) * NL + l) * 512 + (r % 32) * 16 + (k % 8) * 2
// Formats:
//   LT_BF16X3  three bf16 limbs, x = x1 + x2 + x3 EXACTLY for every fp32 x (8 + 8 + 8 significant bits, the fp32 exponent range);
//              six limb products per multiply (gemm_l3.hip's arithmetic);
//   LT_F16X2   two fp16 limbs, x ~ h1 + h2: 22 significant bits while |x| >= 2^-3, an absolute error <= 2^-25 below that, and
//              |x| < 65504 (an fp16 limb has 5 exponent bits: weights are stored pre-multiplied by a power of two that puts their
//              largest magnitude in [2^14, 2^15), undone exactly by the GEMM's alpha; activations go in as they are); three limb
//              products per multiply, each exact in fp32 (11 x 11 bits).
#pragma once
#include "common.h"

namespace sculpt {

static constexpr int LT_BF16X3 = 0, LT_F16X2 = 1;
__host__ __device__ __forceinline__ int lt_limbs(int fmt) { return fmt == LT_F16X2 ? 2 : 3; }

typedef __bf16 lt_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 lt_f16x2 __attribute__((ext_vector_type(2)));
typedef float lt_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lt_cvt_pk(float lo, float hi) {
    const lt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, lt_bf16x2));
}

// four consecutive-k fp32 values -> the 8-byte piece of each limb: x = x1 + x2 + x3 exactly (each limb the round-to-nearest bf16 of
// the exact remainder; 24 significant bits minus two 8-bit limbs leave <= 8 bits) -- gemm_l3.hip's l3_split4
__device__ __forceinline__ void lt_split4(const float (&x)[4], uint2 &p1, uint2 &p2, uint2 &p3) {
#pragma clang fp contract(off)
    const unsigned a1 = lt_cvt_pk(x[0], x[1]), b1 = lt_cvt_pk(x[2], x[3]);
    const float r0 = x[0] - __uint_as_float(a1 << 16), r1 = x[1] - __uint_as_float(a1 & 0xffff0000u);   // exact
    const float r2 = x[2] - __uint_as_float(b1 << 16), r3 = x[3] - __uint_as_float(b1 & 0xffff0000u);
    const unsigned a2 = lt_cvt_pk(r0, r1), b2 = lt_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(a2 << 16), s1 = r1 - __uint_as_float(a2 & 0xffff0000u);       // exact, <= 8 bits
    const float s2 = r2 - __uint_as_float(b2 << 16), s3 = r3 - __uint_as_float(b2 & 0xffff0000u);
    p1 = make_uint2(a1, b1);
    p2 = make_uint2(a2, b2);
    p3 = make_uint2(lt_cvt_pk(s0, s1), lt_cvt_pk(s2, s3));
}

// the same for two fp16 limbs: h1 = fp16(x) (round to nearest even; |x| >= 65520 -> inf), h2 = fp16(x - h1) -- the difference is exact
__device__ __forceinline__ void lt_split4_h(const float (&x)[4], uint2 &p1, uint2 &p2) {
#pragma clang fp contract(off)
    const lt_f32x2 v0 = {x[0], x[1]}, v1 = {x[2], x[3]};
    const lt_f16x2 a = __builtin_convertvector(v0, lt_f16x2), b = __builtin_convertvector(v1, lt_f16x2);   // v_cvt_pk_f16_f32
    const lt_f32x2 r0 = {x[0] - (float)a[0], x[1] - (float)a[1]}, r1 = {x[2] - (float)b[0], x[3] - (float)b[1]};
    const lt_f16x2 c = __builtin_convertvector(r0, lt_f16x2), d = __builtin_convertvector(r1, lt_f16x2);
    p1 = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    p2 = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
}

// X[row][col .. col + 3] (col % 4 == 0) of a limb-tiled matrix with k8 = K / 8 chunks per row: split and store the 8-byte pieces
__device__ __forceinline__ void lt_store4(unsigned char *base, int k8, long row, int col, const float (&x)[4], int fmt = LT_BF16X3) {
    if (fmt == LT_F16X2) {   // kernel-uniform
        uint2 p1, p2;
        lt_split4_h(x, p1, p2);
        unsigned char *d = base + (((row >> 5) * k8 + (col >> 3)) * 2) * 512 + (row & 31) * 16 + ((col >> 2) & 1) * 8;
        *reinterpret_cast<uint2 *>(d) = p1;
        *reinterpret_cast<uint2 *>(d + 512) = p2;
        return;
    }
    uint2 p1, p2, p3;
    lt_split4(x, p1, p2, p3);
    unsigned char *d = base + (((row >> 5) * k8 + (col >> 3)) * 3) * 512 + (row & 31) * 16 + ((col >> 2) & 1) * 8;
    *reinterpret_cast<uint2 *>(d) = p1;
    *reinterpret_cast<uint2 *>(d + 512) = p2;
    *reinterpret_cast<uint2 *>(d + 1024) = p3;
}

}  // namespace sculpt
