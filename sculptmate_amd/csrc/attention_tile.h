// Shared by the two attention translation units (attention.hip: the phase-separated tile loop; attention_pipe.hip: the
// software-pipelined one, compiled with -fno-slp-vectorize -- sculptmate_amd/build.py PER_FILE_FLAGS).
#pragma once
#include "common.h"

namespace sculpt {

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *alds_ptr_t;
typedef const __attribute__((address_space(1))) void *agbl_ptr_t;

__device__ __forceinline__ int a_lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;  // v_max3_f32 without the canonicalising v_max(x, x) the compiler adds around fmaxf of MFMA results
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// PRE: Q already carries softmax_scale * log2(e) (folded into the projection that produced it, before its bf16 rounding), so
// the MFMA result is the exponent of 2 directly, and the running maximum M is subtracted INSIDE the matrix product: a fifth
// k-step multiplies a constant [1, 1, 0, ...] row fragment with (-M_hi, -M_lo, 0, ...) on the query's lane, M = M_hi + M_lo
// split into two bf16 values (16 significant bits; M itself is kept quantised to that sum, so the rescale factors stay
// exact).  That removes the `s * scale - m * scale` v_pk_fma of every score -- the loop is VALU-bound (640 VALU vs 512 MFMA
// cycles per 64-key tile and wave), FMA-class VALU does not overlap the MFMAs of a co-resident wave -- for 2 more of the 16
// MFMAs per tile.  M is allowed to lag the true maximum by up to PRE_THR (p <= 2^PRE_THR: bf16 keeps its relative precision,
// the sums are fp32), so the rescale branch is taken on the first tile and then only on a jump of more than 2^PRE_THR.
static constexpr float PRE_THR = 10.0f;

// A batch of independent attentions in one launch (TSR.forward on B images; blockIdx.y = batch * heads + head): element
// strides from one batch entry to the next -- Q / K / O rows further down, V^T columns further right.
struct AttnBatch {
    int heads;
    int vt_cols;  // V^T columns readable from an entry's first column (ldvt for one entry; ldvt - (batch-1)*vt_bs side by side)
    long q_bs, k_bs, vt_bs, o_bs;
};

// The fused attentions of the tolerance modes (attention_l3.hip, attention_l2.hip).  blockIdx.z = batch entry (`batch` independent
// attentions of one shape in one launch): element strides of Q, K, Vt (a column offset when the V^T of the entries sit side by
// side) and O; for a limb output the entry's first ROW and the limb format (limbs.h)
struct AttnL3Batch { long q_bs, k_bs, vt_bs, o_bs; int o_row_bs; int o_fmt; };
// attention_l2.hip: the pipelined 8-wave kernel with two fp16 limbs per operand
void attention_l2_pipe_launch(dim3 grid, hipStream_t st, const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt,
                              float *O, int ldo, int Tq, int Tk, float scale_log2e, unsigned char *O_lt, int o_row0, int o_k8,
                              AttnL3Batch ab);

// attention_pipe.hip: the pipelined loop for pre-scaled queries, 128 (nqb 4) or 192 (nqb 6) queries per workgroup
void attention_pipe_launch(int nqb, dim3 grid, dim3 block, hipStream_t st, const uint16_t *Q, int ldq, const uint16_t *K, int ldk,
                           const uint16_t *Vt, int ldvt, uint16_t *O, int ldo, int Tq, int Tk, AttnBatch ab);

}  // namespace sculpt
