// Shared by the two fp32-in / fp32-out GEMM kernels of the parity modes: gemm_f32.hip (exact fp32 matrix pipe,
// v_mfma_f32_32x32x2_f32) and gemm_l3.hip (fp32 arithmetic on the bf16 matrix pipe through an exact three-limb split).
// Both leave a 128 x 128 tile as 2 x 2 waves of 2 x 2 accumulator tiles of 32 x 32 in the same C layout, so the epilogue is one.
#pragma once
#include "common.h"

namespace sculpt {

static constexpr int FBM = 128, FBW = 128;

struct GemmF32Args {
    const float *A; int lda;
    const float *W; int ldw;
    const float *bias;
    const float *residual; int ldr;
    float *out; int ldo;
    float *out_t; int ldt;
    int M, N, K;
    int n_split;
    int w_rows;   // valid rows of W (rows beyond are clamped; lets N be padded to a multiple of 4)
    float alpha;  // scale applied to the accumulator before bias (attention scores)
    // (bias is never null inside the kernels: the launcher substitutes the zero page)
    // blockIdx.z = batch entry (the heads of an attention as ONE launch): element strides of A, W, out / out_t, residual
    long a_bs, w_bs, o_bs;
};

__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GEGLU: tile rows in groups of 32 alternate value / gate so a lane holds matching pairs
template <int EPI>
__device__ __forceinline__ int f32_tile_wrow(const GemmF32Args &g, int n0, int j) {
    if (EPI == SCULPT_EPI_GEGLU) {
        const int sub = j >> 5, within = j & 31;
        return ((sub & 1) ? g.N : 0) + n0 + (sub >> 1) * 32 + within;
    }
    return min(n0 + j, g.w_rows - 1);
}

// acc[i][j][r]: column m = m0 + wc*64 + j*32 + l31; tile row = wr*64 + i*32 + (r&3) + 8*(r>>2) + 4*lh
// (JT = 32-row activation sub-tiles per wave: 2 for the 128-row tile, 1 for gemm_l3's 64-row tile)
template <int EPI, int JT = 2>
__device__ __forceinline__ void f32_tile_epilogue(const GemmF32Args &g, const f32x16 (&acc)[2][JT], int n0, int m0, int wr, int wc,
                                                  int l31, int lh) {
    // Every load of the epilogue -- the bias quads (they depend on (q4, i) only) and, sub-tile by sub-tile, the residual quads of
    // this lane -- is issued BEFORE the first store of what it feeds: the output may be the residual updated in place, so hipcc
    // keeps a load behind every earlier store, and vmcnt counts stores too -- written load / store / load / store (the form until
    // round 6) this was eight dependent L2 round trips per sub-tile and lane under a saturated memory pipeline.
    float4 bq[4][2];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = (EPI == SCULPT_EPI_GEGLU) ? (i ? g.N : 0) + n0 + wr * 32 + 8 * q4 + 4 * lh : n0 + wr * 64 + i * 32 + 8 * q4 + 4 * lh;
            // (plain forms: a quad past N -- N is a multiple of 4: entirely in or out -- is never used; read a valid one instead)
            bq[q4][i] = *reinterpret_cast<const float4 *>(g.bias + ((EPI == SCULPT_EPI_GEGLU || n < g.N) ? n : 0));
        }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int m = m0 + wc * (32 * JT) + j * 32 + l31;
        if (m >= g.M) continue;
        if (EPI == SCULPT_EPI_GEGLU) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {  // register quad: rows 8*q4 + 4*lh + {0..3}
                // wave rows [wr*64, +32) = value group, [+32, +64) = gate group of output columns n0 + wr*32 ..
                const int n = n0 + wr * 32 + 8 * q4 + 4 * lh;
                const float4 bv = bq[q4][0], bg = bq[q4][1];
                const float bvs[4] = {bv.x, bv.y, bv.z, bv.w}, bgs[4] = {bg.x, bg.y, bg.z, bg.w};
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[0][j][4 * q4 + r] * g.alpha + bvs[r];
                    const float gt = acc[1][j][4 * q4 + r] * g.alpha + bgs[r];
                    o[r] = v * gelu_erf_exact(gt);
                }
                *reinterpret_cast<float4 *>(g.out + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            float4 rq[4][2];
            if (g.residual) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int n = n0 + wr * 64 + i * 32 + 8 * q4 + 4 * lh;
                        rq[q4][i] = *reinterpret_cast<const float4 *>(g.residual + (long)m * g.ldr + (n < g.N ? n : 0));
                    }
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int n = n0 + wr * 64 + i * 32 + 8 * q4 + 4 * lh;
                    if (n >= g.N) continue;  // N is a multiple of 4: a quad is entirely in or out
                    const float4 b4 = bq[q4][i];
                    const float bs[4] = {b4.x, b4.y, b4.z, b4.w};
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][4 * q4 + r] * g.alpha + bs[r];
                        if (EPI == SCULPT_EPI_GELU) v = gelu_erf_exact(v);
                        if (EPI == SCULPT_EPI_RELU) v = fmaxf(v, 0.f);
                        o[r] = v;
                    }
                    if (g.residual) {
                        const float4 rs = rq[q4][i];
                        o[0] += rs.x; o[1] += rs.y; o[2] += rs.z; o[3] += rs.w;
                    }
                    const bool tpart = n >= g.n_split;
                    if (!tpart && g.out) *reinterpret_cast<float4 *>(g.out + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
                    if (g.out_t && (tpart || g.n_split >= g.N)) {
                        const int nt0 = tpart ? n - g.n_split : n;
#pragma unroll
                        for (int r = 0; r < 4; ++r) g.out_t[(long)(nt0 + r) * g.ldt + m] = o[r];
                    }
                }
            }
        }
    }
}

// apply blockIdx.z's strides (a copy of the arguments per workgroup; all scalar)
__device__ __forceinline__ GemmF32Args f32_batch_entry(GemmF32Args g) {
    const long z = blockIdx.z;
    g.A += z * g.a_bs;
    g.W += z * g.w_bs;
    if (g.out) g.out += z * g.o_bs;
    if (g.out_t) g.out_t += z * g.o_bs;
    if (g.residual) g.residual += z * g.o_bs;
    return g;
}

// gemm.hip: a per-device page of zeros (`*count` floats) that stands in for a missing bias vector, so that the epilogue's bias
// loads are unconditional float4 loads (a per-element `bias ? bias[n] : 0` makes hipcc branch around every load and wait for it)
const float *zero_floats_page(long *count);

// gemm_l3.hip
int gemm_l3_launch(const GemmF32Args &g, int epilogue, int batch, hipStream_t st);

}  // namespace sculpt
