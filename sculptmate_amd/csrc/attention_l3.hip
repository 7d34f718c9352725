// Fused attention for the fast parity mode (TSR(precision="bf16l3")): fp32 Q / K / V^T in, fp32 O out, fp32 ARITHMETIC on the bf16
// matrix pipe -- both products, S = (c q) . k and O = P . v, take their operands split EXACTLY into three bf16 limbs and sum
// the six limb products of order >= 2^-16 in fp32 (gemm_l3.hip has the argument); the softmax (running maximum, exp2, row sums,
// rescale) is plain fp32 in registers.  Replaces the parity modes' three launches per attention (scores into an fp32
// [heads][Tq][Tk] scratch, row softmax, P V: 600 MB written and re-read three times per backbone self-attention) by one.
// Reference: F.scaled_dot_product_attention, TripoSR/tsr/models/transformer/attention.py:629-631 (fp32, no autocast); HF
// ViTSelfAttention's eager softmax(QK^T / 8) V.
//
// One workgroup = 4 waves = 128 queries of one head; a wave owns 32 queries (the query is the MFMA column, lane & 31, so a
// lane holds 2 x 16 scores of ONE query per 64-key tile and the softmax statistics are in-register + one cross-half exchange).
// Per 64-key tile:  K rows and V^T rows arrive as fp32 through registers (loaded one tile ahead), are split and written as
// limbs into a [limb][8-wide k chunk][row] LDS image (gemm_l3.hip's: a fragment is 512 contiguous bytes);
//   S^T = K . Q^T     48 MFMAs (2 row tiles x 4 k-steps x 6 limb products); the Q limbs stay in registers for the whole loop;
//                     key row m of a 32-row tile sits at LDS row m with bits 2 and 3 swapped, so that a lane's accumulator
//                     registers 8 g .. 8 g + 7 hold 8 CONTIGUOUS keys -- exactly one B fragment of the second product;
//   P = exp2(S - M)   fp32; split into three limbs in registers (no LDS round trip);
//   O^T += V^T . P^T  48 MFMAs.
// One 49-KiB LDS buffer, two barriers per tile; two workgroups per CU overlap one's split / softmax with the other's MFMAs.
#include <stdlib.h>

#include "attention_tile.h"
#include "limbs.h"

namespace sculpt {

typedef __bf16 albf16x2 __attribute__((ext_vector_type(2)));
typedef float alf32x2 __attribute__((ext_vector_type(2)));

static constexpr int AL_CS = 64 * 16 + 16;    // bytes from one k-chunk plane (64 rows x 16 B) to the next (+16: the 8-byte writes
                                              // of a 16-lane group land on 16 different 8-byte slots of the 128-byte bank row)
static constexpr int AL_LT = 8 * AL_CS;       // one limb of one operand tile (64 rows x 64 k)
static constexpr int AL_OP = 3 * AL_LT;       // one operand tile, three limbs

__device__ __forceinline__ unsigned al_cvt_pk(float lo, float hi) {
    const alf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, albf16x2));
}

// two fp32 values -> their three packed limb pairs
__device__ __forceinline__ void al_split2(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
#pragma clang fp contract(off)
    p1 = al_cvt_pk(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);   // exact
    p2 = al_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u); // exact, <= 8 bits
    p3 = al_cvt_pk(sa, sb);
}

typedef unsigned alu32x4 __attribute__((ext_vector_type(4)));

// eight fp32 values -> one MFMA operand fragment per limb
__device__ __forceinline__ void al_split8(const float (&x)[8], abf16x8 &f1, abf16x8 &f2, abf16x8 &f3) {
    alu32x4 v1, v2, v3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        al_split2(x[2 * i], x[2 * i + 1], a, b, c);
        v1[i] = a; v2[i] = b; v3[i] = c;
    }
    f1 = __builtin_bit_cast(abf16x8, v1);
    f2 = __builtin_bit_cast(abf16x8, v2);
    f3 = __builtin_bit_cast(abf16x8, v3);
}

__global__ __launch_bounds__(256, 2) void attention_l3_kernel(const float *__restrict__ Q, int ldq, const float *__restrict__ K, int ldk,
                                                              const float *__restrict__ Vt, int ldvt, float *__restrict__ O, int ldo,
                                                              int Tq, int Tk, float scale_log2e, unsigned char *__restrict__ O_lt,
                                                              int o_row0, int o_k8, AttnL3Batch ab) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AL_OP];   // [K limbs | V^T limbs]
    Q += blockIdx.z * ab.q_bs; K += blockIdx.z * ab.k_bs; Vt += blockIdx.z * ab.vt_bs;
    if (O) O += blockIdx.z * ab.o_bs;
    o_row0 += blockIdx.z * ab.o_row_bs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + qc;
    const int qld = min(q, Tq - 1);

    // Q limbs (B operand of the first product: B[k = 8 h + j][column = query]), scaled by softmax_scale * log2(e) in fp32
    abf16x8 qf[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);
        const float4 b = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8 + 4);
        const float x[8] = {a.x * scale_log2e, a.y * scale_log2e, a.z * scale_log2e, a.w * scale_log2e,
                            b.x * scale_log2e, b.y * scale_log2e, b.z * scale_log2e, b.w * scale_log2e};
        al_split8(x, qf[ks][0], qf[ks][1], qf[ks][2]);
    }

    // staging: a tile is 64 rows x 64 fp32 = 1024 float4 per operand; thread t takes quad t % 16 of rows t / 16 + 16 i
    const int sr = tid >> 4, sq = tid & 15;
    const float *Kh = K + head * 64 + 4 * sq;                       // + key * ldk
    const float *Vh = Vt + (long)(head * 64 + sr) * ldvt + 4 * sq;  // + 16 i * ldvt + key0
    // LDS byte offsets of this thread's 8-byte pieces: chunk = sq / 2, half = sq % 2; K rows go to their permuted position
    int kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = sr + 16 * i, m = kk & 31;
        const int pos = (kk & 32) | (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1);
        kofs[i] = (sq >> 1) * AL_CS + pos * 16 + (sq & 1) * 8;
    }
    const int vofs = AL_OP + (sq >> 1) * AL_CS + sr * 16 + (sq & 1) * 8;   // + 16 i rows = + 256 i bytes
    // fragment read offsets: k-step ks reads chunk 2 ks + h, rows 32 t + (lane & 31)
    const int fro = h * AL_CS + qc * 16;

    const int nt = (Tk + 63) / 64;
    float4 rk[4], rv[4];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(t * 64 + sr + 16 * i, Tk - 1);
            rk[i] = *reinterpret_cast<const float4 *>(Kh + (long)key * ldk);
            rv[i] = *reinterpret_cast<const float4 *>(Vh + (long)(16 * i) * ldvt + t * 64);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
    for (int t = 0; t < nt; ++t) {
        if (t > 0) __syncthreads();   // every wave has read the previous tile's fragments
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned a1, a2, a3, b1, b2, b3;
            al_split2(rk[i].x, rk[i].y, a1, a2, a3);
            al_split2(rk[i].z, rk[i].w, b1, b2, b3);
            unsigned char *d = smem + kofs[i];
            *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(d + AL_LT) = make_uint2(a2, b2);
            *reinterpret_cast<uint2 *>(d + 2 * AL_LT) = make_uint2(a3, b3);
            al_split2(rv[i].x, rv[i].y, a1, a2, a3);
            al_split2(rv[i].z, rv[i].w, b1, b2, b3);
            d = smem + vofs + i * 256;
            *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(d + AL_LT) = make_uint2(a2, b2);
            *reinterpret_cast<uint2 *>(d + 2 * AL_LT) = make_uint2(a3, b3);
        }
        __syncthreads();
        if (t + 1 < nt) gload(t + 1);   // the next tile travels while this one is multiplied

        // ---- S^T = K . Q^T (exponents of 2)
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            abf16x8 k0[3], k1[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                k0[l] = *reinterpret_cast<const abf16x8 *>(smem + fro + l * AL_LT + 2 * ks * AL_CS);
                k1[l] = *reinterpret_cast<const abf16x8 *>(smem + fro + l * AL_LT + 2 * ks * AL_CS + 32 * 16);
            }
#define AL_SIX(acc, a, b)                                                           \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0)
            AL_SIX(s0, k0, qf[ks]);
            AL_SIX(s1, k1, qf[ks]);
        }
        // lane (query, h): s{rt}[8 g + j] = key 64 t + 32 rt + 16 g + 8 h + j
        if (t * 64 + 64 > Tk) {   // ragged last tile (wave-uniform)
            const int kb = t * 64 + 8 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 7) + 16 * (r >> 3);
                if (key >= Tk) s0[r] = -INFINITY;
                if (key + 32 >= Tk) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);          // finite from the first tile on: every tile has at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            ps += s0[r] + s1[r];
        }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T . P^T: k-step kstep = 2 rt + g takes P registers 8 g .. 8 g + 7 of s{rt} (contiguous keys)
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (kstep < 2) ? s0[8 * (kstep & 1) + j] : s1[8 * (kstep & 1) + j];
            abf16x8 p[3], v0[3], v1[3];
            al_split8(x, p[0], p[1], p[2]);
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                v0[l] = *reinterpret_cast<const abf16x8 *>(smem + AL_OP + fro + l * AL_LT + 2 * kstep * AL_CS);
                v1[l] = *reinterpret_cast<const abf16x8 *>(smem + AL_OP + fro + l * AL_LT + 2 * kstep * AL_CS + 32 * 16);
            }
            AL_SIX(o0, v0, p);
            AL_SIX(o1, v1, p);
        }
#undef AL_SIX
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Tq && O_lt) {
        // the output as three bf16 limbs in the limb-tiled layout (limbs.h): the operand of the to_out Linear (gemm_l3p.hip);
        // row o_row0 + q of a matrix with o_k8 chunks per row, columns head * 64 ..
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float a[4] = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            const float b[4] = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            lt_store4(O_lt, o_k8, (long)o_row0 + q, head * 64 + 8 * g4 + 4 * h, a, ab.o_fmt);
            lt_store4(O_lt, o_k8, (long)o_row0 + q, head * 64 + 32 + 8 * g4 + 4 * h, b, ab.o_fmt);
        }
    } else if (q < Tq) {
        float *orow = O + (long)q * ldo + head * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // registers 4 g4 .. 4 g4 + 3 = d 8 g4 + 4 h + {0..3} (+ 32 for o1)
            *reinterpret_cast<float4 *>(orow + 8 * g4 + 4 * h) =
                make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + 8 * g4 + 4 * h) =
                make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same attention with the vector work issued in the shadows of the wave's own MFMAs (the default).  In the plain kernel a
// tile costs 96 MFMAs (3 072 matrix cycles) plus ~500 vector instructions that run while the matrix pipe waits (split of the next
// K / V^T rows 176, softmax ~160, split of P 176); a wave's own vector instructions right behind its own MFMA are free up to ~6
// per MFMA (tools/micro/mfma_fill.hip), another wave's are not (mfma_phase.hip).  So:
//   QK phase   8 groups of 6 MFMAs (k-step ks, key row tile rt); group g carries the split of float4 g of the NEXT tile's rows
//              (g < 4: K rows, else V^T rows) -- the limbs wait in 48 registers for the barrier at the end of the iteration;
//              the fragments of the following group are read into the registers of the one before it as they die;
//   softmax    as before (exposed: it needs all scores of the tile);
//   PV phase   4 k-steps of 2 x 6 MFMAs; the split of the 8 probabilities of k-step + 1 rides in k-step's groups.
// Stages are pinned inside their slots by empty asm statements and fenced with sched_barrier (gemm_l3.hip has the reasons).
// Same operands and the same order of every matrix sum as attention_l3_kernel; the softmax updates are written without fused
// multiply-adds here (fp contract off for the exact splits), so the two agree to fp32 rounding, not bit for bit.
// Measured (tools/time_l3_attention.py): 3072 x 3072 x 16 heads 262 -> 244 us = 0.38 of the bf16 peak by executed FLOPs -- the same
// region as the pipelined three-limb GEMM (0.43) and the density kernel (0.45): with every CU issuing MFMAs plus the split's
// vector work the chip is clock-bound, a tighter issue stream returns little.
// ---------------------------------------------------------------------------------------------------------------------
#define ALP_FENCE __builtin_amdgcn_sched_barrier(0)
#define ALP_MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define ALP_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define ALP_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// one group: six MFMAs (smallest terms first) with the three stages of one float4 split behind them; RD0 / RD1 / RD2 = the three
// fragment reads of the group after next (may be empty)
#define ALP_GROUP_SPLIT4(acc, A, B, X, P1, P2, P3, RD0, RD1, RD2)                                                                  \
    do {                                                                                                                           \
        float x0 = (X).x, x1 = (X).y, x2 = (X).z, x3 = (X).w, t0, t1, t2, t3, r0, r1, r2, r3;                                      \
        unsigned a1, b1, a2, b2, a3, b3;                                                                                           \
        ALP_PIN4(x0, x1, x2, x3);                                                                                                  \
        ALP_MF(acc, A[0], B[2]);                                                                                                   \
        a1 = al_cvt_pk(x0, x1); b1 = al_cvt_pk(x2, x3);                                                                            \
        t0 = __uint_as_float(a1 << 16); t1 = __uint_as_float(a1 & 0xffff0000u);                                                    \
        t2 = __uint_as_float(b1 << 16); t3 = __uint_as_float(b1 & 0xffff0000u);                                                    \
        ALP_PIN4(t0, t1, t2, t3);                                                                                                  \
        RD0;                                                                                                                       \
        ALP_FENCE;                                                                                                                 \
        ALP_MF(acc, A[2], B[0]);                                                                                                   \
        r0 = x0 - t0; r1 = x1 - t1; r2 = x2 - t2; r3 = x3 - t3;                                                                    \
        ALP_PIN4(r0, r1, r2, r3);                                                                                                  \
        RD1;                                                                                                                       \
        ALP_FENCE;                                                                                                                 \
        ALP_MF(acc, A[1], B[1]);                                                                                                   \
        a2 = al_cvt_pk(r0, r1); b2 = al_cvt_pk(r2, r3);                                                                            \
        t0 = __uint_as_float(a2 << 16); t1 = __uint_as_float(a2 & 0xffff0000u);                                                    \
        t2 = __uint_as_float(b2 << 16); t3 = __uint_as_float(b2 & 0xffff0000u);                                                    \
        ALP_PIN4(t0, t1, t2, t3);                                                                                                  \
        RD2;                                                                                                                       \
        ALP_FENCE;                                                                                                                 \
        ALP_MF(acc, A[0], B[1]);                                                                                                   \
        r0 = r0 - t0; r1 = r1 - t1; r2 = r2 - t2; r3 = r3 - t3;                                                                    \
        ALP_PIN4(r0, r1, r2, r3);                                                                                                  \
        ALP_FENCE;                                                                                                                 \
        ALP_MF(acc, A[1], B[0]);                                                                                                   \
        a3 = al_cvt_pk(r0, r1); b3 = al_cvt_pk(r2, r3);                                                                            \
        ALP_PIN2(a3, b3);                                                                                                          \
        ALP_FENCE;                                                                                                                 \
        ALP_MF(acc, A[0], B[0]);                                                                                                   \
        ALP_FENCE;                                                                                                                 \
        P1 = make_uint2(a1, b1); P2 = make_uint2(a2, b2); P3 = make_uint2(a3, b3);                                                 \
    } while (0)

// the same six MFMAs without a split (a group that has no staging work left)
#define ALP_GROUP_PLAIN(acc, A, B, RD0, RD1, RD2)                                                                                  \
    do {                                                                                                                           \
        ALP_MF(acc, A[0], B[2]); RD0; ALP_FENCE;                                                                                   \
        ALP_MF(acc, A[2], B[0]); RD1; ALP_FENCE;                                                                                   \
        ALP_MF(acc, A[1], B[1]); RD2; ALP_FENCE;                                                                                   \
        ALP_MF(acc, A[0], B[1]); ALP_FENCE;                                                                                        \
        ALP_MF(acc, A[1], B[0]); ALP_FENCE;                                                                                        \
        ALP_MF(acc, A[0], B[0]); ALP_FENCE;                                                                                        \
    } while (0)

// NW = 8 waves = 256 queries per workgroup: a thread stages two float4 per operand and tile instead of four, which is what lets
// the limbs of the next tile (24 registers) sit beside 48 Q-limb, 64 accumulator and 48 fragment / probability registers
// without spilling (the 4-wave form needs 48 and spilled 14 registers into the loop); a 3072-query head is 12 workgroups, 192
// per attention -- as many CUs busy with 8 waves each as the 4-wave form keeps busy with 8 (two workgroups of 4).
template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_l3_pipe_kernel(const float *__restrict__ Q, int ldq, const float *__restrict__ K,
                                                                   int ldk, const float *__restrict__ Vt, int ldvt,
                                                                   float *__restrict__ O, int ldo, int Tq, int Tk, float scale_log2e,
                                                                   unsigned char *__restrict__ O_lt, int o_row0, int o_k8, AttnL3Batch ab) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AL_OP];   // [K limbs | V^T limbs]
    Q += blockIdx.z * ab.q_bs; K += blockIdx.z * ab.k_bs; Vt += blockIdx.z * ab.vt_bs;
    if (O) O += blockIdx.z * ab.o_bs;
    o_row0 += blockIdx.z * ab.o_row_bs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    constexpr int NS = 1024 / (NW * 64);   // float4 per thread, operand and tile: 2
    static_assert(NS == 2, "the pipelined form is built for 8 waves");
    const int q = blockIdx.x * (NW * 32) + wave * 32 + qc;
    const int qld = min(q, Tq - 1);

    abf16x8 qf[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);
        const float4 b = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8 + 4);
        const float x[8] = {a.x * scale_log2e, a.y * scale_log2e, a.z * scale_log2e, a.w * scale_log2e,
                            b.x * scale_log2e, b.y * scale_log2e, b.z * scale_log2e, b.w * scale_log2e};
        al_split8(x, qf[ks][0], qf[ks][1], qf[ks][2]);
    }

    // staging: thread t takes quad t % 16 of rows t / 16 + 32 i (i = 0, 1); K row kk = sr + 32 i goes to position 32 i + pk(sr)
    const int sr = tid >> 4, sq = tid & 15;
    const float *Kh = K + head * 64 + 4 * sq;
    const float *Vh = Vt + (long)(head * 64 + sr) * ldvt + 4 * sq;
    const int kofs = (sq >> 1) * AL_CS + ((sr & 0x13) | ((sr & 4) << 1) | ((sr & 8) >> 1)) * 16 + (sq & 1) * 8;   // + 512 i
    const int vofs = AL_OP + (sq >> 1) * AL_CS + sr * 16 + (sq & 1) * 8;                                            // + 512 i
    const int fro = h * AL_CS + qc * 16;

    const int nt = (Tk + 63) / 64;
    float4 rk[NS], rv[NS];
    uint2 pk[NS][3], pv[NS][3];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int key = min(t * 64 + sr + 32 * i, Tk - 1);
            rk[i] = *reinterpret_cast<const float4 *>(Kh + (long)key * ldk);
            rv[i] = *reinterpret_cast<const float4 *>(Vh + (long)(32 * i) * ldvt + t * 64);
        }
    };
    auto write_all = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                *reinterpret_cast<uint2 *>(smem + kofs + i * 512 + l * AL_LT) = pk[i][l];
                *reinterpret_cast<uint2 *>(smem + vofs + i * 512 + l * AL_LT) = pv[i][l];
            }
        }
    };
#define ALP_RDK(dst, l, ks_, up) dst = *reinterpret_cast<const abf16x8 *>(smem + fro + (l) * AL_LT + 2 * (ks_) * AL_CS + (up) * (32 * 16))
#define ALP_RDV(dst, l, ks_, up) dst = *reinterpret_cast<const abf16x8 *>(smem + AL_OP + fro + (l) * AL_LT + 2 * (ks_) * AL_CS + (up) * (32 * 16))

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        unsigned a1, a2, a3, b1, b2, b3;
        al_split2(rk[i].x, rk[i].y, a1, a2, a3);
        al_split2(rk[i].z, rk[i].w, b1, b2, b3);
        pk[i][0] = make_uint2(a1, b1); pk[i][1] = make_uint2(a2, b2); pk[i][2] = make_uint2(a3, b3);
        al_split2(rv[i].x, rv[i].y, a1, a2, a3);
        al_split2(rv[i].z, rv[i].w, b1, b2, b3);
        pv[i][0] = make_uint2(a1, b1); pv[i][1] = make_uint2(a2, b2); pv[i][2] = make_uint2(a3, b3);
    }
    write_all();
    __syncthreads();
    if (nt > 1) gload(1);

    for (int t = 0; t < nt; ++t) {
        // ---- QK phase: S^T = K . Q^T with the split of the next tile's rows in the MFMA shadows
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        abf16x8 k0[3], k1[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) { ALP_RDK(k0[l], l, 0, 0); ALP_RDK(k1[l], l, 0, 1); }
        ALP_FENCE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // group (ks, rt = 0) uses k0; k1 of this k-step is already on its way (read during the previous group).
            // k-steps 0 / 1 carry the split of the next tile's two K / two V^T float4 of this thread.
            if (ks < NS) ALP_GROUP_SPLIT4(s0, k0, qf[ks], rk[ks < NS ? ks : 0], pk[ks < NS ? ks : 0][0], pk[ks < NS ? ks : 0][1], pk[ks < NS ? ks : 0][2], (void)0, (void)0, (void)0);
            else ALP_GROUP_PLAIN(s0, k0, qf[ks], (void)0, (void)0, (void)0);
            // group (ks, rt = 1) uses k1; k0 is dead: the next k-step's k0 is read into it
            if (ks < 3) {
                if (ks < NS) ALP_GROUP_SPLIT4(s1, k1, qf[ks], rv[ks < NS ? ks : 0], pv[ks < NS ? ks : 0][0], pv[ks < NS ? ks : 0][1], pv[ks < NS ? ks : 0][2],
                                              ALP_RDK(k0[0], 0, ks + 1, 0), ALP_RDK(k0[1], 1, ks + 1, 0), ALP_RDK(k0[2], 2, ks + 1, 0));
                else ALP_GROUP_PLAIN(s1, k1, qf[ks], ALP_RDK(k0[0], 0, ks + 1, 0), ALP_RDK(k0[1], 1, ks + 1, 0), ALP_RDK(k0[2], 2, ks + 1, 0));
                // k1 is dead now: the next k-step's k1 (its reads land during the next group, which uses k0)
#pragma unroll
                for (int l = 0; l < 3; ++l) ALP_RDK(k1[l], l, ks + 1, 1);
                ALP_FENCE;
            } else {
                ALP_GROUP_PLAIN(s1, k1, qf[ks], (void)0, (void)0, (void)0);
            }
        }
        // ---- softmax (fp32): lane (query, h): s{rt}[8 g + j] = key 64 t + 32 rt + 16 g + 8 h + j
        if (t * 64 + 64 > Tk) {
            const int kb = t * 64 + 8 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 7) + 16 * (r >> 3);
                if (key >= Tk) s0[r] = -INFINITY;
                if (key + 32 >= Tk) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            ps += s0[r] + s1[r];
        }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- PV phase: O^T += V^T . P^T; the split of k-step + 1's probabilities in the shadows of k-step's MFMAs
        abf16x8 p[3], v0[3], v1[3];
        {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = s0[j];
            al_split8(x, p[0], p[1], p[2]);
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) { ALP_RDV(v0[l], l, 0, 0); ALP_RDV(v1[l], l, 0, 1); }
        ALP_FENCE;
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
            alu32x4 n1, n2, n3;   // the next k-step's P limbs, built pair by pair
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (kstep + 1 < 2) ? s0[8 * ((kstep + 1) & 1) + j] : s1[8 * ((kstep + 1) & 1) + j];
            const bool more = kstep < 3;
#define ALP_PAIR(i_)                                                                                    \
    do {                                                                                                \
        if (more) {                                                                                     \
            float ya = y[2 * (i_)], yb = y[2 * (i_) + 1];                                                \
            ALP_PIN2(ya, yb);                                                                            \
            unsigned c1, c2, c3;                                                                         \
            al_split2(ya, yb, c1, c2, c3);                                                               \
            asm volatile("" : "+v"(c1), "+v"(c2), "+v"(c3));                                             \
            n1[i_] = c1; n2[i_] = c2; n3[i_] = c3;                                                       \
        }                                                                                               \
    } while (0)
            // group (kstep, dt = 0): v0; during it v1 lands (read one group ago) -- six MFMAs, two pair splits
            ALP_MF(o0, v0[0], p[2]); ALP_PAIR(0); ALP_FENCE;
            ALP_MF(o0, v0[2], p[0]); ALP_FENCE;
            ALP_MF(o0, v0[1], p[1]); ALP_PAIR(1); ALP_FENCE;
            ALP_MF(o0, v0[0], p[1]); ALP_FENCE;
            ALP_MF(o0, v0[1], p[0]); ALP_FENCE;
            ALP_MF(o0, v0[0], p[0]); ALP_FENCE;
            // group (kstep, dt = 1): v1; v0 is dead: the next k-step's v0 is read into it
            ALP_MF(o1, v1[0], p[2]); ALP_PAIR(2); if (more) ALP_RDV(v0[0], 0, kstep + 1, 0); ALP_FENCE;
            ALP_MF(o1, v1[2], p[0]); if (more) ALP_RDV(v0[1], 1, kstep + 1, 0); ALP_FENCE;
            ALP_MF(o1, v1[1], p[1]); ALP_PAIR(3); if (more) ALP_RDV(v0[2], 2, kstep + 1, 0); ALP_FENCE;
            ALP_MF(o1, v1[0], p[1]); ALP_FENCE;
            ALP_MF(o1, v1[1], p[0]); ALP_FENCE;
            ALP_MF(o1, v1[0], p[0]); ALP_FENCE;
            if (more) {
#pragma unroll
                for (int l = 0; l < 3; ++l) ALP_RDV(v1[l], l, kstep + 1, 1);
                p[0] = __builtin_bit_cast(abf16x8, n1);
                p[1] = __builtin_bit_cast(abf16x8, n2);
                p[2] = __builtin_bit_cast(abf16x8, n3);
                ALP_FENCE;
            }
#undef ALP_PAIR
        }
        if (t + 1 < nt) {
            __syncthreads();   // every wave has read this tile's fragments
            write_all();
            if (t + 2 < nt) gload(t + 2);
            __syncthreads();
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Tq && O_lt) {
        // the output as three bf16 limbs in the limb-tiled layout (limbs.h): the operand of the to_out Linear (gemm_l3p.hip);
        // row o_row0 + q of a matrix with o_k8 chunks per row, columns head * 64 ..
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float a[4] = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            const float b[4] = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            lt_store4(O_lt, o_k8, (long)o_row0 + q, head * 64 + 8 * g4 + 4 * h, a, ab.o_fmt);
            lt_store4(O_lt, o_k8, (long)o_row0 + q, head * 64 + 32 + 8 * g4 + 4 * h, b, ab.o_fmt);
        }
    } else if (q < Tq) {
        float *orow = O + (long)q * ldo + head * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // registers 4 g4 .. 4 g4 + 3 = d 8 g4 + 4 h + {0..3} (+ 32 for o1)
            *reinterpret_cast<float4 *>(orow + 8 * g4 + 4 * h) =
                make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + 8 * g4 + 4 * h) =
                make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
        }
    }
}
#undef ALP_FENCE
#undef ALP_MF
#undef ALP_PIN4
#undef ALP_PIN2
#undef ALP_GROUP_SPLIT4
#undef ALP_GROUP_PLAIN
#undef ALP_RDK
#undef ALP_RDV

}  // namespace sculpt

using namespace sculpt;

static int attention_l3_go(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, float *O, int ldo, void *O_lt,
                           int o_row0, int o_k8, int Tq, int Tk, int heads, float scale, sculpt_stream_t stream, int batch = 1,
                           AttnL3Batch ab = AttnL3Batch{0, 0, 0, 0, 0, 0}, int two_fp16_limbs = 0) {
    SC_REQUIRE(batch >= 1 && batch <= 65535, "attention_f32_l3: bad batch %d", batch);
    SC_REQUIRE(ab.o_fmt == LT_BF16X3 || ab.o_fmt == LT_F16X2, "attention_f32_l3: unknown limb format %d", ab.o_fmt);
    SC_REQUIRE(batch == 1 || (ab.q_bs % 4 == 0 && ab.k_bs % 4 == 0 && ab.vt_bs % 4 == 0 && ab.o_bs % 4 == 0 && ab.o_row_bs >= 0),
               "attention_f32_l3: batch strides must be multiples of 4 elements");
    SC_REQUIRE(Q && K && Vt && (O || O_lt), "attention_f32_l3: null argument");
    SC_REQUIRE(Tq >= 1 && Tk >= 1 && heads >= 1 && heads <= 65535, "attention_f32_l3: bad shape Tq=%d Tk=%d heads=%d", Tq, Tk, heads);
    SC_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldvt % 4 == 0 && ldo % 4 == 0, "attention_f32_l3: row strides must keep 16-byte alignment");
    SC_REQUIRE(ldvt >= ((Tk + 63) / 64) * 64, "attention_f32_l3: ldvt=%d must be >= round_up(Tk=%d, 64) (finite padding columns)", ldvt, Tk);
    SC_REQUIRE(scale > 0.f && scale == scale, "attention_f32_l3: scale must be positive");
    SC_REQUIRE(!O_lt || (o_row0 >= 0 && o_k8 >= heads * 8 && ((uintptr_t)O_lt & 15) == 0),
               "attention_f32_l3_limbs: the limb output needs o_row0 >= 0, >= heads * 64 columns and 16-byte alignment");
    unsigned char *olt = reinterpret_cast<unsigned char *>(O_lt);
    // the pipelined 8-wave form (256 queries per workgroup) where it keeps at least 2/3 of the CUs busy -- the backbone's 3072
    // queries x 16 heads = 192 workgroups --; otherwise (the image tokenizer: 1025 queries x 12 heads) the plain 4-wave form
    // SCULPT_ATTN_FORM tokens l3pipe / nol3pipe: always / never the pipelined form (A/B, tests; read per call)
    const int fpipe = form_has("SCULPT_ATTN_FORM", "l3pipe") ? 1 : (form_has("SCULPT_ATTN_FORM", "nol3pipe") ? 0 : -1);
    // (the two-limb arithmetic exists in the pipelined form only, and that form wins there even on the image tokenizer's 60
    // workgroups: 53 against 59 us on three limbs and 4 waves)
    const bool pipe = fpipe >= 0 ? fpipe != 0 : (two_fp16_limbs || (long)cdiv(Tq, 256) * heads * batch * 3 >= 2L * num_cus());
    // two fp16 limbs per operand (attention_l2.hip: half the matrix work) where the pipelined form runs; the small launches (the
    // image tokenizer's) stay on the three-limb 4-wave kernel
    if (pipe && two_fp16_limbs)
        attention_l2_pipe_launch(dim3(cdiv(Tq, 256), heads, batch), as_stream(stream), Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk,
                                 scale * 1.44269504088896340736f, olt, o_row0, o_k8, ab);
    else if (!pipe)
        hipLaunchKernelGGL(attention_l3_kernel, dim3(cdiv(Tq, 128), heads, batch), dim3(256), 0, as_stream(stream), Q, ldq, K, ldk, Vt,
                           ldvt, O, ldo, Tq, Tk, scale * 1.44269504088896340736f, olt, o_row0, o_k8, ab);
    else
        hipLaunchKernelGGL(attention_l3_pipe_kernel<8>, dim3(cdiv(Tq, 256), heads, batch), dim3(512), 0, as_stream(stream), Q, ldq, K, ldk,
                           Vt, ldvt, O, ldo, Tq, Tk, scale * 1.44269504088896340736f, olt, o_row0, o_k8, ab);
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sculpt_attention_f32_l3(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, float *O, int ldo,
                                       int Tq, int Tk, int heads, float scale, sculpt_stream_t stream) {
    SC_REQUIRE(O, "attention_f32_l3: null argument");
    return attention_l3_go(Q, ldq, K, ldk, Vt, ldvt, O, ldo, nullptr, 0, 0, Tq, Tk, heads, scale, stream);
}

extern "C" int sculpt_attention_f32_l3_limbs(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, void *O_lt,
                                             int format, int o_row0, int o_cols, int Tq, int Tk, int heads, float scale,
                                             int two_fp16_limbs, sculpt_stream_t stream) {
    SC_REQUIRE(O_lt && o_cols % 32 == 0, "attention_f32_l3_limbs: null output or o_cols=%d not a multiple of 32", o_cols);
    return attention_l3_go(Q, ldq, K, ldk, Vt, ldvt, nullptr, 0, O_lt, o_row0, o_cols / 8, Tq, Tk, heads, scale, stream, 1,
                           AttnL3Batch{0, 0, 0, 0, 0, format}, two_fp16_limbs);
}

/* `batch` attentions of one shape in ONE launch (grid z): entry b reads Q + b*q_bs, K + b*k_bs, Vt + b*vt_bs (element strides,
 * multiples of 4; vt_bs may be a column offset into one [heads*64][ldvt] array) and writes O + b*o_bs, or -- O_lt given, O null --
 * rows o_row0 + b*o_row_bs .. of the limb-tiled output. */
extern "C" int sculpt_attention_f32_l3_batched(const float *Q, int ldq, int64_t q_bs, const float *K, int ldk, int64_t k_bs,
                                               const float *Vt, int ldvt, int64_t vt_bs, float *O, int ldo, int64_t o_bs, void *O_lt,
                                               int format, int o_row0, int o_row_bs, int o_cols, int Tq, int Tk, int heads, int batch,
                                               float scale, int two_fp16_limbs, sculpt_stream_t stream) {
    SC_REQUIRE((O != nullptr) != (O_lt != nullptr), "attention_f32_l3_batched: exactly one of O / O_lt");
    SC_REQUIRE(!O_lt || o_cols % 32 == 0, "attention_f32_l3_batched: o_cols=%d not a multiple of 32", o_cols);
    return attention_l3_go(Q, ldq, K, ldk, Vt, ldvt, O, ldo, O_lt, o_row0, o_cols / 8, Tq, Tk, heads, scale, stream, batch,
                           AttnL3Batch{(long)q_bs, (long)k_bs, (long)vt_bs, (long)o_bs, o_row_bs, O_lt ? format : 0}, two_fp16_limbs);
}
