// Fused attention of the tolerance mode TSR(precision="fp16l2"): attention_l3.hip's pipelined kernel with TWO fp16 limbs per operand
// instead of three bf16 limbs -- fp32 Q / K / V^T in, fp32 (or limb-tiled) O out; both products, S = (c q) . k and O = P . v, take
// their operands as x ~ h1 + h2 (h1 = fp16(x), h2 = fp16(x - h1): 22 significant bits; every operand here is O(1): scaled queries,
// keys, values, probabilities in [0, 1]) and sum the three limb products h2.g1 + h1.g2 + h1.g1 in fp32 on
// v_mfma_f32_32x32x16_f16 (each exact: 11 x 11 bits); the softmax is plain fp32 in registers.  Half the matrix work of the
// three-limb kernel: 48 MFMAs per 64-key tile and wave instead of 96.  Measured against it on the model: the scene code stays
// at fp32 rounding noise from the exact-fp32 mode (tests/test_gpu_l3p.py, tools/time_parity_modes.py).
// Reference: F.scaled_dot_product_attention, TripoSR/tsr/models/transformer/attention.py:629-631 (fp32, no autocast).
//
// Same tiles, staging, LDS image ([limb][8-wide k chunk][row], key rows with bits 2 and 3 swapped so that a lane's accumulator
// registers hold 8 contiguous keys = one B fragment of the second product) and software pipeline as attention_l3_pipe_kernel<8>:
//   QK phase   8 groups of 3 MFMAs (k-step ks, key row tile rt); group g < 4 carries the two-limb split of one float4 of the NEXT
//              tile's rows (K rows, then V^T rows) in its shadows -- 6 / 4 / 2 vector instructions behind the three MFMAs;
//   softmax    only the row maximum, the rescale of O and the first k-step's eight probabilities are exposed;
//   PV phase   4 k-steps of 2 x 3 MFMAs; the exponentials, row sums and the split of k-step + 1's eight probabilities (four pairs
//              of ~12 instructions) ride behind four of the six (self-attention 164 -> 156 us against the whole softmax up front).
// One 33-KiB LDS buffer (two limbs x two operands), two barriers per tile, 8 waves = 256 queries per workgroup.
#include <stdlib.h>

#include "attention_tile.h"
#include "limbs.h"

namespace sculpt {

typedef _Float16 ahf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ahf16x2 __attribute__((ext_vector_type(2)));
typedef float ahf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ahu32x4 __attribute__((ext_vector_type(4)));

static constexpr int AH_CS = 64 * 16 + 16;    // bytes from one k-chunk plane (64 rows x 16 B) to the next (attention_l3.hip: AL_CS)
static constexpr int AH_LT = 8 * AH_CS;       // one limb of one operand tile (64 rows x 64 k)
static constexpr int AH_OP = 2 * AH_LT;       // one operand tile, two limbs

__device__ __forceinline__ ahf16x2 ah_cvt_pk(float lo, float hi) {
    const ahf32x2 v = {lo, hi};
    return __builtin_convertvector(v, ahf16x2);   // v_cvt_pk_f16_f32, round to nearest even
}

// two fp32 values -> their two packed limb pairs
__device__ __forceinline__ void ah_split2(float a, float b, unsigned &p1, unsigned &p2) {
#pragma clang fp contract(off)
    const ahf16x2 h = ah_cvt_pk(a, b);
    const float ra = a - (float)h[0], rb = b - (float)h[1];   // exact
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, ah_cvt_pk(ra, rb));
}

// eight fp32 values -> one MFMA operand fragment per limb
__device__ __forceinline__ void ah_split8(const float (&x)[8], ahf16x8 &f1, ahf16x8 &f2) {
    ahu32x4 v1, v2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b;
        ah_split2(x[2 * i], x[2 * i + 1], a, b);
        v1[i] = a; v2[i] = b;
    }
    f1 = __builtin_bit_cast(ahf16x8, v1);
    f2 = __builtin_bit_cast(ahf16x8, v2);
}

#define AHP_FENCE __builtin_amdgcn_sched_barrier(0)
#define AHP_MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define AHP_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define AHP_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// one group: three MFMAs (smallest terms first) with the three stages of one float4's two-limb split behind them (the empty asm
// statements pin each stage inside its slot, attention_l3.hip / gemm_l3.hip have the reasons); RD0 / RD1 = the two fragment reads
// of the group after next (may be empty)
#define AHP_GROUP_SPLIT4(acc, A, B, X, P1, P2, RD0, RD1)                                                                           \
    do {                                                                                                                           \
        float x0 = (X).x, x1 = (X).y, x2 = (X).z, x3 = (X).w, t0, t1, t2, t3, r0, r1, r2, r3;                                      \
        ahf16x2 a1, b1, a2, b2;                                                                                                    \
        AHP_PIN4(x0, x1, x2, x3);                                                                                                  \
        AHP_MF(acc, A[1], B[0]);                                                                                                   \
        a1 = ah_cvt_pk(x0, x1); b1 = ah_cvt_pk(x2, x3);                                                                            \
        t0 = (float)a1[0]; t1 = (float)a1[1]; t2 = (float)b1[0]; t3 = (float)b1[1];                                                \
        AHP_PIN4(t0, t1, t2, t3);                                                                                                  \
        RD0;                                                                                                                       \
        AHP_FENCE;                                                                                                                 \
        AHP_MF(acc, A[0], B[1]);                                                                                                   \
        r0 = x0 - t0; r1 = x1 - t1; r2 = x2 - t2; r3 = x3 - t3;   /* exact */                                                      \
        AHP_PIN4(r0, r1, r2, r3);                                                                                                  \
        RD1;                                                                                                                       \
        AHP_FENCE;                                                                                                                 \
        AHP_MF(acc, A[0], B[0]);                                                                                                   \
        a2 = ah_cvt_pk(r0, r1); b2 = ah_cvt_pk(r2, r3);                                                                            \
        {                                                                                                                          \
            unsigned ua = __builtin_bit_cast(unsigned, a2), ub = __builtin_bit_cast(unsigned, b2);                                 \
            AHP_PIN2(ua, ub);                                                                                                      \
            P2 = make_uint2(ua, ub);                                                                                               \
        }                                                                                                                          \
        AHP_FENCE;                                                                                                                 \
        P1 = make_uint2(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1));                                       \
    } while (0)

// the same three MFMAs without a split (a group that has no staging work left)
#define AHP_GROUP_PLAIN(acc, A, B, RD0, RD1)                                                                                       \
    do {                                                                                                                           \
        AHP_MF(acc, A[1], B[0]); RD0; AHP_FENCE;                                                                                   \
        AHP_MF(acc, A[0], B[1]); RD1; AHP_FENCE;                                                                                   \
        AHP_MF(acc, A[0], B[0]); AHP_FENCE;                                                                                        \
    } while (0)

template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_l2_pipe_kernel(const float *__restrict__ Q, int ldq, const float *__restrict__ K,
                                                                   int ldk, const float *__restrict__ Vt, int ldvt,
                                                                   float *__restrict__ O, int ldo, int Tq, int Tk, float scale_log2e,
                                                                   unsigned char *__restrict__ O_lt, int o_row0, int o_k8, AttnL3Batch ab) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AH_OP];   // [K limbs | V^T limbs]
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

    ahf16x8 qf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);
        const float4 b = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8 + 4);
        const float x[8] = {a.x * scale_log2e, a.y * scale_log2e, a.z * scale_log2e, a.w * scale_log2e,
                            b.x * scale_log2e, b.y * scale_log2e, b.z * scale_log2e, b.w * scale_log2e};
        ah_split8(x, qf[ks][0], qf[ks][1]);
    }

    // staging: thread t takes quad t % 16 of rows t / 16 + 32 i (i = 0, 1); K row kk = sr + 32 i goes to position 32 i + pk(sr)
    const int sr = tid >> 4, sq = tid & 15;
    const float *Kh = K + head * 64 + 4 * sq;
    const float *Vh = Vt + (long)(head * 64 + sr) * ldvt + 4 * sq;
    const int kofs = (sq >> 1) * AH_CS + ((sr & 0x13) | ((sr & 4) << 1) | ((sr & 8) >> 1)) * 16 + (sq & 1) * 8;   // + 512 i
    const int vofs = AH_OP + (sq >> 1) * AH_CS + sr * 16 + (sq & 1) * 8;                                            // + 512 i
    const int fro = h * AH_CS + qc * 16;

    const int nt = (Tk + 63) / 64;
    float4 rk[NS], rv[NS];
    uint2 pk[NS][2], pv[NS][2];
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
            for (int l = 0; l < 2; ++l) {
                *reinterpret_cast<uint2 *>(smem + kofs + i * 512 + l * AH_LT) = pk[i][l];
                *reinterpret_cast<uint2 *>(smem + vofs + i * 512 + l * AH_LT) = pv[i][l];
            }
        }
    };
#define AHP_RDK(dst, l, ks_, up) dst = *reinterpret_cast<const ahf16x8 *>(smem + fro + (l) * AH_LT + 2 * (ks_) * AH_CS + (up) * (32 * 16))
#define AHP_RDV(dst, l, ks_, up) dst = *reinterpret_cast<const ahf16x8 *>(smem + AH_OP + fro + (l) * AH_LT + 2 * (ks_) * AH_CS + (up) * (32 * 16))

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        unsigned a1, a2, b1, b2;
        ah_split2(rk[i].x, rk[i].y, a1, a2);
        ah_split2(rk[i].z, rk[i].w, b1, b2);
        pk[i][0] = make_uint2(a1, b1); pk[i][1] = make_uint2(a2, b2);
        ah_split2(rv[i].x, rv[i].y, a1, a2);
        ah_split2(rv[i].z, rv[i].w, b1, b2);
        pv[i][0] = make_uint2(a1, b1); pv[i][1] = make_uint2(a2, b2);
    }
    write_all();
    __syncthreads();
    if (nt > 1) gload(1);

    for (int t = 0; t < nt; ++t) {
        // ---- QK phase: S^T = K . Q^T with the split of the next tile's rows in the MFMA shadows
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        ahf16x8 k0[2], k1[2];
#pragma unroll
        for (int l = 0; l < 2; ++l) { AHP_RDK(k0[l], l, 0, 0); AHP_RDK(k1[l], l, 0, 1); }
        AHP_FENCE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // group (ks, rt = 0) uses k0; k1 of this k-step is already on its way (read during the previous group).
            // k-steps 0 / 1 carry the split of the next tile's two K / two V^T float4 of this thread.
            if (ks < NS) AHP_GROUP_SPLIT4(s0, k0, qf[ks], rk[ks < NS ? ks : 0], pk[ks < NS ? ks : 0][0], pk[ks < NS ? ks : 0][1], (void)0, (void)0);
            else AHP_GROUP_PLAIN(s0, k0, qf[ks], (void)0, (void)0);
            // group (ks, rt = 1) uses k1; k0 is dead: the next k-step's k0 is read into it
            if (ks < 3) {
                if (ks < NS) AHP_GROUP_SPLIT4(s1, k1, qf[ks], rv[ks < NS ? ks : 0], pv[ks < NS ? ks : 0][0], pv[ks < NS ? ks : 0][1],
                                              AHP_RDK(k0[0], 0, ks + 1, 0), AHP_RDK(k0[1], 1, ks + 1, 0));
                else AHP_GROUP_PLAIN(s1, k1, qf[ks], AHP_RDK(k0[0], 0, ks + 1, 0), AHP_RDK(k0[1], 1, ks + 1, 0));
                // k1 is dead now: the next k-step's k1 (its reads land during the next group, which uses k0)
#pragma unroll
                for (int l = 0; l < 2; ++l) AHP_RDK(k1[l], l, ks + 1, 1);
                AHP_FENCE;
            } else {
                AHP_GROUP_PLAIN(s1, k1, qf[ks], (void)0, (void)0);
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // only the first k-step's eight probabilities are made here; the exponentials, row sums and splits of k-step + 1 ride in
        // the shadows of k-step's MFMAs (the whole softmax up front left the matrix pipe idle for ~30 % of a tile)
        float psa = 0.f, psb = 0.f;

        // ---- PV phase: O^T += V^T . P^T
        ahf16x8 p[2], v0[2], v1[2];
        {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] = __builtin_amdgcn_exp2f(s0[j] - m_new);
                if (j & 1) psb += x[j]; else psa += x[j];
            }
            ah_split8(x, p[0], p[1]);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) { AHP_RDV(v0[l], l, 0, 0); AHP_RDV(v1[l], l, 0, 1); }
        AHP_FENCE;
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
            ahu32x4 n1, n2;   // the next k-step's P limbs, built pair by pair
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (kstep + 1 < 2) ? s0[8 * ((kstep + 1) & 1) + j] : s1[8 * ((kstep + 1) & 1) + j];
            const bool more = kstep < 3;
#define AHP_PAIR(i_)                                                                                    \
    do {                                                                                                \
        if (more) {                                                                                     \
            float ya = y[2 * (i_)] - m_new, yb = y[2 * (i_) + 1] - m_new;                                \
            AHP_PIN2(ya, yb);                                                                            \
            ya = __builtin_amdgcn_exp2f(ya); yb = __builtin_amdgcn_exp2f(yb);                            \
            psa += ya; psb += yb;                                                                        \
            unsigned c1, c2;                                                                             \
            ah_split2(ya, yb, c1, c2);                                                                   \
            asm volatile("" : "+v"(c1), "+v"(c2), "+v"(psa), "+v"(psb));                                 \
            n1[i_] = c1; n2[i_] = c2;                                                                    \
        }                                                                                               \
    } while (0)
            // group (kstep, dt = 0): v0; during it v1 lands (read one group ago) -- three MFMAs, two pair splits
            AHP_MF(o0, v0[1], p[0]); AHP_PAIR(0); AHP_FENCE;
            AHP_MF(o0, v0[0], p[1]); AHP_PAIR(1); AHP_FENCE;
            AHP_MF(o0, v0[0], p[0]); AHP_FENCE;
            // group (kstep, dt = 1): v1; v0 is dead: the next k-step's v0 is read into it
            AHP_MF(o1, v1[1], p[0]); AHP_PAIR(2); if (more) AHP_RDV(v0[0], 0, kstep + 1, 0); AHP_FENCE;
            AHP_MF(o1, v1[0], p[1]); AHP_PAIR(3); if (more) AHP_RDV(v0[1], 1, kstep + 1, 0); AHP_FENCE;
            AHP_MF(o1, v1[0], p[0]); AHP_FENCE;
            if (more) {
#pragma unroll
                for (int l = 0; l < 2; ++l) AHP_RDV(v1[l], l, kstep + 1, 1);
                p[0] = __builtin_bit_cast(ahf16x8, n1);
                p[1] = __builtin_bit_cast(ahf16x8, n2);
                AHP_FENCE;
            }
#undef AHP_PAIR
        }
        l_run = l_run * alpha + (psa + psb);
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
        // the output as limbs in the limb-tiled layout (limbs.h): the operand of the to_out Linear (gemm_l3p.hip);
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
#undef AHP_FENCE
#undef AHP_MF
#undef AHP_PIN4
#undef AHP_PIN2
#undef AHP_GROUP_SPLIT4
#undef AHP_GROUP_PLAIN
#undef AHP_RDK
#undef AHP_RDV

void attention_l2_pipe_launch(dim3 grid, hipStream_t st, const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt,
                              float *O, int ldo, int Tq, int Tk, float scale_log2e, unsigned char *O_lt, int o_row0, int o_k8,
                              AttnL3Batch ab) {
    hipLaunchKernelGGL(attention_l2_pipe_kernel<8>, grid, dim3(512), 0, st, Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, scale_log2e, O_lt,
                       o_row0, o_k8, ab);
}

}  // namespace sculpt
