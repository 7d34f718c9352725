// Flash-style fused attention (no mask, head dim 64, bf16 in/out, fp32 softmax) for gfx950.
//
// Replaces F.scaled_dot_product_attention at
//   TripoSR/tsr/models/transformer/attention.py:629-631   (self 3072x3072, cross 3072x1025; 16 heads)
// and the eager softmax(QK^T/8)V of HF ViTSelfAttention (12 heads, 1025 tokens).
//
// One workgroup = 8 waves = 128 queries of one head; a wave owns 32 queries and one half of every
// 128-key tile pair (partial softmax states merged at the end).  Per 64-key tile:
//   S^T = K . Q^T   on v_mfma_f32_32x32x16_bf16 with the QUERY on the lane (column) -- so a lane
//         holds 2x16 scores of ONE query: row max / row sum are in-register reductions plus one
//         cross-half exchange, no LDS round trip.
//   O^T += V^T . P^T: the S^T accumulator registers, converted pairwise to bf16, are directly the
//         B operand of the second product (k order permuted to key = 16s + 8(j>>2) + 4h + (j&3));
//         the A operand V^T is read from LDS in that same key order.  V arrives already transposed
//         ([head*64+d][key], written by the QKV GEMM epilogue), so no transposing read is needed.
// K and V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4) one tile ahead into double-buffered
// LDS (one barrier per tile); rows use the (row>>1)&7 chunk swizzle (applied to the source address
// and to the read address) so ds_read_b128 is conflict-free.
#include "common.h"

namespace sculpt {

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *alds_ptr_t;
typedef const __attribute__((address_space(1))) void *agbl_ptr_t;

__device__ __forceinline__ int a_lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// One workgroup = 8 waves = 128 queries of one head.  Wave w: query block qi = w & 3 (32 queries),
// key half kh = w >> 2.  Each iteration stages a PAIR of 64-key tiles; the kh = 0 waves consume the
// first, the kh = 1 waves the second (flash-decoding style split of the key range inside the
// workgroup): 2x the waves per SIMD of a 4-wave layout at these small shapes (1.5 workgroups per CU),
// which is what hides the LDS / LDS-DMA latency.  The two partial (max, sum, O) states of a query
// block are merged through LDS at the end.
__global__ __launch_bounds__(512, 4) void attention_kernel(const uint16_t *__restrict__ Q, int ldq,
                                                        const uint16_t *__restrict__ K, int ldk,
                                                        const uint16_t *__restrict__ Vt, int ldvt,
                                                        uint16_t *__restrict__ O, int ldo, int Tq, int Tk,
                                                        float scale_log2e) {
    // [stage][K0 | K1 | V0 | V1] sub-tiles of 8 KiB (64 rows x 128 B); reused as merge scratch at the end
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * 8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = wave & 3, kh = wave >> 2;
    const int qc = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int q = blockIdx.x * 128 + qi * 32 + qc;
    const int qld = min(q, Tq - 1);

    // Q fragments (B operand): Q[q][16*ks + 8h + j]
    abf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const abf16x8 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);

    // staging by LDS-DMA: wave w fills rows 32*(w&1) .. +31 of sub-tile (w>>1): 4 instructions of 8 rows.
    // The chunk XOR is applied to the per-lane source address (the LDS image is lane-linear).
    const int ntp = (Tk + 127) / 128;
    const int sub = wave >> 1;              // 0: K0, 1: K1, 2: V0, 3: V1
    const int srow = lane >> 3, sslot = lane & 7;
    const int rbase = 32 * (wave & 1) + srow;  // + 8*i
    const int sdst = sub * 8192 + 32 * (wave & 1) * 128;
    const bool is_k = sub < 2;
    const int half = sub & 1;

#define STAGE(buf, tp)                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
            const int r = rbase + 8 * i;                                                                       \
            const int cs = (sslot ^ ((r >> 1) & 7)) << 3;                                                      \
            const uint16_t *src;                                                                               \
            if (is_k) {                                                                                        \
                const int key = min((tp) * 128 + half * 64 + r, Tk - 1);                                       \
                src = K + (long)key * ldk + head * 64 + cs;                                                    \
            } else {                                                                                           \
                const int col = min((tp) * 128 + half * 64, ldvt - 64);                                        \
                src = Vt + (long)(head * 64 + r) * ldvt + col + cs;                                            \
            }                                                                                                  \
            __builtin_amdgcn_global_load_lds((agbl_ptr_t)src, (alds_ptr_t)(smem + (buf) * 32768 + sdst + i * 1024), 16, 0, 0); \
        }                                                                                                      \
    } while (0)

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;  // m_run in raw score units (before the softmax scale)

    STAGE(0, 0);
    for (int tp = 0; tp < ntp; ++tp) {
        const int buf = tp & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tp + 1 < ntp) STAGE(buf ^ 1, tp + 1);
        const int key_start = tp * 128 + kh * 64;
        if (key_start < Tk) {  // wave-uniform: a wholly out-of-range tile is skipped
            const unsigned char *Kt = smem + buf * 32768 + kh * 8192;
            const unsigned char *Vtl = smem + buf * 32768 + (2 + kh) * 8192;
            // ---- S^T = K . Q^T
            f32x16 s0, s1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const abf16x8 k0 = *reinterpret_cast<const abf16x8 *>(Kt + a_lds_off(qc, 2 * ks + h));
                const abf16x8 k1 = *reinterpret_cast<const abf16x8 *>(Kt + a_lds_off(32 + qc, 2 * ks + h));
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], s1, 0, 0, 0);
            }
            // ---- online softmax over this lane's 32 keys (+ the other half's 32)
            if (key_start + 64 > Tk) {  // ragged last tile only
                const int kb = key_start + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key0 = kb + (r & 3) + 8 * (r >> 2);
                    if (key0 >= Tk) s0[r] = -INFINITY;
                    if (key0 + 32 >= Tk) s1[r] = -INFINITY;
                }
            }
            float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            const float mc = -m_new * scale_log2e;
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], scale_log2e, mc));
                s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], scale_log2e, mc));
                psum += s0[r] + s1[r];
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            // ---- O^T += V^T . P^T   (4 k-steps of 16 keys)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                abf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = (kk < 2) ? s0[8 * (kk & 1) + j] : s1[8 * (kk & 1) + j];
                    pb[j] = (__bf16)pv;
                }
                // keys of element j: 16*kk + 8*(j>>2) + 4h + (j&3)  -> two 8-byte reads per d row
                const int c0 = 2 * kk, c1 = 2 * kk + 1;
                const abf16x4 a = *reinterpret_cast<const abf16x4 *>(Vtl + a_lds_off(qc, c0) + 8 * h);
                const abf16x4 b = *reinterpret_cast<const abf16x4 *>(Vtl + a_lds_off(qc, c1) + 8 * h);
                const abf16x8 v0 = abf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                const abf16x4 c = *reinterpret_cast<const abf16x4 *>(Vtl + a_lds_off(32 + qc, c0) + 8 * h);
                const abf16x4 d = *reinterpret_cast<const abf16x4 *>(Vtl + a_lds_off(32 + qc, c1) + 8 * h);
                const abf16x8 v1 = abf16x8{c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, o1, 0, 0, 0);
            }
        }
    }
#undef STAGE
    // ---- merge the two key halves of each query block through LDS: layout [qi][r (0..33)][lane]
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    __syncthreads();  // all tile reads done; staging memory becomes scratch
    float *scr = reinterpret_cast<float *>(smem) + qi * 34 * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { scr[r * 64] = o0[r]; scr[(16 + r) * 64] = o1[r]; }
        scr[32 * 64] = m_run;
        scr[33 * 64] = l_tot;
    }
    __syncthreads();
    if (kh == 0) {
        const float m1 = scr[32 * 64], l1 = scr[33 * 64];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run - m) * scale_log2e);
        const float a1 = (l1 > 0.f) ? __builtin_amdgcn_exp2f((m1 - m) * scale_log2e) : 0.f;
        const float inv = 1.0f / (a0 * l_tot + a1 * l1);
        if (q < Tq) {
            uint16_t *orow = O + (long)q * ldo + head * 64;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // registers 4*g4 .. 4*g4+3 are d = 8*g4 + 4h + {0..3}
                abf16x4 x, y;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x[r] = (__bf16)((a0 * o0[4 * g4 + r] + a1 * scr[(4 * g4 + r) * 64]) * inv);
                    y[r] = (__bf16)((a0 * o1[4 * g4 + r] + a1 * scr[(16 + 4 * g4 + r) * 64]) * inv);
                }
                *reinterpret_cast<abf16x4 *>(orow + 8 * g4 + 4 * h) = x;
                *reinterpret_cast<abf16x4 *>(orow + 32 + 8 * g4 + 4 * h) = y;
            }
        }
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" int sculpt_attention_bf16(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt,
                                     int ldvt, uint16_t *O, int ldo, int Tq, int Tk, int heads, float scale,
                                     sculpt_stream_t stream) {
    SC_REQUIRE(Q && K && Vt && O, "attention: null argument");
    SC_REQUIRE(Tq >= 1 && Tk >= 1 && heads >= 1, "attention: bad shape Tq=%d Tk=%d heads=%d", Tq, Tk, heads);
    SC_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attention: row strides must keep 16-byte alignment");
    SC_REQUIRE(ldvt >= ((Tk + 63) / 64) * 64, "attention: ldvt=%d must be >= round_up(Tk=%d, 64)", ldvt, Tk);
    hipLaunchKernelGGL(attention_kernel, dim3(cdiv(Tq, 128), heads), dim3(512), 0, as_stream(stream), Q, ldq, K, ldk,
                       Vt, ldvt, O, ldo, Tq, Tk, scale * 1.44269504088896340736f);
    SC_LAUNCH_CHECK();
    return 0;
}
