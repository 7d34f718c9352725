// The attention tile loop software-pipelined inside the wave (pre-scaled queries: the product path of the TripoSR backbone and
// the image tokenizer).  Same tiles, staging and numerics as attention.hip's attention_kernel<NQB, true>, whose header
// describes the decomposition; this file is compiled with -fno-slp-vectorize (packed fp32 VALU ops issue for ~10 cycles and do
// not slide in behind an MFMA: tools/micro/mfma_fill.hip).
#include <stdlib.h>

#include "attention_tile.h"

namespace sculpt {

// One 16-byte-per-lane global -> LDS DMA as inline asm.  Through the builtin, hipcc knows the instruction writes LDS and, having
// no alias information, makes the LDS reads that follow wait for it (s_waitcnt vmcnt(2..0) a few MFMAs after the issue): the tile
// that was requested a whole iteration ahead is then awaited a quarter of an iteration after its request.  The loop's own
// vmcnt(0) + barrier at the top of the next iteration is the only synchronisation these loads need.  (Loads the compiler does not
// count only make its own vmcnt waits stricter: the counter completes in order.)
typedef int ai32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_lds(ai32x4 rsrc, unsigned lds_byte, unsigned lane_off, unsigned wave_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte), "v"(lane_off), "s"(rsrc), "s"(wave_off) : "memory");  // (m0 cannot be declared clobbered -- reserved; nothing else in this kernel uses it)
}
__device__ __forceinline__ ai32x4 raw_rsrc(const void *base) {
    const unsigned long a = (unsigned long)base;
    return ai32x4{(int)(a & 0xffffffffu), (int)(a >> 32), 0x7fffffff, 0x00020000};
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention with the tile loop software-pipelined INSIDE the wave (pre-scaled queries only; the product path).
// attention_kernel runs, per 64-key tile, 10 MFMAs (scores), then ~100 vector instructions (softmax: 32 quarter-rate v_exp),
// then 8 MFMAs (P.V) -- and on a SIMD the matrix phase of one wave does not overlap the vector phase of another
// (tools/micro/mfma_phase.hip), so the loop costs the SUM.  What does overlap is a wave's own vector instructions issued right
// behind its own MFMAs (tools/micro/mfma_fill.hip: ~24 issue cycles are free behind each).  So iteration tp runs
//   part A   S(tp+1) = -M + K(tp+1).Q^T   (2 + 8 MFMAs)   with   exp2 / row sum / bf16 conversion of the first half of S(tp)
//   part B   O += V(tp)^T.P(tp)           (8 MFMAs)       with   the second half of S(tp), then the row maximum of S(tp+1)
// one MFMA, then two exponentials + two adds (or four conversions / eight max3), fenced by sched_barrier so that hipcc keeps
// the order; the rare rescale (a query's maximum grew by more than 2^PRE_THR) is decided at the END of an iteration, before
// the next scores are requested with the updated maximum.  K(tp+1) and V(tp) live in different ring slots: K is staged one
// tile pair ahead of V.  Edge tiles (the first one, a ragged or missing last one) take the plain pieces.
// ---------------------------------------------------------------------------------------------------------------------
template <int NQB>
__global__ __launch_bounds__(NQB * 128) void attention_pipe_kernel(const uint16_t *__restrict__ Q, int ldq,
                                                                   const uint16_t *__restrict__ K, int ldk,
                                                                   const uint16_t *__restrict__ Vt, int ldvt,
                                                                   uint16_t *__restrict__ O, int ldo, int Tq, int Tk,
                                                                   const AttnBatch ab) {
    constexpr int NW = 2 * NQB;
    constexpr int SMEM = (NQB * 34 * 64 * 4 > 2 * 4 * 8192) ? NQB * 34 * 64 * 4 : 2 * 4 * 8192;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = wave % NQB, kh = wave / NQB;
    const int qc = lane & 31, h = lane >> 5;
    const int tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = tile / gridDim.x, bi = bh / ab.heads;  // scalar: (batch entry, head)
    const int head = bh - bi * ab.heads;
    Q += bi * ab.q_bs; K += bi * ab.k_bs; Vt += bi * ab.vt_bs; O += bi * ab.o_bs;
    const int q = (tile % gridDim.x) * (NQB * 32) + qi * 32 + qc;
    const int qld = min(q, Tq - 1);
    abf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const abf16x8 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);

    const int ntp = (Tk + 127) / 128;
    const int srow = lane >> 3, sslot = lane & 7;
    const int cs0 = ((sslot ^ (srow >> 1)) << 3), cs1 = cs0 ^ 32;
    const unsigned klane0 = (unsigned)(srow * ldk + cs0) * 2u, klane1 = (unsigned)(srow * ldk + cs1) * 2u;
    const unsigned vlane0 = (unsigned)(srow * ldvt + cs0) * 2u, vlane1 = (unsigned)(srow * ldvt + cs1) * 2u;
    const char *Kh = reinterpret_cast<const char *>(K + head * 64);
    const char *Vh = reinterpret_cast<const char *>(Vt + (long)head * 64 * ldvt);
    const ai32x4 k_rs = raw_rsrc(Kh), v_rs = raw_rsrc(Vh);
    // K of tile pair kt into ring slot kbuf, V of tile pair vt into ring slot vbuf (either skipped when past the last pair)
    auto stage2 = [&](int kbuf, int kt, int vbuf, int vt) {
#pragma unroll
        for (int i = 0; i < (32 + NW - 1) / NW; ++i) {
            const int j = wave + NW * i;  // scalar
            if (j < 32) {
                const int sub = j >> 3, rg = j & 7, half = sub & 1;
                if (sub < 2) {
                    if (kt < ntp) {
                        unsigned char *dst = smem + kbuf * 32768 + sub * 8192 + rg * 1024;
                        const int row0 = kt * 128 + half * 64 + 8 * rg;
                        if (kt * 128 + 128 <= Tk) {
                            dma16_lds(k_rs, (unsigned)(unsigned long)(alds_ptr_t)dst, (rg & 1) ? klane1 : klane0, row0 * ldk * 2);
                        } else {
                            const int key = min(row0 + srow, Tk - 1);
                            dma16_lds(k_rs, (unsigned)(unsigned long)(alds_ptr_t)dst, (unsigned)(key * ldk + ((rg & 1) ? cs1 : cs0)) * 2u, 0);
                        }
                    }
                } else if (vt < ntp) {
                    unsigned char *dst = smem + vbuf * 32768 + sub * 8192 + rg * 1024;
                    const bool last = (vt * 128 + 128 > Tk);
                    const int col = last ? min(vt * 128 + half * 64, ab.vt_cols - 64) : vt * 128 + half * 64;  // (columns this ENTRY may read)
                    dma16_lds(v_rs, (unsigned)(unsigned long)(alds_ptr_t)dst, (rg & 1) ? vlane1 : vlane0, ((8 * rg) * ldvt + col) * 2);
                }
            }
        }
    };

    // The same staging for tile pairs that are complete (all but a ragged last one), with everything a unit needs worked out
    // once: per iteration and unit one scalar multiply-add, m0, the DMA -- stage2's per-unit case analysis is ~55 scalar
    // instructions per wave and iteration, 8 % of the loop (tools/micro/attn_ablate.sh).
    constexpr int NU = (32 + NW - 1) / NW;
    ai32x4 u_rs[NU];
    unsigned u_lane[NU], u_lds[NU][2];
    int u_off[NU], u_step[NU];
    bool u_on[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int j = wave + NW * i, sub = (j >> 3) & 3, rg = j & 7, half = sub & 1;
        const bool is_k = sub < 2;
        u_on[i] = j < 32;
        u_rs[i] = is_k ? k_rs : v_rs;
        u_lane[i] = is_k ? ((rg & 1) ? klane1 : klane0) : ((rg & 1) ? vlane1 : vlane0);
        // K unit: rows kt * 128 + half * 64 + 8 rg, kt = tp + 2;  V unit: rows 8 rg, columns vt * 128 + half * 64, vt = tp + 1
        u_step[i] = is_k ? 128 * ldk * 2 : 256;
        u_off[i] = is_k ? ((half * 64 + 8 * rg) * ldk * 2 + 2 * u_step[i]) : (((8 * rg) * ldvt + half * 64) * 2 + u_step[i]);
        const unsigned dst = (unsigned)(unsigned long)(alds_ptr_t)(smem + sub * 8192 + rg * 1024);
        u_lds[i][0] = dst + (is_k ? 0 : 32768);  // iteration parity 0: K into ring slot 0, V into slot 1
        u_lds[i][1] = dst + (is_k ? 32768 : 0);
    }
    auto stage_full = [&](int par, int tp) {
#pragma unroll
        for (int i = 0; i < NU; ++i)
            if (u_on[i]) dma16_lds(u_rs[i], u_lds[i][par], u_lane[i], u_off[i] + tp * u_step[i]);
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = 0.f, l_run = 0.f;  // log2 units; m_run == M_hi + M_lo exactly (two bf16 parts)
    typedef short abf16x4s __attribute__((ext_vector_type(4)));
    // MFMA row m of a score tile is key pk(m) = m with bits 2 and 3 swapped: lane (query, h) then holds, in accumulator
    // registers 8 g .. 8 g + 7, the CONTIGUOUS keys 16 g + 8 h .. + 7 -- exactly the k-slots the P.V step wants from it, so
    // the V^T fragment is one conflict-free 16-byte read like K's (attention_kernel's natural order needs two 8-byte pieces
    // per fragment, 2-way bank-conflicted: half of that kernel's LDS cycles).  Only the K row a lane reads changes.
    const int pk = (qc & 0x13) | ((qc & 4) << 1) | ((qc & 8) >> 1);
    const int kbase = a_lds_off(pk, h);
    const int vbase = a_lds_off(qc, h);

    // the constant [1, 1, 0, ..] row fragment and the (-M_hi, -M_lo, 0, ..) column fragment of the maximum-subtracting k-step;
    // the latter follows m_run (set_mq: the wave's first tile and the rare rescale only)
    const unsigned w_ones = h == 0 ? 0x3f803f80u : 0u;
    const abf16x4s ones = abf16x4s{(short)(w_ones & 0xffffu), (short)(w_ones >> 16), 0, 0};
    abf16x4s mq = abf16x4s{0, 0, 0, 0};
    auto set_mq = [&]() {
        const __bf16 hi = (__bf16)m_run;
        const __bf16 lo = (__bf16)(m_run - (float)hi);
        const unsigned neg = ((unsigned)(unsigned short)__builtin_bit_cast(short, hi) |
                              ((unsigned)(unsigned short)__builtin_bit_cast(short, lo) << 16)) ^ 0x80008000u;
        const unsigned w_mq = h == 0 ? neg : 0u;
        mq = abf16x4s{(short)(w_mq & 0xffffu), (short)(w_mq >> 16), 0, 0};
    };
    auto ldkf = [&](const unsigned char *Kt, int ks, int upper) -> abf16x8 {
        return *reinterpret_cast<const abf16x8 *>(Kt + (kbase ^ (ks << 5)) + upper * 4096);
    };
    // V^T fragment of k-step kk, rows qc (upper = 0) / 32 + qc: keys 16 kk + 8 h .. + 7
    auto ldv = [&](const unsigned char *Vtl, int kk, int upper) -> abf16x8 {
        return *reinterpret_cast<const abf16x8 *>(Vtl + (vbase ^ (kk << 5)) + upper * 4096);
    };
    auto pfrag = [&](const f32x16 &s, int half) -> abf16x8 {  // eight probabilities -> the B operand of one P.V k-step
        abf16x8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (__bf16)s[8 * half + j];
        return pb;
    };

    // ---- plain pieces (edge tiles)
    auto qk_plain = [&](int kslot, f32x16 &s0, f32x16 &s1) {
        const unsigned char *Kt = smem + kslot * 32768 + kh * 8192;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        s0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, s1, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldkf(Kt, ks, 0), qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldkf(Kt, ks, 1), qf[ks], s1, 0, 0, 0);
        }
    };
    auto mask_ragged = [&](int tp, f32x16 &s0, f32x16 &s1) {
        const int key_start = tp * 128 + kh * 64;
        if (key_start + 64 > Tk) {
            const int kb = key_start + 8 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key0 = kb + (r & 7) + 16 * (r >> 3);
                if (key0 >= Tk) s0[r] = -INFINITY;
                if (key0 + 32 >= Tk) s1[r] = -INFINITY;
            }
        }
    };
    // The rescale a complete score tile may force (mx: the LANE's maximum over its 32 keys, relative to M): M moves only on the
    // wave's first tile or when some query's maximum exceeds it by more than 2^PRE_THR; then S, O and l follow M.  The test is
    // per lane + one ballot: the two lanes of a query exchange their maxima only on the rare path.
    auto rescale_if = [&](bool first, float mx_lane, f32x16 &s0, f32x16 &s1) {
        if (__builtin_amdgcn_ballot_w64(first || mx_lane > PRE_THR) != 0) {  // wave-uniform
            const float mx = fmaxf(mx_lane, __shfl_xor(mx_lane, 32, 64));
            const bool need = first || mx > PRE_THR;
            float m_new = m_run;
            if (need && mx > -INFINITY) {
                const float t = m_run + mx;
                const __bf16 hi = (__bf16)t;
                const __bf16 lo = (__bf16)(t - (float)hi);
                m_new = (float)hi + (float)lo;
            }
            const float delta = m_new - m_run;
            m_run = m_new;
            set_mq();
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
    };
    auto lane_max = [&](const f32x16 &s0, const f32x16 &s1) -> float {
        float mx = max3f(s0[0], s1[0], s0[1]);
#pragma unroll
        for (int r = 1; r < 15; ++r) mx = max3f(mx, s1[r], s0[r + 1]);
        return fmaxf(mx, s1[15]);
    };
    // ---- the pipelined iteration: c = S(tp) (complete, maximum settled), K(tp+1) in ring slot kslot, V(tp) in slot vslot;
    // on return n = S(tp+1) relative to the M of this iteration, its row maximum in `mx_next`
#define AT_FENCE __builtin_amdgcn_sched_barrier(0)
#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0
#endif
#if ATTN_ABLATE == 6
#define AT_LD(x) do { } while (0)
#else
#define AT_LD(x) do { x; } while (0)
#endif
#if ATTN_ABLATE == 1
#define AT_E2(v, r) do { } while (0)
#else
#define AT_E2(v, r) do { (v)[r] = __builtin_amdgcn_exp2f((v)[r]); (v)[(r) + 1] = __builtin_amdgcn_exp2f((v)[(r) + 1]); } while (0)
#endif
    // two probabilities -> one packed bf16 pair of the P fragment
#if ATTN_ABLATE == 2
#define AT_C2(pf, j, v, r) do { } while (0)
#else
#define AT_C2(pf, j, v, r) do { (pf)[j] = (__bf16)(v)[r]; (pf)[(j) + 1] = (__bf16)(v)[(r) + 1]; } while (0)
#endif
    // row sums of the fp32 probabilities in FOUR chains: a dependent VALU instruction waits out its predecessor's latency, and the
    // loop is short of issue slots (two chains: +1.7 us on the 3072 x 3072 x 16 launch; v_dot2c_f32_bf16 on the packed pairs,
    // half the instructions: +3.4 us in two chains, +1.7 in four -- it does not issue like a plain VALU op behind an MFMA).
    // The empty asm keeps the sums inside their slot: pure arithmetic otherwise sinks past the fences to its only use.
#define AT_D4(v, r)                                                                                                          \
    do {                                                                                                                     \
        if (ATTN_ABLATE != 3) { pa += (v)[r]; pb_ += (v)[(r) + 1]; pc += (v)[(r) + 2]; pd += (v)[(r) + 3]; }                  \
        asm volatile("" : "+v"(pa), "+v"(pb_), "+v"(pc), "+v"(pd));                                                          \
    } while (0)
    auto pipe_iter = [&](int kslot, int vslot, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1, float &mx_next) {
        const unsigned char *Kt = smem + kslot * 32768 + kh * 8192;
        const unsigned char *Vtl = smem + vslot * 32768 + (2 + kh) * 8192;
        // a fragment is requested two MFMA slots before its use, into the registers of the one just consumed (four slots ahead,
        // from a pool of four, measured the same: SQ_WAIT_INST_LDS is 2 % of the wave cycles)
        abf16x8 f0 = ldkf(Kt, 0, 0), f1 = ldkf(Kt, 0, 1);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float pa = 0.f, pb_ = 0.f, pc = 0.f, pd = 0.f;
        abf16x8 p0 = qf[0], p1 = qf[1];
        AT_FENCE;
        // ---- part A: 2 + 8 score MFMAs of tile tp + 1; exponentials of S(tp), P fragments, row sums
        n0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, zero, 0, 0, 0);
        AT_E2(c0, 0);
        AT_FENCE;
        n1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, zero, 0, 0, 0);
        AT_E2(c0, 2);
        AT_FENCE;
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qf[0], n0, 0, 0, 0);
        AT_LD(f0 = ldkf(Kt, 1, 0));
        AT_E2(c0, 4); AT_E2(c0, 6);
        AT_FENCE;
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qf[0], n1, 0, 0, 0);
        AT_LD(f1 = ldkf(Kt, 1, 1));
        AT_E2(c0, 8); AT_E2(c0, 10);
        AT_FENCE;
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qf[1], n0, 0, 0, 0);
        AT_LD(f0 = ldkf(Kt, 2, 0));
        AT_E2(c0, 12); AT_C2(p0, 0, c0, 0); AT_C2(p0, 2, c0, 2);
        AT_FENCE;
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qf[1], n1, 0, 0, 0);
        AT_LD(f1 = ldkf(Kt, 2, 1));
        AT_E2(c0, 14); AT_C2(p0, 4, c0, 4); AT_C2(p0, 6, c0, 6);
        AT_FENCE;
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qf[2], n0, 0, 0, 0);
        AT_LD(f0 = ldkf(Kt, 3, 0));
        AT_E2(c1, 0); AT_D4(c0, 0);
        AT_FENCE;
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qf[2], n1, 0, 0, 0);
        AT_LD(f1 = ldkf(Kt, 3, 1));
        AT_E2(c1, 2); AT_D4(c0, 4);
        AT_FENCE;
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qf[3], n0, 0, 0, 0);
        AT_LD(f0 = ldv(Vtl, 0, 0));
        AT_E2(c1, 4); AT_C2(p1, 0, c0, 8); AT_C2(p1, 2, c0, 10);
        AT_FENCE;
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qf[3], n1, 0, 0, 0);
        AT_LD(f1 = ldv(Vtl, 0, 1));
        AT_E2(c1, 6); AT_C2(p1, 4, c0, 12); AT_C2(p1, 6, c0, 14);
        AT_FENCE;
        // ---- part B: 8 P.V MFMAs of tile tp; the rest of S(tp); then the lane maximum of S(tp + 1)
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, p0, o0, 0, 0, 0);
        AT_LD(f0 = ldv(Vtl, 1, 0));
        AT_E2(c1, 8); AT_D4(c0, 8);
        AT_FENCE;
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, p0, o1, 0, 0, 0);
        AT_LD(f1 = ldv(Vtl, 1, 1));
        AT_E2(c1, 10); AT_D4(c0, 12);
        AT_FENCE;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, p1, o0, 0, 0, 0);
        AT_LD(f0 = ldv(Vtl, 2, 0));
        AT_E2(c1, 12); AT_C2(p0, 0, c1, 0); AT_C2(p0, 2, c1, 2);
        AT_FENCE;
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, p1, o1, 0, 0, 0);
        AT_LD(f1 = ldv(Vtl, 2, 1));
        AT_E2(c1, 14); AT_C2(p0, 4, c1, 4); AT_C2(p0, 6, c1, 6);
        AT_FENCE;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, p0, o0, 0, 0, 0);
        AT_LD(f0 = ldv(Vtl, 3, 0));
        AT_C2(p1, 0, c1, 8); AT_C2(p1, 2, c1, 10); AT_C2(p1, 4, c1, 12); AT_C2(p1, 6, c1, 14);
        AT_D4(c1, 0);
        AT_FENCE;
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, p0, o1, 0, 0, 0);
        AT_LD(f1 = ldv(Vtl, 3, 1));
        AT_D4(c1, 4); AT_D4(c1, 8); AT_D4(c1, 12);
        AT_FENCE;
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, p1, o0, 0, 0, 0);
        // the lane maximum in four independent chains (a dependent VALU instruction waits out the previous one's latency)
        float mxa = max3f(n0[0], n1[0], n0[1]), mxb = max3f(n0[2], n1[2], n0[3]);
        float mxc = max3f(n0[4], n1[4], n0[5]), mxd = max3f(n0[6], n1[6], n0[7]);
        mxa = max3f(mxa, n1[1], n0[8]); mxb = max3f(mxb, n1[3], n0[9]);
        mxc = max3f(mxc, n1[5], n0[10]); mxd = max3f(mxd, n1[7], n0[11]);
        AT_FENCE;
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, p1, o1, 0, 0, 0);
        mxa = max3f(mxa, n1[8], n0[12]); mxb = max3f(mxb, n1[9], n0[13]);
        mxc = max3f(mxc, n1[10], n0[14]); mxd = max3f(mxd, n1[11], n0[15]);
        mxa = max3f(mxa, n1[12], n1[13]); mxb = max3f(mxb, n1[14], n1[15]);
        mxa = max3f(mxa, mxb, mxc);
        mx_next = fmaxf(mxa, mxd);
        AT_FENCE;
        l_run += (pa + pb_) + (pc + pd);
    };
#undef AT_E2
#undef AT_C2
#undef AT_D4

    // ---- main loop.  wave_has(tp): the wave's 64 keys of tile pair tp exist; wave_full(tp): all 64 do
    auto wave_has = [&](int tp) { return tp < ntp && tp * 128 + kh * 64 < Tk; };
    auto wave_full = [&](int tp) { return tp < ntp && tp * 128 + kh * 64 + 64 <= Tk; };
    f32x16 sa0, sa1, sb0, sb1;
    // one iteration: c = S(tp) ready (maximum settled); leaves S(tp+1) in n, ready the same way
    auto body = [&](auto par_t, int tp, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1) {
        constexpr int par = decltype(par_t)::value;  // tp & 1: ring slots are compile-time
        // Unconditional, tp == 0 included: the refill below overwrites ring slot `par`, and at tp == 0 that slot holds K(0), which
        // slower waves may still be reading in the pre-loop qk_plain(0) -- only a barrier AFTER that read orders their ds_reads
        // against this wave's DMA (one extra barrier per launch).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // K(tp) [slot par] and V(tp-1) [slot par ^ 1] are dead for every wave: refill them
        if ((tp + 3) * 128 <= Tk) stage_full(par, tp);
        else stage2(par, tp + 2, par ^ 1, tp + 1);
        // ONE call site and no else branch: hipcc hoists code common to two branches above the branch (all 32 exponentials,
        // which is the phase-separated loop again).  A missing tile pair tp + 1 is computed from whatever its ring slot holds
        // and dropped; a ragged one is masked afterwards and its maximum redone.
        if (wave_has(tp)) {
            float mx;
            pipe_iter(par ^ 1, par, c0, c1, n0, n1, mx);
            if (wave_has(tp + 1)) {
                if (!wave_full(tp + 1)) {
                    mask_ragged(tp + 1, n0, n1);
                    mx = lane_max(n0, n1);
                }
                rescale_if(false, mx, n0, n1);
            }
        }
    };
    stage2(0, 0, 0, 0);
    stage2(1, 1, 0, ntp);
    // a use of the Q fragments HERE: otherwise hipcc carries their pending-load state into the loop (a wave without a first
    // tile skips their first use) and its vmcnt(3..0) before the loop's first MFMAs would wait on the tile DMAs just issued
    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave_has(0)) {
        qk_plain(0, sa0, sa1);
        mask_ragged(0, sa0, sa1);
        rescale_if(true, lane_max(sa0, sa1), sa0, sa1);
    }
    for (int tp = 0; tp < ntp; tp += 2) {
        body(std::integral_constant<int, 0>{}, tp, sa0, sa1, sb0, sb1);
        if (tp + 1 < ntp) body(std::integral_constant<int, 1>{}, tp + 1, sb0, sb1, sa0, sa1);
    }
#undef AT_FENCE

    // ---- merge the two key halves of each query block through LDS (as in attention_kernel, PRE form)
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    __syncthreads();
    float *scr = reinterpret_cast<float *>(smem) + qi * 34 * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { scr[r * 64] = o0[r]; scr[(16 + r) * 64] = o1[r]; }
        scr[32 * 64] = !(l_tot > 0.f) ? -INFINITY : m_run;
        scr[33 * 64] = l_tot;
    }
    __syncthreads();
    if (kh == 0) {
        const float m1 = scr[32 * 64], l1 = scr[33 * 64];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m);
        const float a1 = (l1 > 0.f) ? __builtin_amdgcn_exp2f(m1 - m) : 0.f;
        const float inv = 1.0f / (a0 * l_tot + a1 * l1);
        if (q < Tq) {
            uint16_t *orow = O + (long)q * ldo + head * 64;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
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

void attention_pipe_launch(int nqb, dim3 grid, dim3 block, hipStream_t st, const uint16_t *Q, int ldq, const uint16_t *K, int ldk,
                           const uint16_t *Vt, int ldvt, uint16_t *O, int ldo, int Tq, int Tk, AttnBatch ab) {
    if (nqb == 6) hipLaunchKernelGGL((attention_pipe_kernel<6>), grid, block, 0, st, Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, ab);
    else hipLaunchKernelGGL((attention_pipe_kernel<4>), grid, block, 0, st, Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, ab);
}

}  // namespace sculpt
