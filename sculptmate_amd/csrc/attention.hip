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
#include <stdlib.h>

#include "common.h"
#include "attention_tile.h"

namespace sculpt {

// One workgroup = 2*NQB waves = NQB*32 queries of one head.  Wave w: query block qi = w % NQB (32 queries),
// key half kh = w / NQB.  Each iteration stages a PAIR of 64-key tiles; the kh = 0 waves consume the first, the
// kh = 1 waves the second (flash-decoding style split of the key range inside the workgroup): twice the waves per
// SIMD of a one-wave-per-query-block layout at these small shapes.  The two partial (max, sum, O) states of a
// query block are merged through LDS at the end.  NQB = 4 (128 queries) is what the launcher uses: 96-query
// workgroups would fill the 2 x 256 workgroup slots evenly at 3072 queries x 16 heads (512 blocks instead of 384)
// but measured 20 % SLOWER -- every workgroup re-stages the head's whole K/V, and that L2 -> LDS traffic (already
// ~10 TB/s at 128 queries) grows with the block count.
//
// Register budget is 128 VGPRs (4 waves per SIMD): everything address-like that is wave-uniform lives in SGPRs
// (the staging source is a scalar base plus one of four per-lane 32-bit offsets), because a single spill puts a
// scratch load -- and with it an s_waitcnt vmcnt(0) that also waits for the LDS-DMA prefetch -- into the loop.
typedef float af32x2 __attribute__((ext_vector_type(2)));

template <int NQB, bool PRE = false>
__global__ __launch_bounds__(NQB * 128, 4) void attention_kernel(const uint16_t *__restrict__ Q, int ldq,
                                                                  const uint16_t *__restrict__ K, int ldk,
                                                                  const uint16_t *__restrict__ Vt, int ldvt,
                                                                  uint16_t *__restrict__ O, int ldo, int Tq, int Tk,
                                                                  float scale_log2e, const AttnBatch ab) {
    constexpr int NW = 2 * NQB;  // waves
    // [stage][K0 | K1 | V0 | V1] sub-tiles of 8 KiB (64 rows x 128 B); reused as merge scratch at the end
    constexpr int SMEM = (NQB * 34 * 64 * 4 > 2 * 4 * 8192) ? NQB * 34 * 64 * 4 : 2 * 4 * 8192;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = wave % NQB, kh = wave / NQB;
    const int qc = lane & 31, h = lane >> 5;
    // each XCD works on whole heads (2 of 16 here): their K/V (0.8 MB per head) stay in that XCD's 4 MB L2 while
    // the head's 24 query blocks re-stage them
    const int tile = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = tile / gridDim.x, bi = bh / ab.heads;  // scalar: (batch entry, head)
    const int head = bh - bi * ab.heads;
    Q += bi * ab.q_bs; K += bi * ab.k_bs; Vt += bi * ab.vt_bs; O += bi * ab.o_bs;
    const int q = (tile % gridDim.x) * (NQB * 32) + qi * 32 + qc;
    const int qld = min(q, Tq - 1);

    // Q fragments (B operand): Q[q][16*ks + 8h + j]
    abf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const abf16x8 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);

    // ---- staging by LDS-DMA.  A tile pair is 32 wave-instructions of 8 rows (1 KiB): instruction j fills rows
    // 8*(j&7).. of sub-tile j>>3 (0: K0, 1: K1, 2: V0, 3: V1); wave w issues j = w, w+NW, ...  The LDS image is
    // lane-linear, so the (row>>1)&7 chunk swizzle is applied to the per-lane SOURCE offset:
    //   row = 8*rg + srow  ->  swizzle = (srow>>1) ^ ((rg&1)<<2): two per-lane variants (rg even / odd).
    const int ntp = (Tk + 127) / 128;
    const int srow = lane >> 3, sslot = lane & 7;
    const int cs0 = ((sslot ^ (srow >> 1)) << 3), cs1 = cs0 ^ 32;  // element offsets of the lane's chunk
    const unsigned klane0 = (unsigned)(srow * ldk + cs0) * 2u, klane1 = (unsigned)(srow * ldk + cs1) * 2u;
    const unsigned vlane0 = (unsigned)(srow * ldvt + cs0) * 2u, vlane1 = (unsigned)(srow * ldvt + cs1) * 2u;
    const char *Kh = reinterpret_cast<const char *>(K + head * 64);
    const char *Vh = reinterpret_cast<const char *>(Vt + (long)head * 64 * ldvt);

    // LDS-DMA through buffer addressing (buffer_load_dwordx4 ... offen lds): descriptor in SGPRs, a 32-bit per-lane offset
    // and a scalar offset per instruction -- no 64-bit per-lane address exists, so nothing address-like is hoisted into
    // VGPR pairs around the tile loop (global_load_lds with base + offset cost 2-8 VGPRs that way and spilled one pair)
    const auto k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(Kh), 0, 0x7fffffff, 0x00020000);
    const auto v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(Vh), 0, 0x7fffffff, 0x00020000);
    // K and V of a tile pair are staged separately: kt / vt = tile pair (skipped when past the last one), kbuf / vbuf = ring slot
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
                        if (kt * 128 + 128 <= Tk) {  // wave-uniform: only the final pair can run past the arrays
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (alds_ptr_t)dst, 16, (rg & 1) ? klane1 : klane0,
                                                                     row0 * ldk * 2, 0, 0);
                        } else {
                            const int key = min(row0 + srow, Tk - 1);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rs, (alds_ptr_t)dst, 16,
                                                                     (unsigned)(key * ldk + ((rg & 1) ? cs1 : cs0)) * 2u, 0, 0, 0);
                        }
                    }
                } else if (vt < ntp) {
                    unsigned char *dst = smem + vbuf * 32768 + sub * 8192 + rg * 1024;
                    const bool last = (vt * 128 + 128 > Tk);
                    const int col = last ? min(vt * 128 + half * 64, ab.vt_cols - 64) : vt * 128 + half * 64;  // (columns this ENTRY may read)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rs, (alds_ptr_t)dst, 16, (rg & 1) ? vlane1 : vlane0,
                                                             ((8 * rg) * ldvt + col) * 2, 0, 0);
                }
            }
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = PRE ? 0.f : -INFINITY, l_run = 0.f;  // raw score units (before the softmax scale); PRE: log2 units, = M_hi + M_lo
    // PRE: the extra k-step runs on v_mfma_f32_32x32x8_bf16_1k (4 bf16 = 2 VGPRs per operand): A[key][k] = 1 for k = 0, 1;
    // B[k][query] = -M_hi, -M_lo; lane l holds k = 4 (l >> 5) + j.  Both fragments are rebuilt per tile from m_run (6 VALU):
    // the loop sits exactly at the 128-VGPR budget and four more live registers spill a pointer into it.
    typedef short abf16x4s __attribute__((ext_vector_type(4)));

    // fragment read offsets: K rows qc / 32+qc, chunk 2*ks + h  ->  base ^ (ks << 5)
    const int kbase = a_lds_off(qc, h);
    const int vbase = a_lds_off(qc, 0) + 8 * h;  // V^T rows qc / 32+qc, chunk c -> base ^ (c << 4)

    // ---- S^T = K . Q^T of the wave's 64 keys of tile pair tp (K in ring slot kbuf)
    auto qk = [&](int kbuf, f32x16 &s0, f32x16 &s1) {
        const unsigned char *Kt = smem + kbuf * 32768 + kh * 8192;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        if (PRE) {  // s = -M for every key of this query (column), first: the K fragments are not live yet; m_run == hi + lo exactly
            const __bf16 hi = (__bf16)m_run;
            const __bf16 lo = (__bf16)(m_run - (float)hi);
            const unsigned neg = ((unsigned)(unsigned short)__builtin_bit_cast(short, hi) |
                                  ((unsigned)(unsigned short)__builtin_bit_cast(short, lo) << 16)) ^ 0x80008000u;
            const unsigned w_ones = h == 0 ? 0x3f803f80u : 0u, w_mq = h == 0 ? neg : 0u;
            const abf16x4s ones = {(short)(w_ones & 0xffffu), (short)(w_ones >> 16), 0, 0};
            const abf16x4s mq = {(short)(w_mq & 0xffffu), (short)(w_mq >> 16), 0, 0};
            s0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ones, mq, s1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const abf16x8 k0 = *reinterpret_cast<const abf16x8 *>(Kt + (kbase ^ (ks << 5)));
            const abf16x8 k1 = *reinterpret_cast<const abf16x8 *>(Kt + (kbase ^ (ks << 5)) + 4096);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], s1, 0, 0, 0);
        }
    };

    // ---- online softmax over this lane's 32 keys (+ the other half's 32) of tile pair tp, then O^T += V^T . P^T (V in slot vbuf)
    auto softmax_pv = [&](int tp, int vbuf, f32x16 &s0, f32x16 &s1) {
        const int key_start = tp * 128 + kh * 64;
        const unsigned char *Vtl = smem + vbuf * 32768 + (2 + kh) * 8192;
        if (key_start + 64 > Tk) {  // ragged last tile only
            const int kb = key_start + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key0 = kb + (r & 3) + 8 * (r >> 2);
                if (key0 >= Tk) s0[r] = -INFINITY;
                if (key0 + 32 >= Tk) s1[r] = -INFINITY;
            }
        }
        float mx = max3f(s0[0], s1[0], s0[1]);
        mx = max3f(mx, s1[1], s0[2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3f(mx, s1[r], s0[r + 1]);
        mx = max3f(mx, s1[15], __shfl_xor(mx, 32, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // both halves now hold the row maximum
        af32x2 ps = {0.f, 0.f};
        if (PRE) {
            // mx = (tile maximum) - M.  Rescale when this is the wave's first tile or some query jumped by > 2^PRE_THR
            const bool need = tp == 0 || mx > PRE_THR;  // (a wave's first tile is tile pair 0, or it has no tile at all)
            if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform
                float m_new = m_run;
                if (need && mx > -INFINITY) {
                    const float t = m_run + mx;
                    const __bf16 hi = (__bf16)t;
                    const __bf16 lo = (__bf16)(t - (float)hi);
                    m_new = (float)hi + (float)lo;  // M stays exactly representable as the two bf16 parts
                }
                const float delta = m_new - m_run;
                m_run = m_new;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                af32x2 a = {__builtin_amdgcn_exp2f(s0[r]), __builtin_amdgcn_exp2f(s0[r + 1])};
                af32x2 b = {__builtin_amdgcn_exp2f(s1[r]), __builtin_amdgcn_exp2f(s1[r + 1])};
                s0[r] = a[0]; s0[r + 1] = a[1];
                s1[r] = b[0]; s1[r + 1] = b[1];
                ps += a;
                ps += b;
            }
        } else {
            const float m_new = fmaxf(m_run, mx);
            // rescale the running state only when some query of the wave saw a new maximum (wave-uniform branch)
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
                m_run = m_new;
            }
            const float mc = -m_run * scale_log2e;
            const af32x2 sc2 = {scale_log2e, scale_log2e}, mc2 = {mc, mc};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                af32x2 a = {s0[r], s0[r + 1]}, b = {s1[r], s1[r + 1]};
                a = a * sc2 + mc2;  // v_pk_fma_f32
                b = b * sc2 + mc2;
                a[0] = __builtin_amdgcn_exp2f(a[0]); a[1] = __builtin_amdgcn_exp2f(a[1]);
                b[0] = __builtin_amdgcn_exp2f(b[0]); b[1] = __builtin_amdgcn_exp2f(b[1]);
                s0[r] = a[0]; s0[r + 1] = a[1];
                s1[r] = b[0]; s1[r + 1] = b[1];
                ps += a;
                ps += b;
            }
        }
        l_run += ps[0] + ps[1];
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
            const abf16x4 a = *reinterpret_cast<const abf16x4 *>(Vtl + (vbase ^ (c0 << 4)));
            const abf16x4 b = *reinterpret_cast<const abf16x4 *>(Vtl + (vbase ^ (c1 << 4)));
            const abf16x8 v0 = abf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            // (volatile: left to itself the compiler pairs the pieces of rows qc and 32 + qc -- a constant 4096 bytes apart --
            // into one ds_read2st64_b64 and then needs 6 v_mov per k-step to regroup them into the two MFMA operands:
            // 24 VALU instructions per key tile; self-attention 53.5 -> 51.1 us, SF3D's fuse attentions 439 / 414 -> 409 / 389)
            typedef const volatile __attribute__((address_space(3))) abf16x4 *lds_vol_t;
            const abf16x4 c = *(lds_vol_t)(Vtl + (vbase ^ (c0 << 4)) + 4096);
            const abf16x4 d = *(lds_vol_t)(Vtl + (vbase ^ (c1 << 4)) + 4096);
            const abf16x8 v1 = abf16x8{c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, o1, 0, 0, 0);
        }
    };

    // A software-pipelined form of this loop -- the scores of tile pair tp + 1 requested from the matrix pipe BEFORE the softmax
    // of tile pair tp (K one tile pair ahead of V in the ring, two score accumulator sets, 159 VGPRs, 12- and 8-wave workgroups)
    // -- was built on these two lambdas, gave identical results and was no faster: 49.5 vs 48.2 us at 3072 x 3072 x 16.  PMC of
    // the loop (tools/attn_pmc.sh): VALU active 41 % of the SIMD cycles (32 quarter-rate v_exp_f32 per wave and key tile are
    // 512 of its ~720 VALU cycles), matrix pipe busy 31 %, some instruction active 68 %: the two do not overlap on a SIMD,
    // whatever the program order -- the loop is bound by their SUM.
    stage2(0, 0, 0, 0);
    for (int tp = 0; tp < ntp; ++tp) {
        const int buf = tp & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage2(buf ^ 1, tp + 1, buf ^ 1, tp + 1);
        if (tp * 128 + kh * 64 < Tk) {  // wave-uniform: a wholly out-of-range tile is skipped
            f32x16 s0, s1;
            qk(buf, s0, s1);
            softmax_pv(tp, buf, s0, s1);
        }
    }
    // ---- merge the two key halves of each query block through LDS: layout [qi][r (0..33)][lane]
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    __syncthreads();  // all tile reads done; staging memory becomes scratch
    float *scr = reinterpret_cast<float *>(smem) + qi * 34 * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { scr[r * 64] = o0[r]; scr[(16 + r) * 64] = o1[r]; }
        scr[32 * 64] = (PRE && !(l_tot > 0.f)) ? -INFINITY : m_run;  // a half without keys has no maximum (PRE starts M at 0)
        scr[33 * 64] = l_tot;
    }
    __syncthreads();
    if (kh == 0) {
        const float m1 = scr[32 * 64], l1 = scr[33 * 64];
        const float m = fmaxf(m_run, m1);
        const float sc = PRE ? 1.0f : scale_log2e;
        const float a0 = __builtin_amdgcn_exp2f((m_run - m) * sc);
        const float a1 = (l1 > 0.f) ? __builtin_amdgcn_exp2f((m1 - m) * sc) : 0.f;
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


// A "ping-pong" variant of this kernel (the two key-half groups offset by half an iteration, so that on every SIMD
// one wave runs its QK^T / PV MFMAs while its partner runs softmax; K/V prefetched two tile pairs ahead into LDS
// rings, one 8-wave workgroup per CU) was built, was bit-identical, and was SLOWER: 88 us vs 70 us at
// 3072 x 3072 x 16 heads, with the matrix and softmax segments adding up exactly (38 + 23 us) instead of overlapping.
// tools/micro/mfma_valu_overlap.hip shows why: on gfx950 a partner wave's fp32 FMA-class VALU work (v_fma, v_pk_fma,
// v_pk_mul, v_pk_add) does not overlap bf16 MFMAs on the same SIMD (times add), while v_exp, v_max3, v_cvt_pk and
// integer VALU do (times ~max).  The structure above therefore stays; what pays is fewer FMA-class instructions.
}  // namespace sculpt

using namespace sculpt;

static int attention_launch(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt, int ldvt, uint16_t *O,
                            int ldo, int Tq, int Tk, int heads, float scale, bool prescaled, sculpt_stream_t stream,
                            int batch = 1, long q_bs = 0, long k_bs = 0, long vt_bs = 0, long o_bs = 0) {
    SC_REQUIRE(Q && K && Vt && O, "attention: null argument");
    SC_REQUIRE(Tq >= 1 && Tk >= 1 && heads >= 1 && batch >= 1, "attention: bad shape Tq=%d Tk=%d heads=%d batch=%d", Tq, Tk, heads, batch);
    SC_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attention: row strides must keep 16-byte alignment");
    SC_REQUIRE(ldvt >= ((Tk + 63) / 64) * 64, "attention: ldvt=%d must be >= round_up(Tk=%d, 64)", ldvt, Tk);
    if (batch > 1) {
        SC_REQUIRE(q_bs % 8 == 0 && k_bs % 8 == 0 && vt_bs % 8 == 0 && o_bs % 4 == 0 && q_bs >= 0 && k_bs >= 0 && vt_bs >= 0 && o_bs > 0,
                   "attention: batch strides must keep 16-byte alignment (q %ld, k %ld, vt %ld, o %ld elements)", q_bs, k_bs, vt_bs, o_bs);
        // V^T of the batch entries: side by side in one [heads * 64][ldvt] array (stride = a column offset) or one array each
        const long tk64 = (long)((Tk + 63) / 64) * 64;
        SC_REQUIRE(vt_bs < ldvt ? (batch - 1) * vt_bs + tk64 <= ldvt : vt_bs >= (long)heads * 64 * ldvt,
                   "attention: V^T batch stride %ld neither keeps %d entries of round_up(Tk, 64) = %ld columns inside ldvt=%d nor skips a whole array",
                   vt_bs, batch, tk64, ldvt);
        SC_REQUIRE((long)batch * heads <= 65535, "attention: batch * heads = %ld exceeds the grid", (long)batch * heads);
    }
    // K / V tiles are staged with buffer addressing: 32-bit per-lane byte offsets from the (batch entry, head) base.  Past 2 GiB
    // an offset wraps or leaves the descriptor's range, and an out-of-range buffer load returns zeros silently -- refuse such shapes.
    {
        const long tk128 = (long)((Tk + 127) / 128) * 128;
        SC_REQUIRE(tk128 * ldk * 2 < 0x7fffffffL, "attention: K extent %ld x %d x 2 B is beyond the 2 GiB buffer range", tk128, ldk);
        SC_REQUIRE((long)64 * ldvt * 2 + tk128 * 2 < 0x7fffffffL, "attention: V^T extent 64 x %d x 2 B is beyond the 2 GiB buffer range", ldvt);
    }
    // Queries per workgroup: 128, 192 or 256 (8 / 12 / 16 waves, 2 - 4 per SIMD).  A workgroup's time grows with its query
    // count, a launch lasts ceil(workgroups / CUs) rounds (the 16-wave form fits one per CU; smaller ones are counted the same
    // way: a second resident workgroup shares the CU's matrix pipe).  3072 queries x 16 heads: 192 workgroups of 256 (3/4 of
    // the CUs for 8 units of time), 384 of 128 (two rounds of 4), 256 of 192 -- every CU once, 6 units.
    const int force = form_int("SCULPT_ATTN_FORM", "nqb", 0);   // A/B: nqb=4 / 6 / 8
    // (measured: self-attention 51.1 -> 48.8 us, cross 25.2 -> 23.5 us with 192; a smaller workgroup re-stages the head's K / V
    // more often, so it has to win by more than 15 %: SF3D's 27 648 queries, 54 against 56 units, ran 4 % slower with 192)
    const long bh = (long)heads * batch;
    int nqb = 8;
    const long cost8 = cdiv((long)cdiv(Tq, 256) * bh, (long)num_cus()) * 8;
    long best = cost8 * 100;
    for (int c : {6, 4}) {
        const long wgs = (long)cdiv(Tq, 32 * c) * bh, cost = cdiv(wgs, (long)num_cus()) * c;
        // pre-scaled queries run the pipelined loop at 128 / 192 queries (attention_pipe.hip: ~0.85 of the time per unit) -- SF3D's
        // 27 648 x 3089: 373 us with 192 (54 units) against 393 us with 256 (56 units)
        const long w = prescaled ? 98 : 115;
        if (cost * w < best) { best = cost * w; nqb = c; }
    }
    if (force == 4 || force == 6 || force == 8) nqb = force;
    const float sl = scale * 1.44269504088896340736f;
    hipStream_t st = as_stream(stream);
    const dim3 grid(cdiv(Tq, 32 * nqb), (unsigned)bh), block(128 * nqb);
    // the ragged last tile pair clamps its V^T columns to what the entry may read: side by side, the LAST entry's row ends
    // (batch - 1) * vt_bs columns earlier than ldvt says (a read past it runs into the next row -- past the array on the last one)
    const int vt_cols = (batch > 1 && vt_bs < ldvt) ? ldvt - (int)((batch - 1) * vt_bs) : ldvt;
    const AttnBatch ab{heads, vt_cols, q_bs, k_bs, vt_bs, o_bs};
#define SCULPT_ATTN_LAUNCH(NQB, PRE, SC) \
    hipLaunchKernelGGL((attention_kernel<NQB, PRE>), grid, block, 0, st, Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, SC, ab)
    // SCULPT_ATTN_FORM=nopipe: the phase-separated loop for pre-scaled queries too (tests pin the pipelined kernel against it)
    const bool pipe = prescaled && nqb != 8 && !form_has("SCULPT_ATTN_FORM", "nopipe");
    if (pipe) {
        attention_pipe_launch(nqb, grid, block, st, Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, ab);
    } else if (prescaled) {  // Q carries scale * log2(e) already
        if (nqb == 8) SCULPT_ATTN_LAUNCH(8, true, 1.0f);
        else if (nqb == 6) SCULPT_ATTN_LAUNCH(6, true, 1.0f);
        else SCULPT_ATTN_LAUNCH(4, true, 1.0f);
    } else {
        if (nqb == 8) SCULPT_ATTN_LAUNCH(8, false, sl);
        else if (nqb == 6) SCULPT_ATTN_LAUNCH(6, false, sl);
        else SCULPT_ATTN_LAUNCH(4, false, sl);
    }
#undef SCULPT_ATTN_LAUNCH
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sculpt_attention_bf16(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt,
                                     int ldvt, uint16_t *O, int ldo, int Tq, int Tk, int heads, float scale,
                                     sculpt_stream_t stream) {
    SC_REQUIRE(scale > 0.f && scale == scale, "attention: scale must be positive (pre-scaled queries: sculpt_attention_bf16_prescaled)");
    return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, heads, scale, false, stream);
}

extern "C" int sculpt_attention_bf16_prescaled(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt,
                                               int ldvt, uint16_t *O, int ldo, int Tq, int Tk, int heads,
                                               sculpt_stream_t stream) {
    return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, heads, 1.0f, true, stream);
}

extern "C" int sculpt_attention_bf16_batched(const uint16_t *Q, int ldq, int64_t q_bs, const uint16_t *K, int ldk, int64_t k_bs,
                                             const uint16_t *Vt, int ldvt, int64_t vt_bs, uint16_t *O, int ldo, int64_t o_bs,
                                             int Tq, int Tk, int heads, int batch, int prescaled, float scale,
                                             sculpt_stream_t stream) {
    SC_REQUIRE(prescaled || (scale > 0.f && scale == scale), "attention_batched: scale must be positive unless prescaled");
    return attention_launch(Q, ldq, K, ldk, Vt, ldvt, O, ldo, Tq, Tk, heads, prescaled ? 1.0f : scale, prescaled != 0, stream, batch,
                            (long)q_bs, (long)k_bs, (long)vt_bs, (long)o_bs);
}
