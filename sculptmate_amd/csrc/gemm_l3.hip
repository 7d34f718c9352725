// fp32-in / fp32-out GEMM with fp32 ARITHMETIC on the bf16 matrix pipe: both operands are split EXACTLY into three bf16 limbs
// while they are staged into LDS, and the six limb products of order >= 2^-16 are accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16 -- the density kernel's technique (triplane.hip) carried to the Linears and the
// per-head attention products of the transformer stack.
//
//   x = x1 + x2 + x3,  W = W1 + W2 + W3      each limb = round-to-nearest bf16 of the exact remainder; a 24-bit significand
//                                            minus two 8-bit limbs leaves <= 8 bits, so the third limb is exact; bf16 has the
//                                            fp32 exponent: no range limit
//   W.x = W1x3 + W3x1 + W2x2 + W1x2 + W2x1 + W1x1   (+ W2x3 + W3x2 + W3x3 < 2^-23 |W||x|, dropped)
// Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32: this is the reference's fp32 Linear
// (TripoSR/tsr/models/transformer/attention.py:569-653, basic_transformer_block.py:291-315; no autocast, generate.py:36-39)
// with 24-bit operands, at up to 16 / 6 = 2.7x the exact-fp32 matrix pipe's rate.
//
// Same interface, tiling (128 x 128 outputs, 2 x 2 waves of 2 x 2 accumulator tiles of 32 x 32) and epilogue as gemm_f32.hip.
// K-step 32.  Staging is through registers -- an LDS-DMA cannot convert -- : a thread loads 8 float4 (4 rows x 4 consecutive k
// of each operand) one K-step ahead, splits them (5.5 vector instructions per value: v_cvt_pk_bf16_f32, two expands, two exact
// subtractions) and writes the limbs as 8-byte pieces into a [limb][k-chunk of 8][row] image: a fragment (32 rows x 8 k) is
// 512 contiguous bytes, so every ds_read_b128 lane group covers all 16 slots of the bank row, and the chunk planes are
// 32 bytes off the bank period so the 8-byte writes of a 16-lane group do not collide either.  One 49-KiB buffer per workgroup
// (two barriers per K-step); up to three workgroups per CU overlap one's split / write phase with another's MFMAs.
#include <stdlib.h>

#include "gemm_f32.h"

namespace sculpt {

typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lbf16x2 __attribute__((ext_vector_type(2)));
typedef float lf32x2 __attribute__((ext_vector_type(2)));

static constexpr int L3_BK = 32;
static constexpr int L3_CS = 128 * 16 + 32;   // bytes from one k-chunk plane (128 rows x 16 B) to the next
static constexpr int L3_LT = 4 * L3_CS;       // one limb of one operand tile
static constexpr int L3_OP = 3 * L3_LT;       // one operand tile, three limbs

__device__ __forceinline__ unsigned l3_cvt_pk(float lo, float hi) {
    const lf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, lbf16x2));
}

// four consecutive-k fp32 values -> the 8-byte piece of each limb plane
__device__ __forceinline__ void l3_split4(const float4 &x, uint2 &p1, uint2 &p2, uint2 &p3) {
#pragma clang fp contract(off)
    const unsigned a1 = l3_cvt_pk(x.x, x.y), b1 = l3_cvt_pk(x.z, x.w);
    const float r0 = x.x - __uint_as_float(a1 << 16), r1 = x.y - __uint_as_float(a1 & 0xffff0000u);   // exact
    const float r2 = x.z - __uint_as_float(b1 << 16), r3 = x.w - __uint_as_float(b1 & 0xffff0000u);
    const unsigned a2 = l3_cvt_pk(r0, r1), b2 = l3_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(a2 << 16), s1 = r1 - __uint_as_float(a2 & 0xffff0000u);     // exact, <= 8 bits
    const float s2 = r2 - __uint_as_float(b2 << 16), s3 = r3 - __uint_as_float(b2 & 0xffff0000u);
    p1 = make_uint2(a1, b1);
    p2 = make_uint2(a2, b2);
    p3 = make_uint2(l3_cvt_pk(s0, s1), l3_cvt_pk(s2, s3));
}

// PIPE = false: the plain form -- per K-step {barrier; split + write; barrier; load next; 48 MFMAs}.  The split is ~180 vector
// instructions per wave and K-step, and on a SIMD the vector phase of one wave does not slide under the matrix phase of
// another (tools/micro/mfma_phase.hip): the matrix pipe sits idle for it (FF1 300 us = 41 % of the bf16 peak by executed FLOPs).
// PIPE = true (default): the split of K-step kt + 1 is issued in the shadows of the MFMAs of K-step kt by the SAME wave (a wave's
// own vector instructions behind its own MFMA are free up to ~6 per MFMA, tools/micro/mfma_fill.hip) -- the limbs wait in 48
// registers for the barrier, then only the 24 ds_write_b64 and the next global loads sit between the two barriers.
// BM = 128 activation rows per tile, or 64 (plain loop only): launches that would leave CUs without a tile at 128 rows -- the
// N = 1024 projections of the backbone (192 tiles of 128 x 128 on 256 CUs), everything in the 1025-token image tokenizer -- get
// twice the tiles, a wave then owns 64 weight rows x 32 activation rows.
template <int EPI, bool PIPE, int BM = 128>
__global__ __launch_bounds__(256, 2) void gemm_l3_kernel(GemmF32Args g_in) {
    static_assert(BM == 128 || (BM == 64 && !PIPE), "64-row tiles run the plain K loop");
    constexpr int JT = BM / 64, AI = BM / 32;   // 32-row activation sub-tiles per wave; float4 per thread and K-step of the activation tile
    const GemmF32Args g = f32_batch_entry(g_in);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * L3_OP];   // [W limbs | A limbs] (the A planes keep 128-row strides)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? FBW / 2 : FBW;
    const int n0 = blockIdx.x * NOUT, m0 = blockIdx.y * BM;

    // staging: a tile is 128 rows x 32 k = 1024 float4; thread t takes k-quad t % 8 of rows t / 8 + 32 i (a row's 128 bytes are
    // one cache line read by 8 neighbouring lanes)
    const int sr = tid >> 3, kq = tid & 7;
    const float *wp[4], *ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wp[i] = g.W + (long)f32_tile_wrow<EPI>(g, n0, sr + 32 * i) * g.ldw + 4 * kq;
        ap[i] = g.A + (long)min(m0 + sr + 32 * (i < AI ? i : 0), g.M - 1) * g.lda + 4 * kq;
    }
    // LDS byte offset of this thread's 8-byte piece of row sr in a limb plane (chunk = kq / 2, half = kq % 2)
    const int wofs = (kq >> 1) * L3_CS + sr * 16 + (kq & 1) * 8;
    float4 rw[4], ra[4];
    uint2 pw[4][3], pa[4][3];   // PIPE: the limbs of the next K-step, waiting for the barrier
    const int nk = g.K / L3_BK;

    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // fragment read offsets: k-step s reads chunk 2 s + lh; rows wr * 64 + i * 32 + l31 (W) / wc * 32 JT + j * 32 + l31 (A)
    const int wfo = lh * L3_CS + (wr * 64 + l31) * 16;
    const int afo = L3_OP + lh * L3_CS + (wc * (32 * JT) + l31) * 16;

    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rw[i] = *reinterpret_cast<const float4 *>(wp[i] + kt * L3_BK);
            if (i < AI) ra[i] = *reinterpret_cast<const float4 *>(ap[i] + kt * L3_BK);
        }
    };
    auto split_all = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            l3_split4(rw[i], pw[i][0], pw[i][1], pw[i][2]);
            if (i < AI) l3_split4(ra[i], pa[i][0], pa[i][1], pa[i][2]);
        }
    };
    auto write_all = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char *d = smem + wofs + i * (32 * 16);
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                *reinterpret_cast<uint2 *>(d + l * L3_LT) = pw[i][l];
                if (i < AI) *reinterpret_cast<uint2 *>(d + L3_OP + l * L3_LT) = pa[i][l];
            }
        }
    };
    auto mfma_kstep = [&](int s) {
        lbf16x8 wf[2][3], af[JT][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                wf[i][l] = *reinterpret_cast<const lbf16x8 *>(smem + wfo + l * L3_LT + 2 * s * L3_CS + i * (32 * 16));
                if (i < JT) af[i < JT ? i : 0][l] = *reinterpret_cast<const lbf16x8 *>(smem + afo + l * L3_LT + 2 * s * L3_CS + i * (32 * 16));
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                f32x16 c = acc[i][j];   // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][2], af[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], af[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], af[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };

    gload(0);
    if constexpr (PIPE) {
        split_all();
        write_all();
        __syncthreads();
        if (nk > 1) gload(1);
        // One K-step = 8 groups of 6 MFMAs (k-step s, weight tile i, activation tile j); group g carries the split of float4 g of
        // the NEXT K-step (g < 4: weight rows, else activation rows) in its shadows, cut into dependency stages of <= 6 vector
        // instructions per MFMA and fenced (sched_barrier) so that hipcc keeps the interleave -- left alone it puts the whole
        // split in front of and behind the MFMA block (sched_group_barrier patterns were ignored).  The 12 fragments of k-step 1
        // are read during k-step 0's groups; only k-step 0's own 12 reads are exposed, once per K-step.
#define L3_FENCE __builtin_amdgcn_sched_barrier(0)
#define L3_MF(W, A) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W, A, c, 0, 0, 0)
#define L3_RD(dst, base, l, s_, i_) dst = *reinterpret_cast<const lbf16x8 *>(smem + (base) + (l) * L3_LT + 2 * (s_) * L3_CS + (i_) * (32 * 16))
        for (int kt = 0; kt < nk; ++kt) {
            lbf16x8 wf0[2][3], af0[2][3], wf1[2][3], af1[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int l = 0; l < 3; ++l) { L3_RD(wf0[i][l], wfo, l, 0, i); L3_RD(af0[i][l], afo, l, 0, i); }
            L3_FENCE;
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
                const int s_ = gi >> 2, i = (gi >> 1) & 1, j = gi & 1;
                const float4 x = gi < 4 ? rw[gi & 3] : ra[gi & 3];
                const lbf16x8 *wf = s_ ? wf1[i] : wf0[i], *af = s_ ? af1[j] : af0[j];
                f32x16 c = acc[i][j];
                unsigned a1, b1, a2, b2, a3, b3;
                float r0, r1, r2, r3, t0, t1, t2, t3;
                float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
                {
#pragma clang fp contract(off)
                    // (the empty asm statements pin each stage inside its slot: pure arithmetic is otherwise hoisted in front of
                    // the first fence or sunk to its only use behind the last one -- IR-level motion the fences do not see)
                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                    L3_MF(wf[0], af[2]);   // smallest terms first
                    a1 = l3_cvt_pk(x0, x1); b1 = l3_cvt_pk(x2, x3);
                    t0 = __uint_as_float(a1 << 16); t1 = __uint_as_float(a1 & 0xffff0000u);
                    t2 = __uint_as_float(b1 << 16); t3 = __uint_as_float(b1 & 0xffff0000u);
                    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                    if (gi < 4) { L3_RD(wf1[gi >> 1][(gi & 1) * 2], wfo, (gi & 1) * 2, 1, gi >> 1); }   // k-step 1 fragments, three per group
                    L3_FENCE;
                    L3_MF(wf[2], af[0]);
                    r0 = x0 - t0; r1 = x1 - t1; r2 = x2 - t2; r3 = x3 - t3;   // exact
                    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
                    if (gi < 4) { L3_RD(af1[gi >> 1][(gi & 1) * 2], afo, (gi & 1) * 2, 1, gi >> 1); }
                    L3_FENCE;
                    L3_MF(wf[1], af[1]);
                    a2 = l3_cvt_pk(r0, r1); b2 = l3_cvt_pk(r2, r3);
                    t0 = __uint_as_float(a2 << 16); t1 = __uint_as_float(a2 & 0xffff0000u);
                    t2 = __uint_as_float(b2 << 16); t3 = __uint_as_float(b2 & 0xffff0000u);
                    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                    if (gi < 4 && (gi & 1) == 0) { L3_RD(wf1[gi >> 1][1], wfo, 1, 1, gi >> 1); }
                    if (gi < 4 && (gi & 1) == 1) { L3_RD(af1[gi >> 1][1], afo, 1, 1, gi >> 1); }
                    L3_FENCE;
                    L3_MF(wf[0], af[1]);
                    r0 = r0 - t0; r1 = r1 - t1; r2 = r2 - t2; r3 = r3 - t3;       // exact, <= 8 bits
                    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
                    L3_FENCE;
                    L3_MF(wf[1], af[0]);
                    a3 = l3_cvt_pk(r0, r1); b3 = l3_cvt_pk(r2, r3);
                    asm volatile("" : "+v"(a3), "+v"(b3));
                    L3_FENCE;
                    L3_MF(wf[0], af[0]);
                    L3_FENCE;
                }
                acc[i][j] = c;
                if (gi < 4) { pw[gi & 3][0] = make_uint2(a1, b1); pw[gi & 3][1] = make_uint2(a2, b2); pw[gi & 3][2] = make_uint2(a3, b3); }
                else { pa[gi & 3][0] = make_uint2(a1, b1); pa[gi & 3][1] = make_uint2(a2, b2); pa[gi & 3][2] = make_uint2(a3, b3); }
            }
            if (kt + 1 < nk) {
                __syncthreads();   // every wave has read K-step kt's fragments
                write_all();
                if (kt + 2 < nk) gload(kt + 2);
                __syncthreads();
            }
        }
#undef L3_FENCE
#undef L3_MF
#undef L3_RD
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt > 0) __syncthreads();   // every wave has read the previous K-step's fragments
            split_all();
            write_all();
            __syncthreads();
            if (kt + 1 < nk) gload(kt + 1);   // the next K-step's operands travel while this one is multiplied
            mfma_kstep(0);
            mfma_kstep(1);
        }
    }
    f32_tile_epilogue<EPI, JT>(g, acc, n0, m0, wr, wc, l31, lh);
}

int gemm_l3_launch(const GemmF32Args &g, int epilogue, int batch, hipStream_t st) {
    // SCULPT_L3_TILE tokens (A/B, tests; read per call): nopipe = the plain (phase-separated) K loop; bm64 / nobm64 = always /
    // never the 64-row tile (default: by CU fill)
    const bool pipe = !form_has("SCULPT_L3_TILE", "nopipe");
    const int gx = epilogue == SCULPT_EPI_GEGLU ? g.N / 64 : cdiv(g.N, FBW);
    const long tiles128 = (long)gx * cdiv(g.M, 128) * batch;
    // fewer than 3 tiles per CU at 128 rows (tools/time_l3_gemm.py: fused Q|K|V 149 -> 133 us, to_out 65 -> 56, FF2 213 -> 190, the
    // image tokenizer's f2 151 -> 99; FF1 with 6 tiles per CU: 286 -> 313, stays)
    const bool bm64 = form_has("SCULPT_L3_TILE", "bm64") ? true : (form_has("SCULPT_L3_TILE", "nobm64") ? false : tiles128 < 3L * num_cus());
    const int mt = cdiv(g.M, bm64 ? 64 : 128);
#define L3_GO(E)                                                                                                          \
    do {                                                                                                                  \
        if (bm64) hipLaunchKernelGGL((gemm_l3_kernel<E, false, 64>), dim3(gx, mt, batch), dim3(256), 0, st, g);           \
        else if (pipe) hipLaunchKernelGGL((gemm_l3_kernel<E, true, 128>), dim3(gx, mt, batch), dim3(256), 0, st, g);      \
        else hipLaunchKernelGGL((gemm_l3_kernel<E, false, 128>), dim3(gx, mt, batch), dim3(256), 0, st, g);               \
    } while (0)
    if (epilogue == SCULPT_EPI_GEGLU) L3_GO(SCULPT_EPI_GEGLU);
    else if (epilogue == SCULPT_EPI_GELU) L3_GO(SCULPT_EPI_GELU);
    else if (epilogue == SCULPT_EPI_RELU) L3_GO(SCULPT_EPI_RELU);
    else L3_GO(SCULPT_EPI_NONE);
#undef L3_GO
    return 0;
}

}  // namespace sculpt
