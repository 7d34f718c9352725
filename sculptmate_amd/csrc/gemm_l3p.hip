// Three-limb GEMM on operands that are ALREADY split ("limbs once"): the same arithmetic as gemm_l3.hip --
//   W.x = W1x3 + W3x1 + W2x2 + W1x2 + W2x1 + W1x1 per 16 k, fp32 accumulation on v_mfma_f32_32x32x16_bf16, k ascending --
// so the results are BIT-IDENTICAL to gemm_l3_kernel's, but the exact split x = x1 + x2 + x3 is no longer redone by every column
// tile of every launch (~5.5 vector instructions per staged value, 2/3 of them on the weights, which never change):
//   * weights are split once at load time,
//   * activations are written as limbs by the kernel that produces them (LayerNorm, attention, the GELU / GEGLU epilogue of the
//     previous Linear; sculpt_limbs_split for the rest),
// into the LIMB-TILED layout below, which is at the same time the LDS image of this kernel: a K-tile of a 32-row block is 3 KiB
// of contiguous HBM that three global_load_lds_dwordx4 wave instructions copy straight into the ring -- no register round trip,
// no vector instruction in the K loop but the MFMAs and their fragment reads.
//
// Limb-tiled matrix X [R][K] (K % 32 == 0), rows in blocks of 32, k in chunks of 8:
//     byte offset of limb l (0 = leading) of X[r][k] = (((r / 32) * (K / 8) + k / 8) * 3 + l) * 512 + (r % 32) * 16 + (k % 8) * 2
// i.e. [row block][k chunk][limb][32 rows][8 k] of bf16: one MFMA fragment (32 rows x 8 k of one limb) is 512 contiguous bytes, read
// by ds_read_b128 with lane l31 on row l31 (conflict-free: the 16 lanes of a read group sit on 16 different 16-byte slots), the
// lanes 32..63 on the next chunk.  ceil(R / 32) blocks are allocated; rows >= R of the last block hold zeros (sculpt_limbs_split)
// or whatever the producer left there -- they only reach output rows / columns that are never stored.
//
// Tile 128 weight rows x BM (128 / 64) activation rows, 2 x 2 waves of 2 x JT accumulator tiles of 32 x 32 like gemm_l3_kernel (same
// epilogue), K-tile = 16 (one MFMA k-step: 24 KiB per stage at BM = 128), 3-stage ring with two K-tiles in flight (counted vmcnt +
// raw s_barrier, as gemm.hip), 72 KiB -> two workgroups per CU.
#include <stdlib.h>

#include "gemm_f32.h"
#include "limbs.h"

namespace sculpt {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

// one limb product of a 32 x 32 x 16 step, operands as raw 16-byte fragments
template <int FMT>
__device__ __forceinline__ f32x16 l3p_mfma(pu32x4 w, pu32x4 x, f32x16 c) {
    if constexpr (FMT == LT_F16X2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, w), __builtin_bit_cast(pf16x8, x), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, w), __builtin_bit_cast(pbf16x8, x), c, 0, 0, 0);
}
typedef __attribute__((address_space(3))) void *p_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *p_gbl_ptr_t;

struct GemmL3pArgs {
    GemmF32Args g;            // A / W unused; everything else as gemm_l3_kernel
    const unsigned char *A_lt;
    const unsigned char *W_lt;
    unsigned char *out_lt;    // when set: the result leaves as limbs (limb-tiled [M][N or N/2 outputs]) instead of g.out
    int a_blocks;             // 32-row blocks allocated in A_lt
    int out_k8;               // 16-byte chunks per row of out_lt = output columns / 8
    int out_fmt;              // limb format of out_lt (limbs.h)
    // XCD-aware tile order (common.h xcd_tile): the workgroups of one XCD (private L2) take a contiguous band of the tile grid -- a
    // band of activation rows with every weight tile, or (n_major = 1, W the larger operand) a band of weight rows with every
    // activation tile; -1 = the natural order (A/B)
    int n_major;
    unsigned long long *stamps;   // -DSCULPT_EXPERIMENTS builds (common.h GEMM_STAMP); nullptr otherwise
};

// The epilogue when the result leaves as limbs: a lane holds four consecutive output columns of one row = one 8-byte half of a
// 16-byte chunk per limb; the 64 lanes of a wave write 512 contiguous bytes per limb and register quad.
template <int EPI, int JT>
__device__ __forceinline__ void l3p_epilogue_limbs(const GemmL3pArgs &a, const f32x16 (&acc)[2][JT], int n0, int m0, int wr, int wc,
                                                   int l31, int lh) {
    const GemmF32Args &g = a.g;
    const int out_blocks = (g.M + 31) >> 5;
    // the bias quads of this lane (they depend on (q4, i) only), all before the first limb store: a load behind a store waits for
    // it as well (f32_tile_epilogue, gemm_f32.h)
    float4 bq[4][2];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = (EPI == SCULPT_EPI_GEGLU) ? (i ? g.N : 0) + n0 + wr * 32 + 8 * q4 + 4 * lh : n0 + wr * 64 + i * 32 + 8 * q4 + 4 * lh;
            bq[q4][i] = *reinterpret_cast<const float4 *>(g.bias + n);
        }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int m = m0 + wc * (32 * JT) + j * 32 + l31;   // rows >= M of the last block are pad rows: written, never read as results
        if ((m >> 5) >= out_blocks) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if (EPI == SCULPT_EPI_GEGLU) {
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
                lt_store4(a.out_lt, a.out_k8, m, n, o, a.out_fmt);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int n = n0 + wr * 64 + i * 32 + 8 * q4 + 4 * lh;
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
                    lt_store4(a.out_lt, a.out_k8, m, n, o, a.out_fmt);
                }
            }
        }
    }
}

// NW = 4: 2 x 2 waves of 2 x JT accumulator tiles (64 weight x BM / 2 activation rows per wave).  NW = 8 (BM = 128): 2 x 4 waves of
// 2 x 1 tiles -- two waves per SIMD inside ONE workgroup, for the launches that put at most one workgroup on a CU (the backbone's
// N = 1024 projections: 192 tiles): a lone wave per SIMD cannot hide its LDS-DMA issue and fragment-read latency behind MFMAs.
// (One 128 x 96 tile per CU on 6 waves -- 256 tiles for the 3072 x 1024 outputs that 128 x 128 tiles spread over 192 CUs -- measured
// no faster: o / q 49.3 against 47.6 us, FF2 164 against 157: these launches run at the clock the chip holds under the matrix load,
// not at a rate the idle quarter of the CUs could add to.)
// FMT (limbs.h): LT_BF16X3 -- three limbs, the six products above; LT_F16X2 -- two fp16 limbs, W.x = W2x1 + W1x2 + W1x1 per 16 k
// (W2x2 < 2^-22 |W||x| dropped): half the matrix work, the K-tile of a row block is 2 KiB.
template <int EPI, int BM, int NW, int FMT>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_l3p_kernel(GemmL3pArgs a) {
    static_assert(((BM == 128 || BM == 64) && NW == 4) || (NW == 8 && BM == 128), "tile forms");
    constexpr int NL = FMT == LT_F16X2 ? 2 : 3;
    constexpr int NWC = NW / 2;           // waves along the activation rows
    constexpr int JT = BM / 32 / NWC;     // 32-row activation sub-tiles per wave
    constexpr int ARB = BM / 32;          // activation row blocks per tile
    constexpr int NRB = 4 + ARB;          // row blocks per ring stage: [4 weight | ARB activation]
    // K-tile per ring stage: one 16-k MFMA step with three limbs; two with two limbs (the same 3 - 4 KiB per row block and stage,
    // and a barrier every 12 - 24 MFMAs per wave instead of every 6 - 12: FF2 104 -> 9x us, fused Q|K|V 77 -> 6x)
    constexpr int KT = NL == 2 ? 2 : 1;
    constexpr int RBK = NL * 1024 * KT;   // bytes of one K-tile (KT x 2 chunks x NL limbs x 512 B) of one row block
    constexpr int STG = NRB * RBK;
    // ring depth: three stages (two K-tiles in flight); two for the 4-wave 128-row two-limb tile (64 KiB: two workgroups per CU)
    constexpr int NST = (NL == 2 && NW == 4 && BM == 128) ? 2 : 3;
    constexpr int DIST = NST - 1;
    constexpr int NPC = NL * KT;          // 1-KiB LDS-DMA pieces per row block and stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STG];
    const GemmF32Args &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? FBW / 2 : FBW;
    int nt = blockIdx.x, mt = blockIdx.y;
    if (a.n_major >= 0) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int tile = xcd_tile(blockIdx.y * gx + blockIdx.x, gx * gy);
        nt = a.n_major ? tile / gy : tile % gx;
        mt = a.n_major ? tile % gy : tile / gx;
    }
    const int n0 = nt * NOUT, m0 = mt * BM;
    GEMM_STAMP(a, 0);
    const long kblk = (long)g.K * (64 * NL);   // bytes of one 32-row block: K / 8 chunks x NL limbs x 512
    // staging: wave w copies row block w of the stage and (when there are more blocks than waves) row block w + NW; blocks 0..3 are
    // the tile's weight rows (a GEGLU weight is stored with its row blocks already in the tile's value / gate order: 4 blocks per
    // 64 output columns), blocks 4.. its activation rows
    auto src_of = [&](int rb) -> const unsigned char * {
        return rb < 4 ? a.W_lt + ((long)nt * 4 + rb) * kblk + lane * 16
                      : a.A_lt + (long)min((m0 >> 5) + rb - 4, a.a_blocks - 1) * kblk + lane * 16;
    };
    const bool two = wave + NW < NRB;     // wave-uniform
    const unsigned char *src0 = src_of(wave), *src1 = src_of(two ? wave + NW : wave);
    const int dst0 = wave * RBK, dst1 = (wave + NW) * RBK;

#define L3P_STAGE(buf, kt)                                                                                                  \
    do {                                                                                                                    \
        unsigned char *sb = smem + (buf) * STG;                                                                             \
        const long ko = (long)(kt) * RBK;                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < NPC; ++q_)                                                                  \
            __builtin_amdgcn_global_load_lds((p_gbl_ptr_t)(src0 + ko + q_ * 1024), (p_lds_ptr_t)(sb + dst0 + q_ * 1024), 16, 0, 0); \
        if (NRB > NW && two) {                                                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < NPC; ++q_)                                                              \
                __builtin_amdgcn_global_load_lds((p_gbl_ptr_t)(src1 + ko + q_ * 1024), (p_lds_ptr_t)(sb + dst1 + q_ * 1024), 16, 0, 0); \
        }                                                                                                                   \
    } while (0)

    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // fragment of limb l: + l * 512; lane (l31, lh): chunk 2 s + lh of the K-tile (s = 16-k step), row l31
    const int wfo = (wr * 2) * RBK + lh * (NL * 512) + l31 * 16;
    const int afo = (4 + wc * JT) * RBK + lh * (NL * 512) + l31 * 16;

    const int nk = g.K / (16 * KT);
    L3P_STAGE(0, 0);
    if (DIST > 1 && nk > 1) L3P_STAGE(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        // wait until K-tile kt has landed; with three stages K-tile kt + 1 (if issued) stays in flight
        if (DIST > 1 && kt + 1 < nk) {
            if (NRB > NW && two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#ifdef SCULPT_EXPERIMENTS
        if (kt == 0) GEMM_STAMP(a, 1);
#endif
        // every wave finished reading stage (kt - 1) % NST == (kt + DIST) % NST before it passed the barrier
        if (kt + DIST < nk) L3P_STAGE((kt + DIST) % NST, kt + DIST);
        const unsigned char *sb = smem + (kt % NST) * STG;
#pragma unroll
        for (int ss = 0; ss < KT; ++ss) {
            pu32x4 wf[2][NL], af[JT][NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i][l] = *reinterpret_cast<const pu32x4 *>(sb + wfo + i * RBK + (2 * ss * NL + l) * 512);
#pragma unroll
                for (int j = 0; j < JT; ++j) af[j][l] = *reinterpret_cast<const pu32x4 *>(sb + afo + j * RBK + (2 * ss * NL + l) * 512);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    f32x16 c = acc[i][j];   // smallest terms first
                    if constexpr (NL == 3) {   // the order of gemm_l3_kernel
                        c = l3p_mfma<FMT>(wf[i][0], af[j][2], c);
                        c = l3p_mfma<FMT>(wf[i][2], af[j][0], c);
                        c = l3p_mfma<FMT>(wf[i][1], af[j][1], c);
                        c = l3p_mfma<FMT>(wf[i][0], af[j][1], c);
                        c = l3p_mfma<FMT>(wf[i][1], af[j][0], c);
                        c = l3p_mfma<FMT>(wf[i][0], af[j][0], c);
                    } else {
                        c = l3p_mfma<FMT>(wf[i][1], af[j][0], c);
                        c = l3p_mfma<FMT>(wf[i][0], af[j][1], c);
                        c = l3p_mfma<FMT>(wf[i][0], af[j][0], c);
                    }
                    acc[i][j] = c;
                }
        }
    }
#undef L3P_STAGE
    GEMM_STAMP(a, 2);
    if (a.out_lt) l3p_epilogue_limbs<EPI, JT>(a, acc, n0, m0, wr, wc, l31, lh);
    else f32_tile_epilogue<EPI, JT>(g, acc, n0, m0, wr, wc, l31, lh);
    GEMM_STAMP(a, 3);
    GEMM_STAMP(a, 4);
    GEMM_STAMP_IDS(a);
}

// fp32 [R][K] (row stride ld) times `scale` (a power of two: exact) -> limb-tiled; one thread per (row of a block, k chunk): the 32
// threads of a block row group write 512 contiguous bytes per limb.  Rows >= R of the last block are written as zeros.
template <int FMT>
__global__ __launch_bounds__(256) void limbs_split_kernel(const float *__restrict__ src, long ld, int R, int K, float scale,
                                                          unsigned char *__restrict__ dst) {
    constexpr int NL = FMT == LT_F16X2 ? 2 : 3;
    const int k8 = K >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)((R + 31) >> 5) * k8 * 32;
    if (idx >= total) return;
    const int r32 = (int)(idx & 31);
    const long bc = idx >> 5;
    const int kc = (int)(bc % k8);
    const long rb = bc / k8;
    const long row = rb * 32 + r32;
    float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < R) {
        const float4 x0 = *reinterpret_cast<const float4 *>(src + row * ld + kc * 8);
        const float4 x1 = *reinterpret_cast<const float4 *>(src + row * ld + kc * 8 + 4);
        lo[0] = x0.x * scale; lo[1] = x0.y * scale; lo[2] = x0.z * scale; lo[3] = x0.w * scale;
        hi[0] = x1.x * scale; hi[1] = x1.y * scale; hi[2] = x1.z * scale; hi[3] = x1.w * scale;
    }
    unsigned char *d = dst + (bc * NL) * 512 + r32 * 16;
    if constexpr (FMT == LT_F16X2) {
        uint2 a1, a2, b1, b2;
        lt_split4_h(lo, a1, a2);
        lt_split4_h(hi, b1, b2);
        *reinterpret_cast<uint4 *>(d) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        *reinterpret_cast<uint4 *>(d + 512) = make_uint4(a2.x, a2.y, b2.x, b2.y);
    } else {
        uint2 a1, a2, a3, b1, b2, b3;
        lt_split4(lo, a1, a2, a3);
        lt_split4(hi, b1, b2, b3);
        *reinterpret_cast<uint4 *>(d) = make_uint4(a1.x, a1.y, b1.x, b1.y);
        *reinterpret_cast<uint4 *>(d + 512) = make_uint4(a2.x, a2.y, b2.x, b2.y);
        *reinterpret_cast<uint4 *>(d + 1024) = make_uint4(a3.x, a3.y, b3.x, b3.y);
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

static bool lt_fmt_ok(int f) { return f == LT_BF16X3 || f == LT_F16X2; }

size_t sculpt_limbs_bytes(int rows, int K, int format) {
    return (size_t)((rows + 31) / 32) * (size_t)K * (size_t)(64 * lt_limbs(format));
}

int sculpt_limbs_split(const float *src, int ld, int rows, int K, float scale, int format, void *dst, sculpt_stream_t stream) {
    SC_REQUIRE(src && dst, "limbs_split: null argument");
    SC_REQUIRE(lt_fmt_ok(format), "limbs_split: unknown limb format %d", format);
    SC_REQUIRE(rows >= 1 && K >= 32 && K % 32 == 0 && ld >= K && ld % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
               "limbs_split: bad shape rows=%d K=%d ld=%d (K %% 32 == 0, ld %% 4 == 0, 16-byte aligned)", rows, K, ld);
    SC_REQUIRE(scale > 0.f && scale == scale && scale < INFINITY, "limbs_split: scale must be a positive finite number (a power of two)");
    const long total = (long)((rows + 31) / 32) * (K / 8) * 32;
    unsigned char *d = reinterpret_cast<unsigned char *>(dst);
    if (format == LT_F16X2)
        hipLaunchKernelGGL(limbs_split_kernel<LT_F16X2>, dim3((unsigned)cdiv(total, 256L)), dim3(256), 0, as_stream(stream), src, (long)ld,
                           rows, K, scale, d);
    else
        hipLaunchKernelGGL(limbs_split_kernel<LT_BF16X3>, dim3((unsigned)cdiv(total, 256L)), dim3(256), 0, as_stream(stream), src, (long)ld,
                           rows, K, scale, d);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_gemm_l3p(const void *A_lt, const void *W_lt, int format, float alpha, const float *bias, const float *residual, int ldr,
                    float *out, int ldo, float *out_t, int ldt, int n_split, void *out_lt, int out_format, int M, int N, int K,
                    int epilogue, sculpt_stream_t stream) {
    SC_REQUIRE(A_lt && W_lt && (out || out_t || out_lt), "gemm_l3p: null argument");
    SC_REQUIRE(lt_fmt_ok(format) && (!out_lt || lt_fmt_ok(out_format)), "gemm_l3p: unknown limb format %d / %d", format, out_format);
    SC_REQUIRE(M >= 1 && K >= 32 && K % 32 == 0, "gemm_l3p: bad shape M=%d K=%d (K %% 32 == 0)", M, K);
    SC_REQUIRE(alpha == alpha && alpha != 0.f && fabsf(alpha) < INFINITY, "gemm_l3p: alpha must be finite and non-zero");
    SC_REQUIRE(((uintptr_t)A_lt & 15) == 0 && ((uintptr_t)W_lt & 15) == 0 && ((uintptr_t)out_lt & 15) == 0, "gemm_l3p: limb arrays must be 16-byte aligned");
    const bool geglu = epilogue == SCULPT_EPI_GEGLU;
    if (geglu) SC_REQUIRE(N % 64 == 0 && (out || out_lt) && !residual && !out_t, "gemm_l3p(GEGLU): N %% 64 == 0, plain or limb output only");
    else SC_REQUIRE(N % 128 == 0 && (epilogue == SCULPT_EPI_NONE || epilogue == SCULPT_EPI_GELU || epilogue == SCULPT_EPI_RELU),
                    "gemm_l3p: N=%d must be a multiple of 128; epilogue %d", N, epilogue);
    SC_REQUIRE(!out_lt || (!residual && !out_t && !out), "gemm_l3p: a limb output excludes the fp32 outputs and the residual");
    SC_REQUIRE(ldo % 4 == 0 && (!residual || ldr % 4 == 0), "gemm_l3p: row strides must be multiples of 4");
    if (n_split <= 0 || n_split > N) n_split = N;
    SC_REQUIRE(n_split % 4 == 0 && (n_split == N || out_t), "gemm_l3p: bad n_split");
    if (!bias) {
        long zn = 0;
        bias = zero_floats_page(&zn);
        SC_REQUIRE(bias && (geglu ? 2L * N : (long)N) + 4 <= zn, "gemm_l3p: N=%d too large without a bias", N);
    }
    SC_REQUIRE(((uintptr_t)bias & 15) == 0, "gemm_l3p: bias must be 16-byte aligned");
    GemmL3pArgs a;
    a.g = GemmF32Args{nullptr, 0, nullptr, 0, bias, residual, ldr, out, ldo, out_t, ldt, M, N, K, n_split, N, alpha, 0, 0, 0};
    a.A_lt = reinterpret_cast<const unsigned char *>(A_lt);
    a.W_lt = reinterpret_cast<const unsigned char *>(W_lt);
    a.out_lt = reinterpret_cast<unsigned char *>(out_lt);
    a.a_blocks = (M + 31) / 32;
    a.out_k8 = N / 8;
    a.out_fmt = out_format;
    a.stamps = nullptr;
#ifdef SCULPT_EXPERIMENTS
    a.stamps = g_gemm_stamps;
#endif
    hipStream_t st = as_stream(stream);
    const int gx = geglu ? N / 64 : N / 128;
    // XCD band order: measured equal to the natural order on every shape of the two transformers (tools/time_l3p.py; the operands
    // sit in the Infinity Cache): off unless SCULPT_L3_TILE has the token "xcd"
    a.n_major = form_has("SCULPT_L3_TILE", "xcd") ? ((geglu ? 2 * N : N) > M ? 1 : 0) : -1;
    // Tile form by the number of 128 x 128 tiles (tools/time_l3p.py, one MI355X, three bf16 limbs, us; 4 waves 128 / 4 waves 64 /
    // 8 waves 128):
    //   image tokenizer (1025 rows)  o    54 tiles  34.7 / 24.5 / 31.0     f2   54 tiles  115.8 / 82.6 / 108.6
    //                                qkv 162 tiles  36.8 / 36.6 / 33.4     f1  216 tiles   49.4 / 46.8 /  42.7
    //   backbone (3072 rows)         o / q 192      52.0 / 53.5 / 49.1     FF2 192        163.6 / 175.3 / 153.6
    //                                Q|K|V 576     127.5 / 136.5 / 124.0   FF1 1536       248.1 / 297.1 / 253.2
    // fewer tiles than half the CUs: 64-row tiles (twice the workgroups); up to three per CU: the 8-wave form (two waves per SIMD
    // even where a CU holds one workgroup); more: 4 waves, two workgroups per CU.
    // SCULPT_L3_TILE tokens (A/B, tests; read per call): bm64 / nobm64 = always / never the 64-row tile; nw8 / nonw8 = always / never
    // 8 waves on the 128-row tile
    const long tiles128 = (long)gx * cdiv(M, 128);
    const bool bm64 = form_has("SCULPT_L3_TILE", "bm64") ? true : (form_has("SCULPT_L3_TILE", "nobm64") ? false : 2 * tiles128 < num_cus());
    const int f8 = form_has("SCULPT_L3_TILE", "nw8") ? 1 : (form_has("SCULPT_L3_TILE", "nonw8") ? 0 : -1);
    // (two fp16 limbs, tools/time_l3p_f16_forms.py: the 8-wave form only where a CU holds at most one workgroup -- fused Q|K|V
    // 77.8 against 80.8 us and FF1 + GEGLU 158 against 173 on 4 waves)
    const bool nw8 = !bm64 && (f8 >= 0 ? f8 != 0 : tiles128 <= (format == LT_F16X2 ? 1L : 3L) * num_cus());
#define L3P_GO2(E, F)                                                                                                      \
    do {                                                                                                                   \
        if (bm64) hipLaunchKernelGGL((gemm_l3p_kernel<E, 64, 4, F>), dim3(gx, cdiv(M, 64)), dim3(256), 0, st, a);          \
        else if (nw8) hipLaunchKernelGGL((gemm_l3p_kernel<E, 128, 8, F>), dim3(gx, cdiv(M, 128)), dim3(512), 0, st, a);    \
        else hipLaunchKernelGGL((gemm_l3p_kernel<E, 128, 4, F>), dim3(gx, cdiv(M, 128)), dim3(256), 0, st, a);             \
    } while (0)
#define L3P_GO(E)                                   \
    do {                                            \
        if (format == LT_F16X2) L3P_GO2(E, LT_F16X2); \
        else L3P_GO2(E, LT_BF16X3);                 \
    } while (0)
    if (geglu) L3P_GO(SCULPT_EPI_GEGLU);
    else if (epilogue == SCULPT_EPI_GELU) L3P_GO(SCULPT_EPI_GELU);
    else if (epilogue == SCULPT_EPI_RELU) L3P_GO(SCULPT_EPI_RELU);
    else L3P_GO(SCULPT_EPI_NONE);
#undef L3P_GO
#undef L3P_GO2
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
