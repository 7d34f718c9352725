// bf16 MFMA GEMM with fused epilogues for the TripoSR transformer stack on gfx950.
//
// Replaces every nn.Linear of the hot path (reference file:line):
//   Attention.to_q/to_k/to_v/to_out       TripoSR/tsr/models/transformer/attention.py:194-206
//   FeedForward / GEGLU                   TripoSR/tsr/models/transformer/basic_transformer_block.py:209-315
//   Transformer1D.proj_in / proj_out      TripoSR/tsr/models/transformer/transformer_1d.py:88,120
//   HF ViT query/key/value/dense/MLP, patch-embedding conv (as a GEMM over 16x16 patches)
//   TriplaneUpsampleNetwork (ConvTranspose2d k2 s2 == GEMM + pixel interleave)  network_utils.py:20-32
//
// out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] ) (+ residual[m][n])
//
// Tiling: 128 activation rows x BW (128 or 64) weight rows per 256-thread workgroup (4 waves as
// 2x2), BK = 64, v_mfma_f32_16x16x32_bf16.  The WEIGHT tile is the MFMA A operand and the ACTIVATION
// tile the B operand, so a lane ends up holding 4 consecutive output columns n of one row m: bias /
// GEGLU / residual are per-lane vector ops and the store is one 16-byte (fp32) or 8-byte (bf16) write.
// Both operands are K-contiguous in HBM = the fragment shape (8 consecutive k per lane).
// Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 3-stage LDS ring with two
// K-tiles in flight (counted s_waitcnt vmcnt + raw s_barrier, one barrier per K-tile): these GEMMs are
// small (M <= 3072, ~1.5 workgroups per CU), so the load latency has to be hidden inside the workgroup.  The LDS image is
// lane-linear per wave instruction (8 rows x 128 B); the (row>>1)&7 chunk XOR that makes every
// ds_read_b128 lane group cover all 64 banks is applied to the per-lane SOURCE address and to the
// read address (guide rule 21).
// The 64-row weight tile is chosen when the 128x128 grid would leave CUs idle (N=1024, M=3072 is
// only 192 tiles for 256 CUs).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace sculpt {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

static constexpr int BM_DEFAULT = 128;   // activation rows per block (64 for launches of fewer tiles than CUs)
#ifndef SCULPT_GEMM_READ_AHEAD
#define SCULPT_GEMM_READ_AHEAD 1
#endif
static constexpr int BK = 64;

// GELU(erf).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, below fp32 noise of the GEMM that
// feeds it): ~12 VALU + v_exp + v_rcp instead of libm erff's ~50 instructions -- the GEGLU epilogue
// evaluates 32 of these per thread and was ~30 % of the FF1 kernel.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// byte offset of 16-byte chunk c (0..7) of row r in a [rows][64 bf16] LDS tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

struct GemmArgs {
    const uint16_t *A; int lda;
    const uint16_t *W; int ldw;
    const float *bias;
    const float *residual; int ldr;
    float *out_f32; uint16_t *out_bf16; int ldo;
    uint16_t *out_t; int ldt;
    int M, N, K;
    int n_split;  // columns >= n_split go ONLY to out_t (row n - n_split); columns < n_split skip out_t
    int n_major;  // XCD chunks run over all activation-row tiles of a few weight tiles (W larger than A) or the reverse
    int n_store;  // only columns n < n_store are written (N is padded to the tile; the output slice may be narrower)
    // implicit 3x3 convolution (CONV): A = channel-last activations [images][H*W][lda], row m = output pixel, K-tile kt =
    // 64 channels cc of tap (ky, kx): source pixel (y + (ky-1) d, x + (kx-1) d) or a zero page outside the image
    int cv_H, cv_W, cv_c64, cv_dil;
    const uint16_t *cv_zero;
    // LayerNorm folded into this GEMM (sculpt_gemm_bf16_ln): A holds the UN-normalised rows x, W has gamma folded in, bias has
    // beta folded in; ln_stats = per-row (mean, M2) of the 32-column slices of x written by the GEMM that produced x:
    //   out = rstd[m] * (acc[m][n] - mean[m] * ln_colsum[n]) + bias[n]
    const float *ln_stats; int ln_slots;  // [ln_slots][stats_ld][2], slot-major
    const float *ln_colsum;
    float ln_eps;
    float *stats_out;  // producer side: per-row (mean, M2) of every 32-column slice of the fp32 result, [N/32][stats_ld][2]
    int stats_ld;      // rows per slice plane of both statistics arrays (>= M)
    const float *zeros;  // >= 2N zero floats: stands in for a missing bias / colsum so the epilogue loads are unconditional
    // Tile order inside an XCD's chunk (0: the 1-D band order chosen by n_major).  gm > 0: tiles are walked in groups of gm
    // activation-row tiles, weight tile outer, row tile inner (the classic grouped order), so the R workgroups an XCD has
    // resident at any time form a gm x R/gm block of the tile grid and stream gm row tiles + R/gm weight tiles through its L2
    // per round, not one row tile + every weight tile (FF1 of a 4-image batch: 6 MB instead of 17 MB per XCD and round).
    int gm;
    // rows m < m_store are written (= M in every product launch).  SCULPT_GEMM_DBG_NOSTORE=1 sets 0: the same loads, K loop and
    // epilogue arithmetic with no output stores at all -- what the stores of an epilogue cost (timing only, tools/gemm_store_cost.py)
    int m_store;
    // gemm256_kernel: 1 = the epilogue stages its bf16 tile in the (now free) LDS ring and writes it out in whole rows -- 16 bytes
    // per lane, 256 / 512 contiguous bytes per tile row -- instead of 8 bytes per lane in 32-byte row segments straight from the
    // MFMA accumulator layout (four partial writes per 128-byte line: the stores were 22-32 % of these launches,
    // tools/gemm_store_cost.py); the transposed (V^T) part of a column-split launch is staged transposed and written the same way
    // instead of as 2-byte scalars.  Set by the launcher when every tile is entirely token-major or entirely transposed.
    int stage;
    // -DSCULPT_EXPERIMENTS builds only (tools/gemm_timeline.py): per workgroup 16 words -- s_memrealtime (100 MHz) at entry, first
    // K-tile landed, K loop done, epilogue arithmetic done, exit; HW_ID; XCC_ID; [8..12] s_memtime (shader clock) at the same points.  nullptr in every product launch.
    unsigned long long *stamps;
};

#ifdef SCULPT_EXPERIMENTS
unsigned long long *g_gemm_stamps = nullptr;   // (declared in common.h with the GEMM_STAMP macros)
extern "C" void sculpt_experiment_gemm_stamps(void *p) { g_gemm_stamps = static_cast<unsigned long long *>(p); }
#endif

// (n tile, m tile) of workgroup-linear index `lin` in a grid of gx weight tiles x gy activation-row tiles
__device__ __forceinline__ void gemm_tile_of(const GemmArgs &g, int lin, int gx, int gy, int &nt, int &mt) {
    const int tile = xcd_tile(lin, gx * gy);
    if (g.gm > 0) {
        const int per = g.gm * gx, grp = tile / per, first = grp * g.gm, within = tile - grp * per;
        const int gsz = min(g.gm, gy - first);
        nt = within / gsz;
        mt = first + (within - nt * gsz);
    } else {
        nt = g.n_major ? tile / gy : tile % gx;
        mt = g.n_major ? tile % gy : tile / gx;
    }
}

static constexpr int LN_SLOT = 64;  // columns per statistics slice (a BW=64 tile; one wave of a BW=128 tile)

// EPI: epilogue; BW: weight rows per block (GEGLU: 128 weight rows = 64 value + 64 gate columns)
// NW: waves per workgroup.  4 = 2x2 waves of 64 activation x BW/2 weight rows; 8 = 2 (weight) x 4 (activation)
// waves of 32 x BW/2: twice the waves per SIMD to hide LDS / barrier latency, at 1.5x the LDS bytes per MFMA.
// NWR: waves along the weight rows (2, or 4 for the 96 x 128 one-round tile: 4 x 2 waves of 32 weight x 48 activation rows).
// KS: the two waves (wr = 0, 1) that share a band of activation rows split the K-tile between them instead of the weight rows --
// wave (wr, wc) multiplies k-step wr of every K-tile into ALL BW weight rows x its activation rows (2 TI x TJ accumulator tiles,
// TI + ... fragments: (2 TI + TJ) reads per 2 TI TJ MFMAs where the weight-row split reads 2 (TI + TJ)), and after the K loop the
// pair exchanges halves through the (free) LDS ring, so that every wave ends with the sums of its usual TI x TJ tiles and the
// epilogue is unchanged.  Same MFMAs, 30 % fewer LDS fragment bytes (192 x 64 tile: 7 instead of 10 reads per K-tile and wave):
// these K loops are bound by LDS read bandwidth (DESIGN_HISTORY.md, transformer stack, round 5).  The sum is (even k-steps) + (odd k-steps): not the
// k order of the other tile forms, i.e. equal to them to fp32 rounding, not bit for bit.
template <int EPI, int BW, int NW, bool CONV = false, int BM = BM_DEFAULT, int NWR = 2, bool KS = false>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_kernel(GemmArgs g) {
    static_assert(NWR == 2 || (EPI == SCULPT_EPI_NONE && !CONV), "the 4 x 2 wave grid is for the plain / residual / LayerNorm-fold form");
    static_assert(!KS || (NWR == 2 && NW == 8 && EPI == SCULPT_EPI_NONE && !CONV), "the k-split pairs are for the 8-wave plain form");
    constexpr int WT = BW * 128;  // bytes of a weight tile
    constexpr int AT = BM * 128;
    // LDS ring depth: 3 stages (two K-tiles in flight) for the 64-row tile, 2 for the 128-row tile --
    // either way 64-72 KiB, i.e. two workgroups per CU (a third stage at 96 KiB measured slower: one
    // workgroup per CU leaves the epilogue and the ramp-up uncovered).
    // (A 256 x 128 tile -- one 144-KiB workgroup per CU, 25 % fewer L2 -> LDS bytes per flop -- measured slower on every shape
    // of the two transformers, as did a 96 x 64 tile inside the pipeline: DESIGN.md 3.4, DESIGN_HISTORY.md.)
    constexpr int NSTAGE = (BW == 64 || BM == 96) ? 3 : 2;
    constexpr int DIST = NSTAGE - 1;  // prefetch distance in K-tiles
    // [stage][W | A], then 2 KiB of exchange space for the LayerNorm statistics (ONE shared array: a second __shared__ object
    // beside the LDS-DMA ring can make hipcc drain the DMA queue before every fragment read)
    constexpr int XCH = NSTAGE * (WT + AT);
    __shared__ __attribute__((aligned(16))) unsigned char smem[XCH + 2 * BM * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NWC = NW / NWR;        // waves along the activation rows
    constexpr int WPW = BW / NWR;        // weight rows (output columns) per wave
    constexpr int TJ = BM / 16 / NWC;    // 16-row activation sub-tiles per wave (4 or 2)
    const int wr = wave / NWC, wc = wave % NWC;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? BW / 2 : BW;  // output columns per block
    constexpr int TI = WPW / 16;                                    // 16-row weight sub-tiles per wave
    // XCD-aware order: the workgroups of one XCD (private L2) take a contiguous band of the tile grid -- a band of
    // activation rows with every weight tile, or (n_major, when W is the larger operand) a band of weight rows with
    // every activation tile -- so the band's panel is fetched into that L2 once and only the smaller operand streams.
    int nt_, mt_;
    gemm_tile_of(g, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, nt_, mt_);
    const int n0 = nt_ * NOUT, m0 = mt_ * BM;
    GEMM_STAMP(g, 0);

    // ---- staging addresses.  One wave instruction fills 8 tile rows (1 KiB).  Lane l of the
    // instruction that fills rows 8q..8q+7 writes LDS chunk (row = 8q + l/8, slot = l%8) and must
    // therefore READ source chunk slot ^ ((row>>1)&7) of that row.
    const int srow = lane >> 3, sslot = lane & 7;
    constexpr int WI = BW / 8 / NW;  // wave instructions per wave for the weight tile (BW/8 rows-of-8 over NW waves)
    // ... and for the activation tile: AG row groups of 8 dealt to the waves -- evenly, or (96 rows on 8 waves: 12 groups) the
    // first AFULL waves take AI groups and the others AI - 1
    constexpr int AG = BM / 8, AI = (AG + NW - 1) / NW, AFULL = AG - NW * (AI - 1);
    constexpr bool UNEVEN = AG % NW != 0;
    const int acnt = (!UNEVEN || wave < AFULL) ? AI : AI - 1;                                   // wave-uniform
    const int abase = (!UNEVEN || wave < AFULL) ? wave * AI : AFULL * AI + (wave - AFULL) * (AI - 1);
    const uint16_t *wsrc0, *wsrc1, *wsrc2, *wsrc3;
    const uint16_t *asrc0, *asrc1, *asrc2, *asrc3;
    unsigned amask0 = 0, amask1 = 0, amask2 = 0, amask3 = 0;  // CONV: bit t = tap t of this staged row is inside the image
    const uint16_t *zsrc = CONV ? g.cv_zero + (sslot << 3) : nullptr;
    {
        auto wrow = [&](int j) -> int {
            if (EPI == SCULPT_EPI_GEGLU) {
                const int sub = j >> 4, within = j & 15;
                return ((sub & 1) ? g.N : 0) + n0 + (sub >> 1) * 16 + within;
            }
            return n0 + j;
        };
        auto wp = [&](int q) -> const uint16_t * {
            const int r = 8 * (wave * WI + q) + srow;
            return g.W + (long)wrow(r) * g.ldw + ((sslot ^ ((r >> 1) & 7)) << 3);
        };
        auto ap = [&](int q) -> const uint16_t * {
            const int r = 8 * (abase + q) + srow;
            const int m = min(m0 + r, g.M - 1);
            return g.A + (long)m * g.lda + ((sslot ^ ((r >> 1) & 7)) << 3);
        };
        auto am = [&](int q) -> unsigned {
            if (!CONV) return 0u;
            const int r = 8 * (abase + q) + srow;
            const int m = min(m0 + r, g.M - 1);
            const int x = m % g.cv_W, y = (m / g.cv_W) % g.cv_H;
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + (t / 3 - 1) * g.cv_dil, xx = x + (t % 3 - 1) * g.cv_dil;
                mask |= (yy >= 0 && yy < g.cv_H && xx >= 0 && xx < g.cv_W) ? (1u << t) : 0u;
            }
            return mask;
        };
        wsrc0 = wp(0); wsrc1 = wp(1 % WI); wsrc2 = wp(2 % WI); wsrc3 = wp(3 % WI);
        asrc0 = ap(0); asrc1 = ap(1 % AI); asrc2 = ap(2 % AI); asrc3 = ap(3 % AI);
        amask0 = am(0); amask1 = am(1 % AI); amask2 = am(2 % AI); amask3 = am(3 % AI);
    }
    const int wdst = (wave * WI) * 1024, adst = abase * 1024;  // wave-uniform LDS byte offsets

#define STAGE(buf, kt)                                                                                       \
    do {                                                                                                     \
        unsigned char *wb = smem + (buf) * (WT + AT);                                                        \
        unsigned char *ab = wb + WT;                                                                         \
        const int ko = (kt) * BK;                                                                            \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc0 + ko), (lds_ptr_t)(wb + wdst), 16, 0, 0);         \
        if (WI >= 2)                                                                                         \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc1 + ko), (lds_ptr_t)(wb + wdst + 1024), 16, 0, 0); \
        if (WI == 4) {                                                                                       \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc2 + ko), (lds_ptr_t)(wb + wdst + 2048), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc3 + ko), (lds_ptr_t)(wb + wdst + 3072), 16, 0, 0); \
        }                                                                                                    \
        long ao = ko;                                                                                        \
        int tap = 0;                                                                                         \
        if (CONV) {                                                                                          \
            tap = (kt) / g.cv_c64;                                                                           \
            const int cc = (kt) - tap * g.cv_c64, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;            \
            ao = ((long)(dy * g.cv_dil) * g.cv_W + dx * g.cv_dil) * g.lda + cc * 64;                         \
        }                                                                                                    \
        const uint16_t *a0 = (!CONV || ((amask0 >> tap) & 1u)) ? asrc0 + ao : zsrc;                          \
        const uint16_t *a1 = (!CONV || ((amask1 >> tap) & 1u)) ? asrc1 + ao : zsrc;                          \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)a0, (lds_ptr_t)(ab + adst), 16, 0, 0);                   \
        if (AI >= 2 && (!UNEVEN || acnt >= 2))                                                               \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)a1, (lds_ptr_t)(ab + adst + 1024), 16, 0, 0);        \
        if (AI >= 3) {                                                                                       \
            const uint16_t *a2 = (!CONV || ((amask2 >> tap) & 1u)) ? asrc2 + ao : zsrc;                      \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)a2, (lds_ptr_t)(ab + adst + 2048), 16, 0, 0);        \
        }                                                                                                    \
        if (AI == 4) {                                                                                       \
            const uint16_t *a3 = (!CONV || ((amask3 >> tap) & 1u)) ? asrc3 + ao : zsrc;                      \
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)a3, (lds_ptr_t)(ab + adst + 3072), 16, 0, 0);        \
        }                                                                                                    \
    } while (0)

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[KS ? 2 * TI : 1][KS ? TJ : 1];   // KS: partial sums of k-step wr over both weight halves
    if (KS) {
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc2[KS ? i : 0][KS ? j : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nk = g.K / BK;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read offsets (bytes) for ks = 0; ks = 1 flips chunk bit 2 -> XOR 64 bytes
    int aoff[TI], boff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) aoff[i] = lds_off(wr * WPW + i * 16 + fr, fq);
    int aoff2[KS ? 2 * TI : 1];   // KS: fragments of all BW weight rows, k-step wr (chunk bit 2 = the k-step: XOR 64 bytes)
    if (KS) {
#pragma unroll
        for (int i = 0; i < 2 * TI; ++i) aoff2[KS ? i : 0] = lds_off((i / TI) * WPW + (i % TI) * 16 + fr, fq) ^ (wr << 6);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) boff[j] = lds_off(wc * (16 * TJ) + j * 16 + fr, fq);

    // LayerNorm fold: the row's slice statistics (every slice has LN_SLOT columns) are combined into (mean, rstd) with Chan's
    // parallel-variance formula.  Row m of sub-tile j sits on the four lanes fr, fr+16, fr+32, fr+48: lane fq takes slices fq,
    // fq+4, ...  Statistics are slot-major [slot][row]: the 16 rows of a sub-tile are 128 contiguous bytes per slice (row-major
    // cost 16 partial cache lines per wave instruction and 3x the time).
    // The loads are issued HERE, ahead of everything, and consumed only AFTER the K loop: under a saturated L2 -> LDS stream a
    // load takes 2-3 us to come back, and consumed in front of the loop that latency was paid by every workgroup round
    // (+3 / +4 / +7 us on the 384 / 576 / 1536-workgroup launches).  Register loads and LDS-DMA were observed to retire out of
    // issue order relative to each other (a counted `s_waitcnt vmcnt(LPT)` behind these loads let wrong rows through in ~1 of
    // 3 launches), so the first K-tile is waited for with vmcnt(0) while they may still be in flight (below).  Slices beyond
    // ln_slots are read clamped and masked out.
    constexpr int MAXU = 4;  // up to 16 slices = 1024 normalised columns
    float2 sv[TJ][MAXU];
    // the two waves (wr = 0, 1) that share these rows would load the same statistics: only wr = 0 loads and computes, and
    // hands (mean, rstd) to its partner through LDS after the K loop -- these launches are bound by L2 -> CU bytes, and the
    // statistics re-read by every column tile were +17 % of them with 32-column slices loaded by both waves
    const bool ln_load = g.ln_stats && wr == 0;
    if (ln_load) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = min(m0 + wc * (16 * TJ) + j * 16 + fr, g.M - 1);
            const float2 *sp = reinterpret_cast<const float2 *>(g.ln_stats) + m;
#pragma unroll
            for (int u = 0; u < MAXU; ++u) sv[j][u] = sp[(long)min(fq + 4 * u, g.ln_slots - 1) * g.stats_ld];
        }
    }

    // 3-stage ring, two K-tiles in flight: wait for tile kt only (counted vmcnt), one raw barrier per
    // K-tile (a __syncthreads() here would drain the LDS-DMA queue: guide "Pipelining across barriers").
    constexpr int LPT = WI + AI;  // LDS-DMA instructions per thread per K-tile
    STAGE(0, 0);
    if (DIST > 1 && nk > 1) STAGE(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NSTAGE;
        // wait until tile kt has landed; tiles kt+1 .. kt+DIST-1 (if issued) stay in flight
        if (DIST > 1 && kt + 1 < nk && !(kt == 0 && g.ln_stats)) {
            if (!UNEVEN || acnt == AI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT - 1) : "memory");   // this wave issued one LDS-DMA fewer per K-tile
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#ifdef SCULPT_EXPERIMENTS
        if (kt == 0) GEMM_STAMP(g, 1);
#endif
        // every wave finished reading stage (kt-1)%NSTAGE == (kt+DIST)%NSTAGE before it passed the barrier
        if (kt + DIST < nk) STAGE((kt + DIST) % NSTAGE, kt + DIST);
        const unsigned char *wb = smem + buf * (WT + AT);
        const unsigned char *ab = wb + WT;
        if constexpr (KS) {
            bf16x8_t af[2 * TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < 2 * TI; ++i) af[i] = *reinterpret_cast<const bf16x8_t *>(wb + aoff2[i]);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t *>(ab + (boff[j] ^ (wr << 6)));
#pragma unroll
            for (int i = 0; i < 2 * TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc2[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TI + TJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2 * TI * TJ, 0);
        } else if ((NW == 8 || (NW == 4 && BM == 96)) && SCULPT_GEMM_READ_AHEAD) {
            // both k-steps' fragments are requested before the first MFMA: one exposed LDS latency per K-tile instead of one
            // per group of four MFMAs (the register budget of the 8-wave tiles allows the 2 (TI + TJ) fragments)
            bf16x8_t af[2][TI], bfr[2][TJ];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TI; ++i) af[ks][i] = *reinterpret_cast<const bf16x8_t *>(wb + (aoff[i] ^ (ks << 6)));
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8_t *>(ab + (boff[j] ^ (ks << 6)));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
            // keep the request order: all 2 (TI + TJ) LDS reads, then the MFMAs (mask 0x100 = DS read, 0x8 = MFMA)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TI + TJ), 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2 * TI * TJ, 0);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8_t *>(wb + (aoff[i] ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t *>(ab + (boff[j] ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        }
    }
#undef STAGE
    if constexpr (KS) {
        // the pair (wr = 0, 1) of a band swaps halves: wave wr keeps the tiles of ITS weight rows (i2 = wr TI + i) and adds the
        // partner's partial sums of them.  Exchange buffer = the LDS ring, free once every wave is past its last fragment read.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 *xb = reinterpret_cast<f32x4 *>(smem);
        const int partner = (1 - wr) * NWC + wc;
        static_assert(NW * TI * TJ * 64 * 16 <= NSTAGE * (WT + AT), "the exchange fits the ring");
        if (wr == 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) xb[(wave * TI * TJ + i * TJ + j) * 64 + lane] = acc2[TI + i][j];
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) xb[(wave * TI * TJ + i * TJ + j) * 64 + lane] = acc2[i][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = acc2[i][j] + xb[(partner * TI * TJ + i * TJ + j) * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = xb[(partner * TI * TJ + i * TJ + j) * 64 + lane] + acc2[TI + i][j];
        }
    }

    GEMM_STAMP(g, 2);
    float ln_mean[TJ], ln_rstd[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 1.f; }
    float2 *xch = reinterpret_cast<float2 *>(smem + XCH);  // [wr][BM rows]
    if (g.ln_stats) {
        if (ln_load) {
            const float inv_slots = 1.0f / (float)g.ln_slots;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float sm = 0.f;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) sm += (fq + 4 * u < g.ln_slots) ? sv[j][u].x : 0.f;
                sm += __shfl_xor(sm, 16, 64);
                sm += __shfl_xor(sm, 32, 64);
                const float mean = sm * inv_slots;
                float m2 = 0.f;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const float d = sv[j][u].x - mean;
                    m2 += (fq + 4 * u < g.ln_slots) ? fmaf((float)LN_SLOT * d, d, sv[j][u].y) : 0.f;
                }
                m2 += __shfl_xor(m2, 16, 64);
                m2 += __shfl_xor(m2, 32, 64);
                ln_mean[j] = mean;
                ln_rstd[j] = rsqrtf(m2 * inv_slots * (1.0f / LN_SLOT) + g.ln_eps);
                if (fq == 0) xch[wc * (16 * TJ) + j * 16 + fr] = make_float2(ln_mean[j], ln_rstd[j]);
            }
        }
        // (the exchange space is not part of the staging ring: no barrier needed before the writes; raw barrier -- a
        // __syncthreads() would also wait for vector memory)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!ln_load) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float2 v = xch[wc * (16 * TJ) + j * 16 + fr];
                ln_mean[j] = v.x;
                ln_rstd[j] = v.y;
            }
        }
    }

    // epilogue: acc[i][j][r] = out[m = m0 + wc*16*TJ + j*16 + fr][tile row = wr*(BW/2) + i*16 + fq*4 + r]
    // Per-column vectors (bias, LayerNorm column sums) depend on i only: ONE unconditional float4 load each per sub-tile, all
    // issued together (a per-element `ptr ? ptr[n] : 0` makes hipcc branch around every load and wait for it: 32 dependent
    // L2 round trips per lane).  A missing vector reads the zero page.
    const float *biasp = g.bias ? g.bias : g.zeros;
    const float *csp = g.ln_stats ? g.ln_colsum : g.zeros;
    float4 b4[TI], c4[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int wrow = (EPI == SCULPT_EPI_GEGLU) ? ((i & 1) ? g.N : 0) + n0 + (wr * (TI / 2) + (i >> 1)) * 16 + fq * 4
                                                   : n0 + wr * WPW + i * 16 + fq * 4;
        b4[i] = *reinterpret_cast<const float4 *>(biasp + wrow);
        c4[i] = *reinterpret_cast<const float4 *>(csp + wrow);
    }
    auto f4 = [](const float4 &v, int r) -> float { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); };
    // ... and every residual tile of this lane before the first store: vmcnt counts stores too, so a load issued after a
    // store would wait for that store's completion as well (one L2 round trip per sub-tile, serialised)
    float4 rs[TI][TJ];
    if (EPI != SCULPT_EPI_GEGLU && g.residual) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = min(m0 + wc * (16 * TJ) + j * 16 + fr, g.M - 1);
#pragma unroll
            for (int i = 0; i < TI; ++i)
                rs[i][j] = *reinterpret_cast<const float4 *>(g.residual + (long)m * g.ldr + n0 + wr * WPW + i * 16 + fq * 4);
        }
    }
    // phase 1: the whole tile in registers (acc is overwritten by the results): every loaded value is consumed here, so the
    // loads are waited for once, before the first store
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        if (EPI == SCULPT_EPI_GEGLU) {
#pragma unroll
            for (int ip = 0; ip < TI / 2; ++ip)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // mean = 0, rstd = 1, colsum = 0 without the fold: exactly acc + bias
                    const float v = ln_rstd[j] * (acc[2 * ip][j][r] - ln_mean[j] * f4(c4[2 * ip], r)) + f4(b4[2 * ip], r);
                    const float gt = ln_rstd[j] * (acc[2 * ip + 1][j][r] - ln_mean[j] * f4(c4[2 * ip + 1], r)) + f4(b4[2 * ip + 1], r);
                    acc[2 * ip][j][r] = v * gelu_erf(gt);
                }
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ln_rstd[j] * (acc[i][j][r] - ln_mean[j] * f4(c4[i], r)) + f4(b4[i], r);
                    if (EPI == SCULPT_EPI_GELU) v = gelu_erf(v);
                    if (EPI == SCULPT_EPI_RELU) v = fmaxf(v, 0.f);
                    if (g.residual) v += f4(rs[i][j], r);
                    acc[i][j][r] = v;
                }
        }
    }
    GEMM_STAMP(g, 3);
    // phase 2: stores (and the slice statistics of the fp32 result)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int m = m0 + wc * (16 * TJ) + j * 16 + fr;
        if (m >= g.m_store) continue;
        if (EPI == SCULPT_EPI_GEGLU) {
#pragma unroll
            for (int ip = 0; ip < TI / 2; ++ip) {
                const int n = n0 + (wr * (TI / 2) + ip) * 16 + fq * 4;  // output column of r = 0
                const f32x4 o = acc[2 * ip][j];
                if (g.out_bf16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(o[0], o[1]);
                    pk.y = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                }
                if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int n = n0 + wr * WPW + i * 16 + fq * 4;
                if (n >= g.n_store) continue;
                const f32x4 o = acc[i][j];
                // (statistics of the fp32 result: after the store loops, below)
                const bool tpart = n >= g.n_split;  // wave-uniform per sub-tile (n_split is a multiple of 16)
                if (!tpart) {
                    if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
                    if (g.out_bf16) {
                        uint2 pk;
                        pk.x = pack_bf16x2(o[0], o[1]);
                        pk.y = pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                    }
                }
                if (g.out_t && (tpart || g.n_split >= g.N)) {
                    const int nt0 = tpart ? n - g.n_split : n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g.out_t[(long)(nt0 + r) * g.ldt + m] = f32_to_bf16(o[r]);
                }
            }
        }
    }
    // phase 3 (producer side of the LayerNorm fold): per row the (mean, M2) of every LN_SLOT = 64 columns of the fp32 result.
    // A wave holds BW/2 columns of 16 * TJ rows: row m's values sit on the 4 lanes fr + 16 {0..3}, 4 * TI each.
    //   BW = 128: the wave's 64 columns are one slice.   BW = 64: the two waves (wr = 0, 1) that share the rows each reduce
    //   their 32 columns and wr = 0 merges the pair through LDS (Chan: M2 = M2a + M2b + n/2 (mean_a - mean_b)^2).
    if (EPI == SCULPT_EPI_NONE && g.stats_out) {
        constexpr int WCOLS = WPW;  // columns of this wave
        float pm[TJ], pq[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * (1.0f / WCOLS);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean; m2 = fmaf(d, d, m2); }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            pm[j] = mean;
            pq[j] = m2;
        }
        float2 *so = reinterpret_cast<float2 *>(g.stats_out);
        if (WCOLS == LN_SLOT) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = m0 + wc * (16 * TJ) + j * 16 + fr;
                if (fq == 0 && m < g.m_store) so[(long)((n0 + wr * WCOLS) / LN_SLOT) * g.stats_ld + m] = make_float2(pm[j], pq[j]);
            }
        } else {
            static_assert(WCOLS == LN_SLOT || 2 * WCOLS == LN_SLOT, "statistics slice = one or two waves' columns");
            // the waves wr = 2p and 2p + 1 share a slice: the odd one hands its half to the even one (pair p in xch[p][row])
            if ((wr & 1) == 1 && fq == 0) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) xch[(wr >> 1) * BM + wc * (16 * TJ) + j * 16 + fr] = make_float2(pm[j], pq[j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // raw barrier: the output stores above stay in flight
            __builtin_amdgcn_s_barrier();
            if ((wr & 1) == 0 && fq == 0) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int m = m0 + wc * (16 * TJ) + j * 16 + fr;
                    const float2 o = xch[(wr >> 1) * BM + wc * (16 * TJ) + j * 16 + fr];
                    const float d = pm[j] - o.x;
                    if (m < g.m_store) so[(long)((n0 + wr * WCOLS) / LN_SLOT) * g.stats_ld + m] = make_float2(0.5f * (pm[j] + o.x), pq[j] + o.y + (0.5f * WCOLS) * d * d);
                }
            }
        }
    }
    GEMM_STAMP(g, 4);
    GEMM_STAMP_IDS(g);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 along the weight rows x 4 along the activation rows: a wave owns 128 x 64 = 8 x 4 MFMA tiles, 128
// accumulator registers), one workgroup per CU, for the two big LayerNorm-folded projections of a backbone block (the fused
// Q|K|V^T projection and FF1 + GEGLU).  Why a second kernel: the 128-row tiles above move 768 B of LDS fragments per MFMA (wave
// tile 32 x 64) and their K loop runs at the rate at which a CU's LDS can be filled while it is being read (DESIGN_HISTORY.md); a
// 128 x 64 wave tile needs 384 B per MFMA -- but with only two waves per SIMD nothing hides a fragment read or a DMA round
// trip unless the loop does it itself.  So the K-tile (BK = 64) is cut into four phases of 16 MFMAs (one 64 x 32 quadrant of
// the wave tile x K = 64), and the loop is pipelined at FRAGMENT granularity:
//   phase 0  Q(A0, B0)   reads B1 of this K-tile            DMA: units 6, 7 of K-tile kt + 1
//   phase 1  Q(A0, B1)   reads A1 (into the A0 registers as they die)
//            -- lgkmcnt(0) + barrier: every wave has finished reading this K-tile's buffer --
//   phase 2  Q(A1, B1)                                      DMA: units 0 - 2 of K-tile kt + 2 (into the buffer just freed)
//            -- counted vmcnt (never 0 while a later K-tile is in flight) + barrier: K-tile kt + 1 has landed --
//   phase 3  Q(A1, B0)   reads A0, B0 of K-tile kt + 1      DMA: units 3 - 5 of K-tile kt + 2
// (A0 / A1 = the wave's weight sub-tiles 0-3 / 4-7, B0 / B1 = its activation sub-tiles 0-1 / 2-3; a DMA "unit" = one
// global_load_lds_dwordx4 per thread = 64 rows of one operand; 8 units per K-tile.)  A fragment is overwritten by the read
// of its successor right after its last MFMA, so the fragments take 64 registers, not 128; every LDS read is issued at least
// eight MFMAs before its first use, every DMA at least two phases before the barrier that publishes it; two barriers per K-tile.
// The epilogue is the one of the kernel above (TI = 8, TJ = 4): LayerNorm fold, bias, GEGLU, the Q|K / V^T column split.
// ---------------------------------------------------------------------------------------------------------------------
// BM = 256 or 192 activation rows per tile (TJ = 4 or 3 sub-tiles of 16 per wave; with 192 the second activation half B1 is one
// sub-tile and phases 1 / 2 have 8 MFMAs): 3072 rows are 12 tiles of 256 or 16 of 192 -- 512 instead of 384 FF1 tiles, i.e.
// two FULL rounds of 256 CUs instead of one and a half.
// RES (round 4, 192-row tiles only): the residual form -- out_f32 = acc + bias + residual (in place), its bf16 copy and the 64-column
// slice statistics of the fp32 result, i.e. what the 128-row kernel does for the N = 1024 projections that write the residual
// stream.  Dispatched where 192 x 256 tiles give every CU a tile: a 4-image batch (M = 12288, N = 1024 -> 256 tiles), where the
// 128 x 128 tiles (768 of them, 3 per CU) move 1.75x the bytes through L2 -> LDS.
template <int EPI, int BM, bool RES = false>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g) {
    static_assert(BM == 256 || BM == 192, "activation rows per tile");
    static_assert(!RES || (BM == 192 && EPI == SCULPT_EPI_NONE), "the residual form: 192-row tiles, no activation");
    constexpr int BW = 256, NW = 8, NWC = 4, TI = 8, TJ = BM / 64, WROWS = BM / 4;  // WROWS: activation rows per wave (64 / 48)
    constexpr int JB1 = TJ - 2, AU = BM / 64;  // sub-tiles in B1; DMA units of the activation tile (one per 8 rows per wave)
    constexpr int NU = 4 + AU;                 // DMA units per K-tile: 8 or 7
    constexpr int WT = BW * 128, AT = BM * 128;  // 32 KiB, 32 / 24 KiB
    constexpr int XCH = 2 * (WT + AT);           // two K-tile buffers
    constexpr int XBC = XCH + 2 * BM * 8;        // bias and column sums of the tile's 256 weight rows (2 x 1 KiB), see below
    __shared__ __attribute__((aligned(16))) unsigned char smem[XBC + 2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? BW / 2 : BW;
    int nt_, mt_;
    gemm_tile_of(g, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, nt_, mt_);
    const int n0 = nt_ * NOUT, m0 = mt_ * BM;
    GEMM_STAMP(g, 0);

    // staging: wave w fills rows 32 w .. 32 w + 31 of both operand tiles, 8 rows (1 KiB) per wave instruction; the LDS image is
    // lane-linear, the (row >> 1) & 7 chunk swizzle is applied to the per-lane SOURCE address and to the read address
    const int srow = lane >> 3, sslot = lane & 7;
    // per-lane BYTE offsets (32 bits) from the operands' uniform bases: the DMA takes an SGPR base + VGPR offset, 8 registers of
    // addresses instead of 16 (the launcher checks that both operands stay below 4 GiB)
    unsigned woff[4], aof[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 8 * (wave * 4 + q) + srow;
        const int ra = 8 * (wave * AU + (q < AU ? q : AU - 1)) + srow;  // activation tile: AU row groups per wave
        int wrow = n0 + r;
        if (EPI == SCULPT_EPI_GEGLU) {
            const int sub = r >> 4, within = r & 15;  // 16-row sub-tiles alternate value / gate rows
            wrow = ((sub & 1) ? g.N : 0) + n0 + (sub >> 1) * 16 + within;
        }
        woff[q] = ((unsigned)wrow * (unsigned)g.ldw + ((sslot ^ ((r >> 1) & 7)) << 3)) * 2u;
        aof[q] = ((unsigned)min(m0 + ra, g.M - 1) * (unsigned)g.lda + ((sslot ^ ((ra >> 1) & 7)) << 3)) * 2u;
    }
    const char *Wb = reinterpret_cast<const char *>(g.W), *Ab = reinterpret_cast<const char *>(g.A);
    const int sdst = wave * 4096, sdsta = wave * (AU * 1024);  // wave-uniform byte offsets inside the operand tiles
#define G256_STAGE(buf, kt, u)                                                                                                   \
    do {                                                                                                                         \
        unsigned char *tb = smem + (buf) * (WT + AT) + ((u) >= 4 ? WT + sdsta : sdst) + ((u) & 3) * 1024;                        \
        const char *src = ((u) >= 4 ? Ab + aof[(u) & 3] : Wb + woff[(u) & 3]) + (size_t)(kt) * (BK * 2);                         \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)tb, 16, 0, 0);                                               \
    } while (0)

    const int nk = g.K / BK;
    // prologue: K-tile 0 entirely, units 0 - 5 of K-tile 1 (its last one or two are issued by phase 0 of K-tile 0 like everyone's)
#pragma unroll
    for (int u = 0; u < NU; ++u) G256_STAGE(0, 0, u);
    if (nk > 1) {
#pragma unroll
        for (int u = 0; u < 6; ++u) G256_STAGE(1, 1, u);
    }

    const int fr = lane & 15, fq = lane >> 4;
    // bias and LayerNorm column sums of the tile's weight rows -> LDS, now: loaded in the epilogue they were a chain of eight (GEGLU:
    // four) dependent round trips to L2 per lane that nothing overlapped -- 3-4 us of a 6 us epilogue (tools/gemm_timeline.py).
    // Order in LDS = the order the epilogue walks: plain, tile rows 0..255; GEGLU, the 128 value rows, then the 128 gate rows.
    if (tid < 128) {
        const float *srcv = tid < 64 ? (g.bias ? g.bias : g.zeros) : (g.ln_stats ? g.ln_colsum : g.zeros);
        const int e = (tid & 63) * 4;
        const int row = EPI == SCULPT_EPI_GEGLU ? (e < 128 ? n0 + e : g.N + n0 + e - 128) : n0 + e;
        *reinterpret_cast<float4 *>(smem + XBC + (tid >> 6) * 1024 + e * 4) = *reinterpret_cast<const float4 *>(srcv + row);
    }
    // LayerNorm fold: (mean, rstd) of the wave's 64 activation rows from the producer's slice statistics, BEFORE the K loop here
    // (the 16 float2 per lane the other kernel carries across its loop do not fit beside 128 accumulators); wr = 0 computes,
    // wr = 1 receives through LDS.  These plain loads make hipcc wait vmcnt(0), which also waits for the prologue's DMA: wanted.
    float ln_mean[TJ], ln_rstd[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 1.f; }
    float2 *xch = reinterpret_cast<float2 *>(smem + XCH);  // [BM rows]
    if (g.ln_stats) {
        if (wr == 0) {
            constexpr int MAXU = 4;
            const float inv_slots = 1.0f / (float)g.ln_slots;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = min(m0 + wc * WROWS + j * 16 + fr, g.M - 1);
                const float2 *sp = reinterpret_cast<const float2 *>(g.ln_stats) + m;
                float2 sv[MAXU];
#pragma unroll
                for (int u = 0; u < MAXU; ++u) sv[u] = sp[(long)min(fq + 4 * u, g.ln_slots - 1) * g.stats_ld];
                float sm = 0.f;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) sm += (fq + 4 * u < g.ln_slots) ? sv[u].x : 0.f;
                sm += __shfl_xor(sm, 16, 64);
                sm += __shfl_xor(sm, 32, 64);
                const float mean = sm * inv_slots;
                float m2 = 0.f;
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const float d = sv[u].x - mean;
                    m2 += (fq + 4 * u < g.ln_slots) ? fmaf((float)LN_SLOT * d, d, sv[u].y) : 0.f;
                }
                m2 += __shfl_xor(m2, 16, 64);
                m2 += __shfl_xor(m2, 32, 64);
                ln_mean[j] = mean;
                ln_rstd[j] = rsqrtf(m2 * inv_slots * (1.0f / LN_SLOT) + g.ln_eps);
                if (fq == 0) xch[wc * WROWS + j * 16 + fr] = make_float2(ln_mean[j], ln_rstd[j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float2 v = xch[wc * WROWS + j * 16 + fr];
                ln_mean[j] = v.x;
                ln_rstd[j] = v.y;
            }
        }
    }

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) of sub-tiles 0-3 (A) / 0-1 (B) for ks = 0; the second half of the wave tile is + 64 rows
    // (A: + 8192 B) / + 32 rows (B: + 4096 B) -- a multiple of 16 rows leaves the swizzle term unchanged; ks = 1 is ^ 64
    int aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = lds_off(wr * 128 + i * 16 + fr, fq);
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = WT + lds_off(wc * WROWS + j * 16 + fr, fq);
    bf16x8_t af[4][2], bq[4][2];  // af: the current A half; bq[0..1]: B0, bq[2..3]: B1
#define G256_LDA(buf, ih, i, ks) af[i][ks] = *reinterpret_cast<const bf16x8_t *>(smem + (buf) * (WT + AT) + (ih) * 8192 + (aoff[i] ^ ((ks) << 6)))
#define G256_LDB(buf, jh, j, ks) bq[2 * (jh) + (j)][ks] = *reinterpret_cast<const bf16x8_t *>(smem + (buf) * (WT + AT) + (jh) * 4096 + (boff[j] ^ ((ks) << 6)))
#define G256_MF(ih, jh, i, j, ks) \
    acc[4 * (ih) + (i)][2 * (jh) + (j)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bq[2 * (jh) + (j)][ks], acc[4 * (ih) + (i)][2 * (jh) + (j)], 0, 0, 0)
#define G256_FENCE __builtin_amdgcn_sched_barrier(0)

    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    GEMM_STAMP(g, 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) G256_LDA(0, 0, i, ks);
#pragma unroll
        for (int j = 0; j < 2; ++j) G256_LDB(0, 0, j, ks);
    }

    // One K-tile.  MORE1 / MORE2: K-tiles kt + 1 / kt + 2 exist -- compile-time, so that the steady-state body is ONE basic
    // block (a run-time test around every DMA and fragment read cut the phases into a dozen blocks with a branch each)
    auto ktile = [&](int kt, auto more1_t, auto more2_t) __attribute__((always_inline)) {
        constexpr bool MORE1 = decltype(more1_t)::value, MORE2 = decltype(more2_t)::value;
        const int buf = kt & 1, nbuf = buf ^ 1;
        // ---- phase 0: Q(A0, B0); read B1 of this K-tile; DMA units 6, 7 of K-tile kt + 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                G256_MF(0, 0, i, 0, ks);
                G256_MF(0, 0, i, 1, ks);
                if (i < JB1) G256_LDB(buf, 1, i, ks);
                if (ks == 0 && i >= 2 && 4 + i < NU && MORE1) G256_STAGE(nbuf, kt + 1, 4 + i);
                G256_FENCE;
            }
        // ---- phase 1: Q(A0, B1); every A0 fragment is replaced by its A1 successor right after its last MFMA
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                G256_MF(0, 1, i, 0, ks);
                if (JB1 == 2) G256_MF(0, 1, i, 1, ks);
                G256_LDA(buf, 1, i, ks);
                G256_FENCE;
            }
        // every wave is done reading buffer `buf`: it may be refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: Q(A1, B1); DMA units 0 - 2 of K-tile kt + 2 into `buf`
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                G256_MF(1, 1, i, 0, ks);
                if (JB1 == 2) G256_MF(1, 1, i, 1, ks);
                if (ks == 0 && i < 3 && MORE2) G256_STAGE(buf, kt + 2, i);
                G256_FENCE;
            }
        // K-tile kt + 1 has landed: this wave's pieces by the counted wait (the three DMAs just issued stay in flight),
        // everybody's by the barrier
        if (MORE2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: Q(A1, B0); read A0 and B0 of K-tile kt + 1; DMA units 3 - 5 of K-tile kt + 2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                G256_MF(1, 0, i, 0, ks);
                G256_MF(1, 0, i, 1, ks);
                if (MORE1) G256_LDA(nbuf, 0, i, ks);
                if (MORE1 && i == 3) {  // the B0 fragments of this k-step are dead now
                    G256_LDB(nbuf, 0, 0, ks);
                    G256_LDB(nbuf, 0, 1, ks);
                }
                if (ks == 0 && i < 3 && MORE2) G256_STAGE(buf, kt + 2, 3 + i);
                G256_FENCE;
            }
    };
    {
        int kt = 0;
        for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); ++kt; }
        ktile(kt, std::false_type{}, std::false_type{});
    }
    GEMM_STAMP(g, 2);
#undef G256_STAGE
#undef G256_LDA
#undef G256_LDB
#undef G256_MF
#undef G256_FENCE

    // ---- epilogue (as in gemm_bf16_kernel): acc[i][j][r] = out[m = m0 + wc*64 + j*16 + fr][tile row = wr*128 + i*16 + fq*4 + r]
    // staged stores (GemmArgs::stage): the K loop's last LDS reads are behind its mid-tile barrier, so the ring is free here
    constexpr int RSB = NOUT * 2 + 16;   // staged row: NOUT bf16 + 16 bytes (rows 4 banks apart: 2-way at worst on the 8-byte writes)
    constexpr int TSB = BM * 2 + 16;     // transposed staging: a row = the BM activation rows of one weight row
    static_assert(BM * RSB <= XCH + 2 * BM * 8 && 256 * TSB <= XCH + 2 * BM * 8, "the staged tile fits the LDS of the ring");
    const bool staged = !RES && g.stage != 0;
    // token-major staging: a lane writes the 8-byte piece of its column group fq of tile row fr; rows are 68 (132) dwords apart,
    // so the 16 lanes of one fq group -- one 128-byte pass of a ds_write_b64 -- hit banks 4 fr (mod 32): rows fr and fr + 8 collide
    // (the 7-9 % conflict cycles of the counters, gone with direct stores).  Rows 8..15 of every 16 therefore swap their even /
    // odd pieces (fq ^ 1): the 16 lanes then cover 32 distinct banks; the 16-byte read-out swaps the halves back for those rows.
    const int fqs = fq ^ (fr >> 3);
    const bool tileT = staged && EPI != SCULPT_EPI_GEGLU && g.out_t && n0 >= g.n_split;   // workgroup-uniform
    const float4 *bias_s = reinterpret_cast<const float4 *>(smem + XBC), *cs_s = reinterpret_cast<const float4 *>(smem + XBC + 1024);
    auto f4 = [](const float4 &v, int r) -> float { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); };
    if (EPI == SCULPT_EPI_GEGLU) {
#pragma unroll
        for (int ip = 0; ip < TI / 2; ++ip) {
            const int wv = n0 + (wr * (TI / 2) + ip) * 16 + fq * 4;  // value row; the gate row is N further
            const int li = (wr * (TI / 2) + ip) * 16 + fq * 4;   // = wv - n0
            const float4 bv = bias_s[li >> 2], bg = bias_s[(128 + li) >> 2], cv = cs_s[li >> 2], cg = cs_s[(128 + li) >> 2];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = m0 + wc * WROWS + j * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ln_rstd[j] * (acc[2 * ip][j][r] - ln_mean[j] * f4(cv, r)) + f4(bv, r);
                    const float gt = ln_rstd[j] * (acc[2 * ip + 1][j][r] - ln_mean[j] * f4(cg, r)) + f4(bg, r);
                    o[r] = v * gelu_erf(gt);
                }
                if (staged) {
                    uint2 pk;
                    pk.x = pack_bf16x2(o[0], o[1]);
                    pk.y = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<uint2 *>(smem + (wc * WROWS + j * 16 + fr) * RSB + ((wr * (TI / 2) + ip) * 16 + fqs * 4) * 2) = pk;
                } else if (m < g.m_store) {
                    if (g.out_bf16) {
                        uint2 pk;
                        pk.x = pack_bf16x2(o[0], o[1]);
                        pk.y = pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + wv) = pk;
                    }
                    if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + wv) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else if (RES) {
        // every residual tile of this lane before the first store (vmcnt counts stores: a load behind a store waits for it too);
        // the 64 fragment registers of the K loop are free now
        float4 rs[TI][TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = min(m0 + wc * WROWS + j * 16 + fr, g.M - 1);
#pragma unroll
            for (int i = 0; i < TI; ++i) rs[i][j] = *reinterpret_cast<const float4 *>(g.residual + (long)m * g.ldr + n0 + wr * 128 + i * 16 + fq * 4);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int n = n0 + wr * 128 + i * 16 + fq * 4;
            const float4 b4 = bias_s[(n - n0) >> 2], c4 = cs_s[(n - n0) >> 2];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = ln_rstd[j] * (acc[i][j][r] - ln_mean[j] * f4(c4, r)) + f4(b4, r) + f4(rs[i][j], r);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = m0 + wc * WROWS + j * 16 + fr;
            if (m >= g.m_store) continue;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int n = n0 + wr * 128 + i * 16 + fq * 4;
                const f32x4 o = acc[i][j];
                *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
                if (g.out_bf16) {
                    uint2 pk;
                    pk.x = pack_bf16x2(o[0], o[1]);
                    pk.y = pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                }
            }
        }
        // (mean, M2) of the two 64-column slices this wave owns (sub-tiles 0-3 and 4-7): two passes over registers, the four lanes
        // that share a row (fq) combine by shuffles -- gemm_bf16_kernel's WCOLS == LN_SLOT case
        if (g.stats_out) {
            float2 *so = reinterpret_cast<float2 *>(g.stats_out);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    float sm = 0.f;
#pragma unroll
                    for (int i = 4 * sl; i < 4 * sl + 4; ++i) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                    sm += __shfl_xor(sm, 16, 64);
                    sm += __shfl_xor(sm, 32, 64);
                    const float mean = sm * (1.0f / LN_SLOT);
                    float m2 = 0.f;
#pragma unroll
                    for (int i = 4 * sl; i < 4 * sl + 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean; m2 = fmaf(d, d, m2); }
                    m2 += __shfl_xor(m2, 16, 64);
                    m2 += __shfl_xor(m2, 32, 64);
                    const int m = m0 + wc * WROWS + j * 16 + fr;
                    if (fq == 0 && m < g.m_store) so[(long)((n0 + wr * 128 + sl * 64) / LN_SLOT) * g.stats_ld + m] = make_float2(mean, m2);
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int n = n0 + wr * 128 + i * 16 + fq * 4;
            const float4 b4 = bias_s[(n - n0) >> 2], c4 = cs_s[(n - n0) >> 2];
            const bool tpart = n >= g.n_split;  // wave-uniform per sub-tile (n_split is a multiple of 16)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int m = m0 + wc * WROWS + j * 16 + fr;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ln_rstd[j] * (acc[i][j][r] - ln_mean[j] * f4(c4, r)) + f4(b4, r);
                    if (EPI == SCULPT_EPI_GELU) v = gelu_erf(v);
                    o[r] = v;
                }
                if (staged) {
                    if (tileT) {   // [tile row n][activation row m], 2 bytes each
                        // The lanes fr = 2k and 2k + 1 hold the neighbouring activation rows m, m + 1 of the same four tile rows:
                        // they swap two values each, so that the even lane writes the (m, m + 1) pairs of tile rows r = 0, 1 and
                        // the odd lane those of r = 2, 3 as whole dwords -- two ds_write_b32 per lane on 64 distinct banks instead
                        // of four 2-byte writes with two lanes per dword (the LDS conflicts of round 4's counters)
                        const bool odd = fr & 1;
                        const float s0 = odd ? o[0] : o[2], s1 = odd ? o[1] : o[3];
                        const float g0 = __shfl_xor(s0, 1, 64), g1 = __shfl_xor(s1, 1, 64);   // the partner's values of MY tile rows
                        const float a0 = odd ? g0 : o[0], b0 = odd ? o[2] : g0;   // (row m_even, row m_odd) of my first tile row
                        const float a1 = odd ? g1 : o[1], b1 = odd ? o[3] : g1;   // ... and of my second
                        const int r0 = odd ? 2 : 0;
                        unsigned char *dst = smem + (wr * 128 + i * 16 + fq * 4 + r0) * TSB + (wc * WROWS + j * 16 + (fr & ~1)) * 2;
                        *reinterpret_cast<uint32_t *>(dst) = pack_bf16x2(a0, b0);
                        *reinterpret_cast<uint32_t *>(dst + TSB) = pack_bf16x2(a1, b1);
                    } else {
                        uint2 pk;
                        pk.x = pack_bf16x2(o[0], o[1]);
                        pk.y = pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(smem + (wc * WROWS + j * 16 + fr) * RSB + (wr * 128 + i * 16 + fqs * 4) * 2) = pk;
                    }
                    continue;
                }
                if (m >= g.m_store) continue;
                if (!tpart) {
                    if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
                    if (g.out_bf16) {
                        uint2 pk;
                        pk.x = pack_bf16x2(o[0], o[1]);
                        pk.y = pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                    }
                }
                if (g.out_t && (tpart || g.n_split >= g.N)) {
                    const int nt0 = tpart ? n - g.n_split : n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g.out_t[(long)(nt0 + r) * g.ldt + m] = f32_to_bf16(o[r]);
                }
            }
        }
    }
    GEMM_STAMP(g, 3);
    if (!RES && staged) {
        // the staged tile -> HBM in whole rows: 16 bytes per lane, a row's 256 / 512 (transposed: 384 / 512) bytes contiguous
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tileT) {
            constexpr int CPR = BM / 8;   // 16-byte pieces per tile row (24 / 32); tile rows = 256 weight rows
            uint16_t *ob = g.out_t + (long)(n0 - g.n_split) * g.ldt + m0;
#pragma unroll
            for (int c0 = 0; c0 < 256 * CPR; c0 += 512) {
                const int c = c0 + tid, nl = c / CPR, mc = c - nl * CPR;
                if (c < 256 * CPR && m0 + mc * 8 < g.m_store) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(smem + nl * TSB + mc * 16);
                    if (m0 + mc * 8 + 8 <= g.m_store) {
                        *reinterpret_cast<uint4 *>(ob + (long)nl * g.ldt + mc * 8) = v;
                    } else {   // the ragged last piece of a row (M not a multiple of 8): its first M % 8 values, one by one
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 7; ++e)
                            if (m0 + mc * 8 + e < g.m_store) ob[(long)nl * g.ldt + mc * 8 + e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
                    }
                }
            }
        } else {
            constexpr int CPR = NOUT / 8;   // 16-byte pieces per tile row (16 / 32)
            uint16_t *ob = g.out_bf16 + (long)m0 * g.ldo + n0;
#pragma unroll
            for (int c0 = 0; c0 < BM * CPR; c0 += 512) {
                const int c = c0 + tid, row = c / CPR, cc = c - row * CPR;
                if (c < BM * CPR && m0 + row < g.m_store) {
                    uint4 v = *reinterpret_cast<const uint4 *>(smem + row * RSB + cc * 16);
                    if (row & 8) v = make_uint4(v.z, v.w, v.x, v.y);   // rows 8..15 of every 16 hold their piece pairs swapped (fqs above)
                    *reinterpret_cast<uint4 *>(ob + (long)row * g.ldo + cc * 8) = v;
                }
            }
        }
    }
    GEMM_STAMP(g, 4);
    GEMM_STAMP_IDS(g);
}

}  // namespace sculpt

using namespace sculpt;

extern "C" int sculpt_gemm_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                                const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                                uint16_t *out_bf16_t, int ldt, int n_split, int M, int N, int K, int epilogue,
                                sculpt_stream_t stream) {
    return sculpt_gemm_bf16_ex(A, lda, W, ldw, bias, residual, ldr, out_f32, out_bf16, ldo, out_bf16_t, ldt, n_split, 0, M, N, K,
                               epilogue, stream);
}

static constexpr long ZERO_FLOATS = 65536;  // a missing bias / colsum vector reads from here: N (GEGLU: 2N) <= 65536
static const uint16_t *zero_page() {
    // 256 KiB of zeros per device: the out-of-image taps of the implicit convolution and the stand-in for a missing per-column
    // vector (allocated once, never freed)
    static const uint16_t *pages[64] = {nullptr};
    static std::mutex mu;  // first use from several host threads at once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, ZERO_FLOATS * 4) != hipSuccess || hipMemset(p, 0, ZERO_FLOATS * 4) != hipSuccess) return nullptr;
        pages[dev] = reinterpret_cast<const uint16_t *>(p);
    }
    return pages[dev];
}

namespace sculpt {
const float *zero_floats_page(long *count) {
    if (count) *count = ZERO_FLOATS;
    return reinterpret_cast<const float *>(zero_page());
}
}  // namespace sculpt

extern "C" int sculpt_conv3x3_bf16(const uint16_t *in, int ld_in, int n_images, int H, int W, int C_pad, int dilation,
                                   const uint16_t *Wt, const float *bias, float *out_f32, uint16_t *out_bf16, int ldo,
                                   int n_store, int N, int epilogue, sculpt_stream_t stream) {
    SC_REQUIRE(in && Wt && (out_f32 || out_bf16), "conv3x3_bf16: null argument");
    SC_REQUIRE(n_images >= 1 && H >= 1 && W >= 1 && dilation >= 1, "conv3x3_bf16: bad image shape");
    SC_REQUIRE(C_pad >= 64 && C_pad % 64 == 0 && ld_in % 8 == 0, "conv3x3_bf16: C_pad=%d must be a multiple of 64 (ld %d of 8)", C_pad, ld_in);
    SC_REQUIRE(N % 128 == 0 && ldo % 4 == 0, "conv3x3_bf16: N=%d must be a multiple of 128", N);
    SC_REQUIRE(epilogue == SCULPT_EPI_NONE || epilogue == SCULPT_EPI_RELU, "conv3x3_bf16: epilogue must be NONE or RELU");
    if (n_store <= 0 || n_store > N) n_store = N;
    SC_REQUIRE(n_store % 4 == 0, "conv3x3_bf16: n_store must be a multiple of 4");
    const uint16_t *zp = zero_page();
    SC_REQUIRE(zp, "conv3x3_bf16: could not allocate the zero page");
    const long M = (long)n_images * H * W;
    const int K = 9 * C_pad;
    GemmArgs g{in, ld_in, Wt, K, bias, nullptr, 0, out_f32, out_bf16, ldo, nullptr, 0, (int)M, N, K, N, (long)N > M ? 1 : 0,
               n_store, H, W, C_pad / 64, dilation, zp, nullptr, 0, nullptr, 0.f, nullptr, 0, reinterpret_cast<const float *>(zp)};
    SC_REQUIRE((long)N <= ZERO_FLOATS, "conv3x3_bf16: N=%d too large", N);
    g.m_store = (int)M;
    const int mt = cdiv(M, BM_DEFAULT);
    hipStream_t st = as_stream(stream);
    const bool small = (long)(N / 128) * mt < (long)num_cus() * 3 / 2;
    if (epilogue == SCULPT_EPI_RELU) {
        if (small) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 64, 8, true>), dim3(N / 64, mt), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 128, 8, true>), dim3(N / 128, mt), dim3(512), 0, st, g);
    } else {
        if (small) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 8, true>), dim3(N / 64, mt), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 128, 8, true>), dim3(N / 128, mt), dim3(512), 0, st, g);
    }
    SC_LAUNCH_CHECK();
    return 0;
}

extern "C" int sculpt_gemm_bf16_ex(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                                   const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                                   uint16_t *out_bf16_t, int ldt, int n_split, int n_store, int M, int N, int K, int epilogue,
                                   sculpt_stream_t stream) {
    return sculpt_gemm_bf16_ln(A, lda, W, ldw, bias, residual, ldr, out_f32, out_bf16, ldo, out_bf16_t, ldt, n_split, n_store, M, N,
                               K, epilogue, nullptr, stream);
}

extern "C" int sculpt_gemm_bf16_ln(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                                   const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                                   uint16_t *out_bf16_t, int ldt, int n_split, int n_store, int M, int N, int K, int epilogue,
                                   const sculpt_ln_fold_t *ln, sculpt_stream_t stream) {
    SC_REQUIRE(A && W, "gemm_bf16: null operand");
    SC_REQUIRE(out_f32 || out_bf16 || out_bf16_t, "gemm_bf16: no output");
    SC_REQUIRE(M >= 1 && N >= 1 && K >= BK, "gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    SC_REQUIRE(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
    SC_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, "gemm_bf16: lda/ldw must be multiples of 8 elements (16-byte rows)");
    SC_REQUIRE(ldo % 4 == 0 && (!residual || ldr % 4 == 0), "gemm_bf16: ldo/ldr must be multiples of 4");
    if (n_split <= 0 || n_split > N) n_split = N;  // no split: every column goes to every given output
    SC_REQUIRE(n_split % 16 == 0, "gemm_bf16: n_split=%d must be a multiple of 16", n_split);
    SC_REQUIRE(n_split == N || out_bf16_t, "gemm_bf16: n_split needs the transposed output");
    const long w_rows = (epilogue == SCULPT_EPI_GEGLU) ? 2L * N : N;
    if (n_store <= 0 || n_store > N) n_store = N;
    SC_REQUIRE(n_store % 4 == 0, "gemm_bf16: n_store=%d must be a multiple of 4", n_store);
    SC_REQUIRE(n_store == N || (epilogue != SCULPT_EPI_GEGLU && !out_bf16_t), "gemm_bf16: n_store is for plain outputs only");
    // the epilogue fetches the residual of every column sub-tile before it knows which ones are stored: with a narrower
    // output slice that would read past the residual's last row
    SC_REQUIRE(!residual || n_store == N, "gemm_bf16: a residual needs n_store == N");
    GemmArgs g{A, lda, W, ldw, bias, residual, ldr, out_f32, out_bf16, ldo, out_bf16_t, ldt, M, N, K, n_split,
               w_rows > (long)M ? 1 : 0, n_store, 0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 0.f, nullptr, 0, nullptr};
    SC_REQUIRE(w_rows <= ZERO_FLOATS, "gemm_bf16: N=%d too large", N);
    g.m_store = M;
#ifdef SCULPT_EXPERIMENTS
    g.stamps = g_gemm_stamps;
#endif
    g.zeros = reinterpret_cast<const float *>(zero_page());
    SC_REQUIRE(g.zeros, "gemm_bf16: could not allocate the zero page");
    SC_REQUIRE((!bias || ((uintptr_t)bias & 15) == 0) && (!ln || !ln->colsum || ((uintptr_t)ln->colsum & 15) == 0),
               "gemm_bf16: bias / colsum must be 16-byte aligned");
    if (ln && ln->stats_in) {
        SC_REQUIRE(ln->colsum && ln->slots_in >= 1 && ln->slots_in <= 16, "gemm_bf16_ln: stats_in needs colsum and 1 <= slots_in <= 16 (K <= 1024)");
        SC_REQUIRE(ln->slots_in * LN_SLOT == K, "gemm_bf16_ln: slots_in=%d x %d columns must cover the K=%d normalised columns", ln->slots_in, LN_SLOT, K);
        g.ln_stats = ln->stats_in; g.ln_slots = ln->slots_in; g.ln_colsum = ln->colsum; g.ln_eps = ln->eps;
    }
    if (ln && (ln->stats_in || ln->stats_out)) {
        SC_REQUIRE(ln->stats_ld >= M, "gemm_bf16_ln: stats_ld=%d must be >= M=%d", ln->stats_ld, M);
        g.stats_ld = ln->stats_ld;
    }
    if (ln && ln->stats_out) {
        SC_REQUIRE(epilogue == SCULPT_EPI_NONE && out_f32 && n_store == N && n_split == N,
                   "gemm_bf16_ln: stats_out is for the plain fp32 output (no activation, no column split / n_store)");
        g.stats_out = ln->stats_out;
    }
    const int mt = cdiv(M, BM_DEFAULT);
    // The tile FORM is chosen for Mh rows: M itself, or -- a batched pass that stacks the token rows of several images and says so
    // (ln->rows_per_image) -- the rows of one image: the pass then runs the kernels of the single-image pass on taller grids, so
    // that every output AND every slice statistic is accumulated in the single-image order (the k-split pairs of the 192 x 64
    // tiles, the half-slice merges of the 64-row weight tiles): scene codes bit-identical to one image at a time.
    const int Mh = (ln && ln->rows_per_image > 0 && ln->rows_per_image < M) ? ln->rows_per_image : M;
    const int mth = cdiv(Mh, BM_DEFAULT);
    hipStream_t st = as_stream(stream);
    // Grouped tile order (GemmArgs::gm): group height ~ sqrt(R * weight rows per tile / activation rows per tile) with R the
    // workgroups an XCD keeps resident (32 CUs x 1 or 2), as a power of two; launches whose tiles are all resident at once (or
    // whose grid is shorter than two groups) keep the band order.  SCULPT_GEMM_TILE=gm=0 restores the band order everywhere (A/B),
    // gm=n forces a group height.
    constexpr const char *FORM = "SCULPT_GEMM_TILE";   // tokens: 256 / no256, 192 / no192, res / nores, bm192 / nobm192, ks0, nostage, gm=n
    auto group_rows = [&](int w_rows_tile, int a_rows_tile, int wg_per_cu, long tiles, int gy) -> int {
        const int forced = form_int(FORM, "gm", -1);
        if (forced >= 0) return forced;
        const long R = 32L * wg_per_cu;
        if (tiles <= 8 * R) return 0;
        int gmv = 1;
        while ((long)(2 * gmv) * (2 * gmv) * a_rows_tile <= R * w_rows_tile * 2) gmv *= 2;   // gm^2 ~ R * bw / bm, rounded up in log2
        return gy >= 2 * gmv ? gmv : 0;
    };
    // The 256 x 256 tile (gemm256_kernel) where it wins: launches of at least two full rounds of CUs -- StableFast-3D's 27 648-token
    // FF1 + GEGLU (622 -> 500 us), wide plain projections (8192 x 8192 x 1024: 205 -> 168 us).  Its K loop runs at 1.14 us per
    // K-tile (80 % of the matrix peak at two waves per SIMD), but a tile costs ~10-14 us besides (pipeline fill, and an epilogue
    // that nothing overlaps with one workgroup per CU: 128 accumulators per lane through LayerNorm fold + GEGLU is ~6 us), so
    // TripoSR's own launches -- FF1 is 384 such tiles = 1.5 rounds, the fused QKV 144 -- stay on the 128-row tiles
    // (tools/time_gemm256.py: 72.8 vs 60.7 us, 38.1 vs 34.9 us; outputs bit-identical).  No residual / statistics / n_store.
    {
        // no256: never, default: by the rules below, 256: whenever legal (tests, A/B)
        const int p256 = form_has(FORM, "no256") ? 0 : (form_has(FORM, "256") ? 2 : 1);
        const int nout = epilogue == SCULPT_EPI_GEGLU ? 128 : 256;
        const long tiles = (long)(N / nout) * cdiv(Mh, 256), tiles192 = (long)(N / nout) * cdiv(Mh, 192);
        const bool pays = tiles >= 2L * num_cus() && (epilogue == SCULPT_EPI_GEGLU || N >= 4096) && (Mh % 256 == 0 || Mh % 256 >= 128);
        // 192 x 256 tiles (round 4, tools/gemm_order_ab.py, interleaved in one process): wherever 3072-row multiples give them at
        // least 3/4 of the CUs a tile -- B = 1: FF1 512 tiles 58.8 vs 61.4 us on the 128-row tiles, fused Q|K|V^T 192 tiles 31.3 vs
        // 35.0; a 4-image batch: Q|K|V^T 93.8 vs 101.0, cross-attention q 33.2 vs 36.6 -- except where the 256-row tile has four
        // rounds of its own (FF1 of a 4-image batch: 211.9 vs 225.7 us).  no192 / 192 force never / always.
        const int f192 = form_has(FORM, "no192") ? 0 : (form_has(FORM, "192") ? 1 : -1);
        const bool pays192 = (Mh % 192 == 0 && M % 192 == 0 && K >= 1024 && tiles192 * 4 >= 3L * num_cus() && !(pays && tiles >= 4L * num_cus())) ||
                             // ... and a wide launch whose rows leave the last 192-row tile at most 1/8 of the rows short: the cross-attention
                             // K | V^T projection of all sixteen blocks (1025 x 32768 x 768: 62 us against 72 on the 128 x 128 tiles)
                             (Mh == M && epilogue == SCULPT_EPI_NONE && N >= 8192 && K >= 512 && tiles192 >= 2L * num_cus() &&
                              (long)cdiv(M, 192) * 192 - M <= M / 8);
        // the residual form on 192 x 256 tiles (round 4): every CU gets a tile where the 128-row tiles need three -- the N = 1024
        // projections of a batched pass (to_out of both attentions, FF2): M = 12288 -> 4 x 64 = 256 tiles
        {
            const int fres = form_has(FORM, "nores") ? 0 : (form_has(FORM, "res") ? 1 : -1);   // never / whenever legal (A/B, tests)
            const bool legal = p256 && residual && out_f32 && epilogue == SCULPT_EPI_NONE && n_store == N && n_split == N && !out_bf16_t &&
                               N % 256 == 0 && K >= 2 * BK && (long)N * ldw * 2 < 0xffff0000L && (long)M * lda * 2 < 0xffff0000L;
            const long t192 = (long)(N / 256) * cdiv(Mh, 192);
            const bool pays_res = Mh % 192 == 0 && M % 192 == 0 && t192 * 4 >= 3L * num_cus() && t192 * 2 <= 5L * num_cus();
            if (legal && (fres >= 0 ? fres != 0 : pays_res)) {
                const dim3 grid(N / 256, cdiv(M, 192));
                g.gm = group_rows(256, 192, 1, (long)grid.x * grid.y, grid.y);
                hipLaunchKernelGGL((gemm256_kernel<SCULPT_EPI_NONE, 192, true>), grid, dim3(512), 0, st, g);
                SC_LAUNCH_CHECK();
                return 0;
            }
        }
        const bool fits = p256 && !residual && !g.stats_out && n_store == N && N % nout == 0 && K >= 2 * BK &&
                          (epilogue == SCULPT_EPI_GEGLU || epilogue == SCULPT_EPI_NONE || epilogue == SCULPT_EPI_GELU) &&
                          n_split % 16 == 0 && (long)w_rows * ldw * 2 < 0xffff0000L && (long)M * lda * 2 < 0xffff0000L &&
                          (p256 >= 2 || pays || (pays192 && f192 != 0));
        if (fits) {
            const bool bm192 = f192 >= 0 ? f192 != 0 : pays192;
            const dim3 grid(N / nout, bm192 ? cdiv(M, 192) : cdiv(M, 256));
            g.gm = group_rows(256, bm192 ? 192 : 256, 1, (long)grid.x * grid.y, grid.y);
            {
                // staged stores: bf16 outputs only, every tile entirely token-major or entirely transposed, 16-byte aligned rows
                const bool split = out_bf16_t && n_split < N;
                // (out_bf16 must exist: the token-major tiles of a split launch are written through it unconditionally)
                // nostage: direct stores from the accumulator layout (A/B)
                g.stage = !form_has(FORM, "nostage") && !out_f32 && out_bf16 && (!out_bf16_t || split) &&
                          (!split || (n_split % nout == 0 && ldt % 8 == 0 && ((uintptr_t)out_bf16_t & 15) == 0)) &&
                          ldo % 8 == 0 && ((uintptr_t)out_bf16 & 15) == 0;
            }
#define SCULPT_G256(E)                                                                                     \
    do {                                                                                                   \
        if (bm192) hipLaunchKernelGGL((gemm256_kernel<E, 192>), grid, dim3(512), 0, st, g);                \
        else hipLaunchKernelGGL((gemm256_kernel<E, 256>), grid, dim3(512), 0, st, g);                      \
    } while (0)
            if (epilogue == SCULPT_EPI_GEGLU) SCULPT_G256(SCULPT_EPI_GEGLU);
            else if (epilogue == SCULPT_EPI_GELU) SCULPT_G256(SCULPT_EPI_GELU);
            else SCULPT_G256(SCULPT_EPI_NONE);
#undef SCULPT_G256
            SC_LAUNCH_CHECK();
            return 0;
        }
    }
    // 8-wave workgroups (wave tile 32 x BW/2) measured 5-13 % faster than 4-wave ones (64 x BW/2) on every shape of
    // the two transformers except the deep-K 64-row-tile case (K = 4096, N = 1024: -4 %), which keeps 4 waves
    // ... unless the launch has fewer workgroups than CUs (the ViT's 1025 x 768 x 3072): then 8 waves are the only
    // latency hiding a CU gets
    const bool underfilled = (long)(N / 64) * mth < (long)num_cus();
    const bool nw8 = true, nw8s = K < 2048 || underfilled;
    if (epilogue == SCULPT_EPI_GEGLU) {
        SC_REQUIRE(N % 64 == 0, "gemm_bf16(GEGLU): N=%d must be a multiple of 64", N);
        SC_REQUIRE(!residual && !out_bf16_t, "gemm_bf16(GEGLU): residual/transposed output unsupported");
        g.gm = group_rows(128, BM_DEFAULT, 2, (long)(N / 64) * mt, mt);
        if (nw8) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GEGLU, 128, 8>), dim3(N / 64, mt), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GEGLU, 128, 4>), dim3(N / 64, mt), dim3(256), 0, st, g);
    } else {
        SC_REQUIRE(N % 128 == 0, "gemm_bf16: N=%d must be a multiple of 128", N);
        // fill the chip: with fewer than ~1.5 tiles per CU use the 64-row weight tile
        // (128 x 128 tiles for the N = 1024 launches -- 192 workgroups, a third fewer L2 -> LDS bytes -- measured the same at
        // K = 1024 and 10 % slower at K = 4096: one workgroup per CU pulls ~35 GB/s through its LDS-DMA queue, two pull ~57)
        const bool small = (long)(N / 128) * mth < (long)num_cus() * 3 / 2;
        g.gm = small ? group_rows(64, BM_DEFAULT, 2, (long)(N / 64) * mt, mt) : group_rows(128, BM_DEFAULT, 2, (long)(N / 128) * mt, mt);
        if (epilogue == SCULPT_EPI_GELU) {
            if (small && nw8s) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GELU, 64, 8>), dim3(N / 64, mt), dim3(512), 0, st, g);
            else if (small) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GELU, 64, 4>), dim3(N / 64, mt), dim3(256), 0, st, g);
            else if (nw8) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GELU, 128, 8>), dim3(N / 128, mt), dim3(512), 0, st, g);
            else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_GELU, 128, 4>), dim3(N / 128, mt), dim3(256), 0, st, g);
        } else if (epilogue == SCULPT_EPI_RELU) {
            if (small && nw8s) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 64, 8>), dim3(N / 64, mt), dim3(512), 0, st, g);
            else if (small) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 64, 4>), dim3(N / 64, mt), dim3(256), 0, st, g);
            else if (nw8) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 128, 8>), dim3(N / 128, mt), dim3(512), 0, st, g);
            else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_RELU, 128, 4>), dim3(N / 128, mt), dim3(256), 0, st, g);
        } else if (epilogue == SCULPT_EPI_NONE) {
            // Fewer tiles than CUs: every workgroup has a CU to itself and its K loop runs at that CU's L2 -> LDS fill rate
            // (~57 GB/s, whatever the ring depth or the number of barriers: a six-stage ring and two K-tiles per barrier both
            // measured +-0); 64 x 64 tiles put the same bytes through up to twice as many CUs.
            // One round of 192 x 64 tiles (round 5): M = 3072, N = 1024 is exactly 16 x 16 = 256 tiles, one per CU, where the 128 x 64
            // tiles are 384 (1.5 per CU) and move 14 % more bytes through L2 -> LDS -- these launches run at the chip's L2 -> LDS
            // rate (~17 TB/s; hipBLASLt's 128 x 96 stream-K kernel on the same shape moves 486 MB at the same rate), so the bytes
            // are the time.  nobm192 / bm192: never / whenever legal (A/B).  (A 96 x 128 tile -- the same 256 tiles with 12.5 %
            // fewer bytes again -- was built in round 5, measured bit-identical and NOT faster (FF2 43.6 against 42.6 us, to_out
            // 18.3 / 18.1): below ~540 MB per launch the bytes stop being the time.  Removed in round 6, DESIGN_HISTORY.md.)
            const int f192r = form_has(FORM, "nobm192") ? 0 : (form_has(FORM, "bm192") ? 1 : -1);
            const long t192r = (long)(N / 64) * (Mh / 192);
            const bool one_round = Mh % 192 == 0 && M % 192 == 0 && t192r <= (long)num_cus() && t192r * 4 >= 3L * num_cus();
            if (f192r >= 0 ? (f192r != 0 && M % 192 == 0) : one_round) {
                g.gm = 0;
                // k-split pairs (see the kernel): FF2 + residual 41.6 -> 38.4 us, plain K = 4096 37.5 -> 33.9, K = 1024 -0.3 us;
                // ks0: the weight-row split, bit-identical to the 128 x 64 tiles (A/B, tests)
                if (!form_has(FORM, "ks0"))
                    hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 8, false, 192, 2, true>), dim3(N / 64, M / 192), dim3(512), 0, st, g);
                else
                    hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 8, false, 192>), dim3(N / 64, M / 192), dim3(512), 0, st, g);
            } else
            if (small && underfilled && (long)(N / 64) * cdiv(Mh, 64) <= 2L * num_cus())
                hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 8, false, 64>), dim3(N / 64, cdiv(M, 64)), dim3(512), 0, st, g);
            else if (small && nw8s) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 8>), dim3(N / 64, mt), dim3(512), 0, st, g);
            else if (small) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 64, 4>), dim3(N / 64, mt), dim3(256), 0, st, g);
            else if (nw8) hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 128, 8>), dim3(N / 128, mt), dim3(512), 0, st, g);
            else hipLaunchKernelGGL((gemm_bf16_kernel<SCULPT_EPI_NONE, 128, 4>), dim3(N / 128, mt), dim3(256), 0, st, g);
        } else {
            SC_REQUIRE(false, "gemm_bf16: unknown epilogue %d", epilogue);
        }
    }
    SC_LAUNCH_CHECK();
    return 0;
}
