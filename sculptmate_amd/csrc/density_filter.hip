// Filtered dense density grid for gfx950 (MI355X): sculpt_density_grid_filtered.
//
// Replaces, like csrc/triplane.hip (reference file:line):
//   dense query over MarchingCubeHelper.grid_vertices     TripoSR/tsr/system.py:171-184
//   TriplaneNeRFRenderer.query_triplane + NeRFMLP.forward  TripoSR/tsr/models/nerf_renderer.py:41-91, network_utils.py:116-124
// for the one consumer the dense grid has, MarchingCubeHelper.forward (TripoSR/tsr/models/isosurface.py:41-54): marching cubes
// reads the MAGNITUDE of the volume only at the end points of sign-changing lattice edges (vertex interpolation) and at all
// corners of the cells whose sign pattern is ambiguous (Lewiner's face / interior tests, centre vertex), and the SIGN everywhere
// else.  The fp32-equivalent three-limb evaluation (six bf16 products per hidden layer, triplane.hip) is therefore needed at a
// few per cent of the lattice points; everywhere else a sign that is certainly right is enough.
//
//   pass A  density_coarse_kernel    every lattice point with ONE 16-bit product per hidden layer (fp16 or bf16 operands, fp32
//                                    accumulate: 8 instead of 48 MFMAs per layer and 32 points, no limb split).  Writes the coarse
//                                    value exp(d~ + bias) + out_add, and per 32 points along z one word of SIGN bits (value > 0)
//                                    and one word of MARK bits: |d~ + bias - log(level)| < margin, or not finite.
//   pass B  filter_points<0> +       the MARKED points (list of them, then the exact three-limb arithmetic of
//           density_list_l3k_kernel  density_grid_l3k_kernel -- the same device function, so the same bits: a point's value depends
//                                    on its own MFMA column only); a sign that comes out different is corrected in the sign plane.
//                                    After this pass the sign of EVERY lattice point is certain.
//   pass C  filter_cells +           bit arithmetic on the sign planes: the points whose VALUE marching cubes reads -- the end
//           filter_points<1> +       points of the lattice edges whose signs differ, and all corners of the cells whose sign
//           density_list_l3k_kernel  pattern is one of the 128 ambiguous ones (face / interior tests, centre vertex): see
//                                    filter_cells_kernel -- minus the marked points; list, exact arithmetic, scatter.
//
// If no coarse error |d~ - d| reaches the margin, a point whose coarse sign is wrong lies within the margin, hence is marked, hence
// re-evaluated in pass B: every sign is then the full evaluation's, pass C lists exactly the values marching cubes reads for those
// signs, and every one of them carries the full evaluation's bits -- the mesh is bit-identical to it.  The margin is calibrated by
// the caller (8 x the largest coarse error measured with SCULPT_FILTER_MARK_ALL on a probe lattice of the same scene code) and
// guarded at run time, by three observations of the list kernel (which knows both values at every point it re-evaluates):
//   max_err       the largest |d~ - d| over ALL re-evaluated points (passes B and C), wherever the exact value lies;
//   n_mismatch    UNMARKED points whose exact sign differs from the coarse one -- each is a proof of a coarse error >= margin, and
//                 the sign planes pass C's lists were built from were wrong there: the call is void (max_err = inf);
//   audit_err     the largest |d~ - d| over a pseudo-random AUDIT sample of the points nothing else looks at (unmarked, value not
//                 read by marching cubes: ~0.5 % of the lattice, one candidate per 16 % of the z words, appended to pass C's list;
//                 pass A keeps the candidate's d~ in the audit plane so that the comparison is exact however far from the level).
// What a wrong sign needs to escape the second observation: let W be the points whose sign is still wrong after pass B.  A lattice
// edge from p in W to a neighbour q outside W has different signs in the planes iff p and q have the SAME true sign; then both
// end points are listed by pass C, p is re-evaluated and counted in n_mismatch.  So the call passes only if every edge leaving W
// crosses the true surface, i.e. W is a union of whole 6-connected components of the true inside or outside, every point of which
// carries a coarse error >= margin and >= its own distance from the level.  The audit sample covers that case statistically.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "triplane_mlp.h"

namespace sculpt {

struct FilterHeader {        // first 64 bytes of the filter workspace; zeroed by every call
    int32_t n_refined;       // points re-evaluated exactly (passes B + C, audit sample included)
    uint32_t max_err_bits;   // bits of max |log coarse - log exact| over the refined points; inf: a sign mismatch or a NaN
    int32_t n_marked;        // points within the margin of the level in pass A, non-finite ones included
    int32_t n_nonfinite;     // non-finite coarse values (all marked, all re-evaluated)
    int32_t n_cells;         // active cells (corner signs differ) once every sign is certain
    int32_t n_points;        // nx * R * R
    int32_t n_first;         // list entries of pass B: the marked points
    int32_t n_second;        // list entries of pass C: the values marching cubes reads, marked points excluded, + the audit sample
    uint32_t audit_err_bits; // bits of max |d~ - d| over the audit sample (log units; inf: an audited sign is wrong)
    int32_t n_audit;         // audit points (unmarked, not read by marching cubes) in pass C's list
    int32_t n_mismatch;      // unmarked points whose exact sign differs from the coarse one (audit points included)
    int32_t n_sign_fixed;    // marked points whose coarse sign pass B corrected
    uint32_t audit_seed;     // selects the audit candidates (pass A writes it)
    int32_t pad[3];
};
static_assert(sizeof(FilterHeader) == 64, "header is 16 words");

struct FilterView {
    FilterHeader *hd;
    uint32_t *sign, *mark, *cell;        // [nx*R rows][nw words]
    float *audit;                        // [nx*R rows][nw words]: d~ + bias of the word's audit candidate (point audit_bit(word))
    uint32_t *list;                      // [nx*R*R] packed (ix << 20 | iy << 10 | iz): pass B's entries, then pass C's, upwards
                                         // from 0; the audit sample downwards from the end (n_points - 1 - k)
};

// Audit candidates: word i (32 points along z) of the planes offers point z = 32 w + (hash & 31), and is a candidate with
// probability AUDIT_Q / 256.  The same function in pass A (which keeps the candidate's coarse log density) and in pass C's list.
#ifndef SCULPT_AUDIT_Q
#define SCULPT_AUDIT_Q 41
#endif
constexpr uint32_t AUDIT_Q = SCULPT_AUDIT_Q;   // 41 / 256 / 32 = 0.50 % of the lattice
__host__ __device__ __forceinline__ uint32_t audit_hash(uint32_t word, uint32_t seed) {
    uint32_t h = word * 0x9E3779B1u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t filter_layout(int R, int nx, void *base, FilterView *v) {
    const size_t rows = (size_t)nx * R, nw = (R + 31) / 32, words = rows * nw;
    char *p = reinterpret_cast<char *>(base);
    size_t o = 0;
    auto take = [&](size_t bytes) { char *q = p ? p + o : nullptr; o += align256(bytes); return q; };
    char *hd = take(sizeof(FilterHeader));
    char *sign = take(words * 4), *mark = take(words * 4), *cell = take(words * 4);
    char *audit = take(words * 4);
    char *list = take(rows * (size_t)R * 4);
    if (v) {
        v->hd = reinterpret_cast<FilterHeader *>(hd);
        v->sign = reinterpret_cast<uint32_t *>(sign); v->mark = reinterpret_cast<uint32_t *>(mark);
        v->cell = reinterpret_cast<uint32_t *>(cell);
        v->audit = reinterpret_cast<float *>(audit);
        v->list = reinterpret_cast<uint32_t *>(list);
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// pass A: one 16-bit product per hidden layer
// ---------------------------------------------------------------------------------------------
struct CState {
    float x[8], t[8];
    unsigned h[4];
};

template <typename V8> __device__ __forceinline__ unsigned cvt_pk16(float a, float b);
template <> __device__ __forceinline__ unsigned cvt_pk16<tbf16x8>(float a, float b) { return cvt_pk_bf16(a, b); }
template <> __device__ __forceinline__ unsigned cvt_pk16<tf16x8>(float a, float b) {
    const tf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tf16x2));  // v_cvt_pk_f16_f32 (RNE; overflow -> inf)
}

// chunk C (0..6) of the SiLU + conversion of the 8 values in s.x (activations carried scaled by log2 e, silu_f); stage-major
// so that neighbours are independent
template <int C, typename V8>
__device__ __forceinline__ void cchunk(CState &s) {
    if constexpr (C == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.t[i] = __builtin_amdgcn_exp2f(-s.x[i]);
    } else if constexpr (C == 1) {
#pragma unroll
        for (int i = 4; i < 8; ++i) s.t[i] = __builtin_amdgcn_exp2f(-s.x[i]);
    } else if constexpr (C == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.t[i] = 1.0f + s.t[i];
    } else if constexpr (C == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.t[i] = __builtin_amdgcn_rcpf(s.t[i]);
    } else if constexpr (C == 4) {
#pragma unroll
        for (int i = 4; i < 8; ++i) s.t[i] = __builtin_amdgcn_rcpf(s.t[i]);
    } else if constexpr (C == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.x[i] = s.x[i] * s.t[i];
    } else if constexpr (C == 6) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s.h[q] = cvt_pk16<V8>(s.x[2 * q], s.x[2 * q + 1]);
    }
}

template <typename V8>
__device__ __forceinline__ V8 cstate_operand(const CState &s) {
    const u32x4 w = {s.h[0], s.h[1], s.h[2], s.h[3]};
    return __builtin_bit_cast(V8, w);
}

// LDS image of pass A: [leading parts of the hidden weights: NH * 2048 floats][bacc][wlast][blast]
__host__ __device__ __forceinline__ int coarse_lds_floats(int NH) { return NH * 2048 + (NH + 1) * 64 + 256 + 4; }

// One hidden layer of pass A for a tile of 32 points: in x0 / x1 the pre-activations (accumulator layout), out the next layer's.
// k-step software pipeline like l3_kstep: the SiLU + conversion of the 8 values of k-step g + 1 is issued behind the two MFMAs
// of k-step g (the MFMAs are 3 % of the layer: the SiLU is what it costs, tools/micro/silu_cost.hip).
template <typename V8>
__device__ __forceinline__ void coarse_layer(const LdsView &L, const V8 *Al, int l, int h, const f32x16 &x0, const f32x16 &x1,
                                             f32x16 &o0, f32x16 &o1) {
    f32x16 acc0 = lds_bias16(L.bacc, l + 1, h, 0);
    f32x16 acc1 = lds_bias16(L.bacc, l + 1, h, 1);
    V8 a0 = Al[0], a1 = Al[256];
    V8 b;
    {   // k-step 0's SiLU has no MFMA of this tile to run behind
        CState s;
#pragma unroll
        for (int i = 0; i < 8; ++i) s.x[i] = x0[i];
        cchunk<0, V8>(s); cchunk<1, V8>(s); cchunk<2, V8>(s); cchunk<3, V8>(s); cchunk<4, V8>(s); cchunk<5, V8>(s);
        cchunk<6, V8>(s);
        b = cstate_operand<V8>(s);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        CState s;
        V8 a0n = a0, a1n = a1;
        if (g < 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s.x[i] = (g + 1 < 2 ? x0 : x1)[8 * ((g + 1) & 1) + i];
        }
        acc0 = mfma16(a0, b, acc0);
        if (g < 3) {
            cchunk<0, V8>(s); cchunk<1, V8>(s); cchunk<2, V8>(s);
            a0n = Al[(g + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc1 = mfma16(a1, b, acc1);
        if (g < 3) {
            cchunk<3, V8>(s); cchunk<4, V8>(s); cchunk<5, V8>(s); cchunk<6, V8>(s);
            a1n = Al[256 + (g + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g < 3) b = cstate_operand<V8>(s);
        a0 = a0n; a1 = a1n;
    }
    o0 = acc0;
    o1 = acc1;
}

template <typename V8>
__global__ __launch_bounds__(1024) void density_coarse_kernel(
    const float *__restrict__ blob, const float *__restrict__ FA, const float *__restrict__ FB,
    const float *__restrict__ FC, int R, int nx, float density_bias, float out_add, float level_log, float margin,
    int mark_all, float *__restrict__ out, uint32_t *__restrict__ signbits, uint32_t *__restrict__ markbits,
    float *__restrict__ audit, uint32_t audit_seed, FilterHeader *__restrict__ hdr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpPackHeader hd = *reinterpret_cast<const MlpPackHeader *>(blob);
    const int NH = hd.NH;  // >= 1
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr->audit_seed = audit_seed;
    {
        // the leading 16-bit part of every hidden weight: the first half of each layer's [part][T][s4][lane][8] image
        const float *src = blob + (__is_same(V8, tf16x8) ? hd.off_x3h : hd.off_x3);
        for (int i = threadIdx.x; i < NH * 512; i += blockDim.x) {
            const int l = i >> 9, r = i & 511;
            reinterpret_cast<f32x4 *>(smem)[i] = reinterpret_cast<const f32x4 *>(src + (long)l * 4096)[r];
        }
        float *bacc = smem + NH * 2048;
        for (int i = threadIdx.x; i < (NH + 1) * 64; i += blockDim.x) bacc[i] = blob[hd.off_bacc + i];
        float *wl = bacc + (NH + 1) * 64;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) wl[i] = blob[hd.off_wlast + i];
        if (threadIdx.x < 4) wl[256 + threadIdx.x] = blob[hd.off_blast + threadIdx.x];
        __syncthreads();
    }
    LdsView L;
    L.hid = smem;
    L.bacc = smem + NH * 2048;
    L.wlast = L.bacc + (NH + 1) * 64;
    L.blast = L.wlast + 256;
    const int lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int nzb = (R + 31) / 32;
    const long ntiles = (long)nx * nzb * R;
    const long nw_total = (long)gridDim.x * nwave;
    // tile order of density_grid_l3k_kernel: a wave walks iy at fixed (ix, z block); the workgroups of one XCD take one band of ix
    long wid = (long)blockIdx.x * nwave + wave;
    if (gridDim.x % 8 == 0) wid = ((long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * nwave + wave;
    const long t_begin = ntiles * wid / nw_total, t_end = ntiles * (wid + 1) / nw_total;
    int iy = (int)(t_begin % R);
    int zb = (int)((t_begin / R) % nzb), ixl = (int)((t_begin / R) / nzb);
    const V8 *A = reinterpret_cast<const V8 *>(smem) + lane;  // [l][T][s4][lane]
    int n_marked = 0, n_nonfinite = 0;
    bool fresh = true;
    f32x16 fb0, fb1;  // the FB row of this lane's point: constant while the wave walks iy

    for (long t = t_begin; t < t_end; ++t, ++iy) {
        if (iy == R) {
            iy = 0;
            if (++zb == nzb) { zb = 0; ++ixl; }
            fresh = true;
        }
        const int iz = zb * 32 + p;
        const int izc = min(iz, R - 1);
        if (fresh) {
            load_row32(FB + ((long)ixl * R + izc) * 64 + h * 32, fb0, fb1);
            fresh = false;
        }
        f32x16 x0, x1, y0, y1;
        load_row32(FA + ((long)ixl * R + iy) * 64 + h * 32, x0, x1);
        load_row32(FC + ((long)iy * R + izc) * 64 + h * 32, y0, y1);
        x0 += fb0; x1 += fb1;
        x0 += y0; x1 += y1;
        // two layers per trip: x -> y -> x, so that the accumulators of one layer ARE the inputs of the next without the 32
        // register copies a loop-carried `x = acc` costs (16 v_mov_b64 of ~1050 issue cycles per layer)
        {
            int l = 0;
            for (; l + 1 < NH; l += 2) {
                f32x16 y0, y1;
                coarse_layer<V8>(L, A + (long)l * 512, l, h, x0, x1, y0, y1);
                coarse_layer<V8>(L, A + (long)(l + 1) * 512, l + 1, h, y0, y1, x0, x1);
            }
            if (l < NH) {
                f32x16 y0, y1;
                coarse_layer<V8>(L, A + (long)l * 512, l, h, x0, x1, y0, y1);
                x0 = y0;
                x1 = y1;
            }
        }
        x0 = silu16_scalar(x0);
        x1 = silu16_scalar(x1);
        const float d = last_dot(L, 0, h, x0, x1) + density_bias;
        const float v = d - level_log;                       // log-domain distance from the level
        const bool live = iz < R;
        const bool finite = fabsf(v) < INFINITY;             // false for NaN and +-inf
        const bool marked = live && (mark_all || !(fabsf(v) >= margin));  // a NaN is marked
        const float c = exp_f(d) + out_add;
        const uint32_t sb = (uint32_t)__ballot(live && c > 0.0f), mb = (uint32_t)__ballot(marked);
        n_marked += __popc(mb);
        n_nonfinite += __popc((uint32_t)__ballot(live && !finite));
        const long row = (long)ixl * R + iy;
        const long word = row * nzb + zb;
        if (lane == 0) {
            signbits[word] = sb;
            markbits[word] = mb;
        }
        if (lane == (int)(audit_hash((uint32_t)word, audit_seed) & 31u)) audit[word] = d;   // the word's audit candidate
        if (h == 0 && live) out[row * R + iz] = c;
    }
    if (lane == 0) {
        if (n_marked) atomicAdd(&hdr->n_marked, n_marked);
        if (n_nonfinite) atomicAdd(&hdr->n_nonfinite, n_nonfinite);
    }
}

// ---------------------------------------------------------------------------------------------
// pass B: possibly active cells, refined points, packed list (bit arithmetic on the sign / mark words)
// ---------------------------------------------------------------------------------------------
// Which corner VALUES does marching cubes read?  (skimage's Lewiner implementation as restated in csrc/mc.hip::classify)
//   * the case of a cell comes from the 8 corner SIGNS alone;
//   * cases 1, 2, 5, 8, 9, 11, 14 pick their tiling from the sign pattern, and every Lewiner tiling places its vertices on the
//     sign-changing edges of the cell: the only values read are the two END POINTS of each sign-changing lattice edge;
//   * cases 3, 4, 6, 7, 10, 12, 13 (128 of the 256 sign patterns) run face / interior tests on the corner values and may add the
//     centre vertex (a weighted mean of all 8): ALL 8 corners.  Those patterns are exactly the ones with a "checkerboard" face
//     (diagonal corners equal, neighbours different) or with exactly two minority corners at opposite ends of a space diagonal --
//     checked against the case table for all 256 patterns (tests/test_oracle_mc.py).
// So, once every sign is certain (pass B), a lattice point needs the exact value iff it is an end point of a lattice edge whose
// signs differ or a corner of a cell whose sign pattern is one of the 128 ambiguous ones: 6.2 % of the lattice on the bench field,
// where all corners of all possibly active cells (the first version's rule) were 16.7 %.
//
// cell (x, y, z) has corners (x..x+1, y..y+1, z..z+1); one thread per word of 32 cells along z.  Runs when every sign is certain
// (after pass B).  Output: the cells ALL of whose corners are read (an ambiguous sign pattern).
__global__ __launch_bounds__(1024) void filter_cells_kernel(const uint32_t *__restrict__ sign, int R, int nx,
                                                            uint32_t *__restrict__ cell, FilterHeader *__restrict__ hdr) {
    __shared__ int wsum[16];
    const int nw = (R + 31) / 32;
    const long words = (long)nx * R * nw;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0, active = 0;
    if (i < words) {
        const int w = (int)(i % nw);
        const long row = i / nw;
        const int y = (int)(row % R), x = (int)(row / R);
        if (x + 1 < nx && y + 1 < R) {
            const long r00 = row * nw, r01 = (row + 1) * nw, r10 = (row + R) * nw, r11 = (row + R + 1) * nw;
            const bool hi = w + 1 < nw;
            auto w64 = [&](const uint32_t *a, long r) { return (uint64_t)a[r + w] | (hi ? (uint64_t)a[r + w + 1] << 32 : 0); };
            // corner planes c[dx][dy][dz]: bit z = sign of lattice point (x + dx, y + dy, z + dz)
            const uint64_t c000 = w64(sign, r00), c010 = w64(sign, r01), c100 = w64(sign, r10), c110 = w64(sign, r11);
            const uint64_t c001 = c000 >> 1, c011 = c010 >> 1, c101 = c100 >> 1, c111 = c110 >> 1;
            const uint64_t A = c000 & c010 & c100 & c110, O = c000 | c010 | c100 | c110;
            const uint64_t same = (A & (A >> 1)) | (~O & ~(O >> 1));   // bit z: the 8 corners (z, z + 1) agree
            // a face (a, b, d, e in cyclic order) is a checkerboard: a == d, b == e, a != b
            auto chk = [](uint64_t a, uint64_t b, uint64_t d, uint64_t e) { return ~(a ^ d) & ~(b ^ e) & (a ^ b); };
            uint64_t amb = chk(c000, c010, c011, c001) | chk(c100, c110, c111, c101) | chk(c000, c100, c101, c001) |
                           chk(c010, c110, c111, c011) | chk(c000, c100, c110, c010) | chk(c001, c101, c111, c011);
            // exactly two minority corners at the ends of a space diagonal (case 4): p, q of one sign, the other six of the other
            auto diag = [](uint64_t p, uint64_t q, uint64_t o1, uint64_t o2, uint64_t o3, uint64_t o4, uint64_t o5, uint64_t o6) {
                return (p & q & ~(o1 | o2 | o3 | o4 | o5 | o6)) | (~p & ~q & (o1 & o2 & o3 & o4 & o5 & o6));
            };
            amb |= diag(c000, c111, c001, c010, c011, c100, c101, c110) | diag(c001, c110, c000, c010, c011, c100, c101, c111) |
                   diag(c010, c101, c000, c001, c011, c100, c110, c111) | diag(c011, c100, c000, c001, c010, c101, c110, c111);
            // cells exist for z <= R - 2
            const int zmax = R - 2 - 32 * w;                            // last valid bit of this word
            const uint32_t valid = zmax >= 31 ? 0xffffffffu : (zmax < 0 ? 0u : ((2u << zmax) - 1u));
            c = (uint32_t)amb & valid;
            active = (uint32_t)~same & valid;
        }
        cell[i] = c;
    }
    // active cells, for the statistics: one atomic per workgroup
    int tot = __popc(active);
#pragma unroll
    for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += wsum[k];
        if (t) atomicAdd(&hdr->n_cells, t);
    }
}

// One thread per word of 32 points along z; a workgroup reserves one contiguous range of the list for its points (one atomic):
// the list is ordered inside a workgroup's 1024 words and unordered between workgroups -- its order is irrelevant to the values.
//   STAGE 0 (pass B): the MARKED points -> list[0 .. n_first).
//   STAGE 1 (pass C): the points whose value marching cubes reads -- point (x, y, z) is a corner of the cells (x-1..x, y-1..y,
//           z-1..z): all corners of an ambiguous cell; and an end point of the six lattice edges to its neighbours: the edges
//           whose signs differ -- minus the marked ones (exact already) -> list[n_first .. n_first + n_second); and the word's
//           audit candidate when it is none of those -> list[n_points - 1 - k], k < n_audit (a segment of its own: its points
//           are scattered over the lattice, and tiles that mix them with the surface's points lose the locality of those).
template <int STAGE>
__global__ __launch_bounds__(1024) void filter_points_kernel(const uint32_t *__restrict__ cell, const uint32_t *__restrict__ sign,
                                                             const uint32_t *__restrict__ mark, int R, int nx,
                                                             uint32_t *__restrict__ list, FilterHeader *__restrict__ hdr) {
    __shared__ int wsum[16], asum[16];
    __shared__ int base_s, abase_s;
    const int nw = (R + 31) / 32;
    const long words = (long)nx * R * nw;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t r = 0, entry = 0, aud = 0;
    if (i < words) {
        const int w = (int)(i % nw);
        const long row = i / nw;
        const int y = (int)(row % R), x = (int)(row / R);
        const int zrem = R - 32 * w;                                  // points exist for z < R
        const uint32_t pvalid = zrem < 32 ? (1u << zrem) - 1u : 0xffffffffu;
        if (STAGE == 0) {
            r = mark[i] & pvalid;
        } else {
            // all corners of the ambiguous cells
            uint32_t c = 0, prev = 0;
#pragma unroll
            for (int dx = -1; dx <= 0; ++dx)
#pragma unroll
                for (int dy = -1; dy <= 0; ++dy)
                    if (x + dx >= 0 && y + dy >= 0) {
                        const long q = (row + (long)dx * R + dy) * nw + w;
                        c |= cell[q];
                        if (w) prev |= cell[q - 1];
                    }
            r = c | (c << 1) | (prev >> 31);
            // end points of the lattice edges whose signs differ
            const uint32_t S = sign[i];
            const uint32_t next0 = (w + 1 < nw) ? (sign[i + 1] & 1u) : 0u;
            uint32_t dz = S ^ ((S >> 1) | (next0 << 31));             // bit z: the edge (z, z + 1) changes sign ...
            const int ez = zrem - 1;                                  // ... where z + 1 < R: ez valid bits
            dz &= ez >= 32 ? 0xffffffffu : (ez <= 0 ? 0u : ((1u << ez) - 1u));
            uint32_t e = dz | (dz << 1);
            if (w) e |= ((sign[i - 1] >> 31) ^ S) & 1u;               // the edge (32 w - 1, 32 w) of the previous word
            if (y + 1 < R) e |= S ^ sign[i + nw];
            if (y > 0) e |= S ^ sign[i - nw];
            if (x + 1 < nx) e |= S ^ sign[i + (long)R * nw];
            if (x > 0) e |= S ^ sign[i - (long)R * nw];
            r = (r | e) & ~mark[i] & pvalid;
            const uint32_t hsh = audit_hash((uint32_t)i, hdr->audit_seed);
            if (((hsh >> 8) & 255u) < AUDIT_Q) aud = (1u << (hsh & 31u)) & ~r & ~mark[i] & pvalid;
        }
        entry = ((uint32_t)x << 20) | ((uint32_t)y << 10) | (uint32_t)(32 * w);
    }
    const int n = __popc(r);
    int incl = n, aincl = aud ? 1 : 0;  // inclusive scans inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
        if (STAGE == 1) {
            const int a = __shfl_up(aincl, o, 64);
            if (lane >= o) aincl += a;
        }
    }
    if (lane == 63) { wsum[wv] = incl; asum[wv] = aincl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0, ta = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
            const int v = wsum[k]; wsum[k] = t; t += v;
            const int a = asum[k]; asum[k] = ta; ta += a;
        }
        // (stage 1 starts behind stage 0's entries: n_first is final, pass B's list kernel has run)
        base_s = (t ? atomicAdd(STAGE == 0 ? &hdr->n_first : &hdr->n_second, t) : 0) + (STAGE == 0 ? 0 : hdr->n_first);
        if (STAGE == 1) abase_s = ta ? atomicAdd(&hdr->n_audit, ta) : 0;
        if (STAGE == 0 && blockIdx.x == 0) hdr->n_points = nx * R * R;
    }
    __syncthreads();
    int o = base_s + wsum[wv] + incl - n;
    while (r) {
        const int z = __ffs(r) - 1;
        r &= r - 1;
        list[o++] = entry + (uint32_t)z;
    }
    if (STAGE == 1 && aud)   // a point is in at most one segment, so the two ends never meet
        list[(long)nx * R * R - 1 - (abase_s + asum[wv] + aincl - 1)] = entry + (uint32_t)(__ffs(aud) - 1);
}

// ---------------------------------------------------------------------------------------------
// pass C: the exact three-limb arithmetic at the listed points (tiles of 32 list entries)
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void density_list_l3k_kernel(
    const float *__restrict__ blob, const float *__restrict__ FA, const float *__restrict__ FB,
    const float *__restrict__ FC, int R, float density_bias, float out_add,
    const uint32_t *__restrict__ list_all, FilterHeader *__restrict__ hdr, float *__restrict__ out, int stage,
    uint32_t *__restrict__ signbits, const float *__restrict__ audit, long n_points) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [W1 | W2: NH*4096][bacc][wlast][blast]
    const MlpPackHeader hd = *reinterpret_cast<const MlpPackHeader *>(blob);
    const int NH = hd.NH;
    // stage 0 (pass B): the marked points, whose coarse sign may be wrong -- a sign that comes out different is corrected in the
    // sign plane (rare: one atomic per such point), so that every sign is certain when pass C's bit kernels read the planes;
    // stage 1 (pass C): the second part of the list.
    // stage 0 / 1: list[0 .. n_first) / list[n_first .. n_first + n_second); stage 1 then takes the audit segment (read
    // backwards from the end of the list) as a second round of the same loop
    const int n_main = stage ? hdr->n_second : hdr->n_first;
    const int n_aud = stage ? hdr->n_audit : 0;
    if (stage && blockIdx.x == 0 && threadIdx.x == 0) hdr->n_refined = hdr->n_first + hdr->n_second + hdr->n_audit;
    if (n_main + n_aud <= 0) return;
    l3_load_lds(smem, blob, hd);
    const LdsView L = lds_view(smem, NH);
    const int lane = threadIdx.x & 63, nwave = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 31, h = lane >> 5;
    const long nw_total = (long)gridDim.x * nwave;
    const tbf16x8 *A = reinterpret_cast<const tbf16x8 *>(smem) + lane;
    const tbf16x8 *A3 = reinterpret_cast<const tbf16x8 *>(blob + hd.off_w3) + lane;
    float worst = 0.f, worst_audit = 0.f;
    const int main_base = stage ? hdr->n_first : 0;
    // Tile order.  The list follows the lattice (x slowest) inside a filter_points workgroup, and every re-evaluated point gathers
    // its own FC row (iy, iz) -- 256 B per point, 714 MB per launch on the bench field if none is reused.  Neighbouring x planes cut
    // the surface at nearly the same (iy, iz), so: the workgroups of one XCD (blockIdx % 8, one private L2) take one contiguous
    // eighth of a segment, and inside it the tiles are dealt ROUND-ROBIN to the XCD's waves -- at any time the XCD works on ~512
    // consecutive tiles (one or two x planes) whose FC rows fit its L2 and are still there when the next plane comes by.  (A
    // contiguous range per wave put 32 different planes into an XCD's L2 at once: FETCH_SIZE 373 MB raw per launch.)
    // seg 0: this stage's entries; seg 1 (stage 1 only): the audit sample, dealt the same way.
    for (int seg = 0; seg < (n_aud > 0 ? 2 : 1); ++seg) {
    const bool is_audit = seg;
    const int n = seg ? n_aud : n_main;
    if (n <= 0) continue;
    const long ntiles = ((long)n + 31) / 32;
    long t_begin, t_end, t_step;
    if (gridDim.x % 8 == 0) {
        const long nxw = (long)(gridDim.x >> 3) * nwave, xcd = blockIdx.x & 7;
        const long c_begin = ntiles * xcd / 8, c_end = ntiles * (xcd + 1) / 8;
        t_begin = c_begin + (long)(blockIdx.x >> 3) * nwave + wave;
        t_end = c_end;
        t_step = nxw;
    } else {
        t_begin = (long)blockIdx.x * nwave + wave;
        t_end = ntiles;
        t_step = nw_total;
    }
    for (long t = t_begin; t < t_end; t += t_step) {
        const long j = t * 32 + p;
        const bool valid = j < n;
        const long jj = valid ? j : n - 1;
        const uint32_t e = seg ? list_all[n_points - 1 - jj] : list_all[main_base + jj];
        const int ixl = (int)(e >> 20), iy = (int)((e >> 10) & 1023u), iz = (int)(e & 1023u);
        f32x16 x0, x1;
        l3_table_sum(FA, FB, FC, R, ixl, iy, iz, h, x0, x1);
        l3k_hidden(L, NH, A, A3, h, x0, x1);
        const float d = last_dot(L, 0, h, x0, x1);
        if (h == 0 && valid) {
            const int nzb = (R + 31) / 32;
            const long row = (long)ixl * R + iy, idx = row * R + iz, word = row * nzb + (iz >> 5);
            const float dl = d + density_bias;
            const float coarse = out[idx], exact = exp_f(dl) + out_add;
            // The guard.  The coarse value was exp(d~ + bias) + out_add: log(coarse - out_add) gives d~ + bias back to an ulp of
            // the level over the density, i.e. exactly enough unless the coarse density is a thousand times below the level
            // (no error is recorded there: only the sign is compared); an audit point has its d~ + bias in the audit plane.
            float err = 0.f;
            if (is_audit) err = fabsf(audit[word] - dl);
            else if (coarse - out_add > -1e-3f * out_add) err = fabsf(__logf(coarse - out_add) - dl);
            // a non-finite coarse value (err not finite) was marked and is replaced here: n_nonfinite counts those
            if (err < INFINITY) { if (is_audit) worst_audit = fmaxf(worst_audit, err); else worst = fmaxf(worst, err); }
            out[idx] = exact;
            // a NaN among the exact values fails the guard: the caller redoes the grid in full and marching cubes on the whole
            // volume reports it (the classification from the sign planes looks at values near the surface only)
            if (exact != exact) worst = INFINITY;
            if ((coarse > 0.0f) != (exact > 0.0f)) {
                atomicXor(&signbits[word], 1u << (iz & 31));
                // stage 0: a marked point, the correction pass B is there for.  stage 1: an unmarked point -- the coarse error
                // reached the margin, and pass C's lists were built from a wrong sign: the call is void.  (Rare: one atomic each.)
                if (stage) { if (is_audit) worst_audit = INFINITY; else worst = INFINITY; }
                atomicAdd(stage ? &hdr->n_mismatch : &hdr->n_sign_fixed, 1);
            }
        }
    }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        worst = fmaxf(worst, __shfl_xor(worst, o, 64));
        worst_audit = fmaxf(worst_audit, __shfl_xor(worst_audit, o, 64));
    }
    if (lane == 0) {
        if (worst > 0.f) atomicMax(&hdr->max_err_bits, __float_as_uint(worst));
        if (worst_audit > 0.f) atomicMax(&hdr->audit_err_bits, __float_as_uint(worst_audit));
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

size_t sculpt_density_filter_workspace_bytes(int R, int nx) {
    if (R < 2 || nx < 1) return 0;
    return filter_layout(R, nx, nullptr, nullptr);
}

int sculpt_density_grid_filtered(const void *mlp_packed, int n_hidden_64, int R, int x_begin, int x_end, float density_bias,
                                 float out_add, float margin, const void *workspace, void *filter_workspace, float *out,
                                 unsigned flags, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    SC_REQUIRE(mlp_packed && workspace && filter_workspace && out, "density_grid_filtered: null argument");
    SC_REQUIRE(R >= 2 && R <= 1024 && x_begin >= 0 && x_end <= R && x_begin < x_end,
               "density_grid_filtered: bad range [%d,%d) of %d (R <= 1024)", x_begin, x_end, R);
    SC_REQUIRE(n_hidden_64 >= 1, "density_grid_filtered: needs at least one 64x64 hidden layer (got %d)", n_hidden_64);
    SC_REQUIRE(flags & SCULPT_DENSITY_BF16L3, "density_grid_filtered: the exact pass is the three-limb kernel (pass SCULPT_DENSITY_BF16L3)");
    const unsigned pass_bits = SCULPT_FILTER_PASS_A | SCULPT_FILTER_PASS_B | SCULPT_FILTER_PASS_C;
    SC_REQUIRE((flags & ~(SCULPT_DENSITY_BF16L3 | SCULPT_FILTER_COARSE_FP16 | SCULPT_FILTER_MARK_ALL | pass_bits)) == 0,
               "density_grid_filtered: unknown flags %u", flags);
    const unsigned passes = (flags & pass_bits) ? (flags & pass_bits) : pass_bits;
    const bool mark_all = flags & SCULPT_FILTER_MARK_ALL;
    SC_REQUIRE(mark_all || (out_add < 0.f && margin > 0.f && margin < INFINITY),
               "density_grid_filtered: needs a positive level (out_add = -level < 0, got %g) and a finite margin > 0 (got %g)",
               (double)out_add, (double)margin);
    const size_t lds_c = (size_t)coarse_lds_floats(n_hidden_64) * sizeof(float);
    const size_t lds_x = (size_t)lds_floats_for(n_hidden_64) * sizeof(float);
    SC_REQUIRE(lds_x <= 160 * 1024, "density_grid_filtered: %d hidden layers do not fit LDS", n_hidden_64);
    const int nx = x_end - x_begin;
    const float *FA = reinterpret_cast<const float *>(workspace);
    const float *FB = FA + (size_t)nx * R * 64;
    const float *FC = FB + (size_t)nx * R * 64;
    FilterView v;
    filter_layout(R, nx, filter_workspace, &v);
    const float level_log = out_add < 0.f ? logf(-out_add) : 0.f;
    const float *blob = reinterpret_cast<const float *>(mlp_packed);
    const long ntiles = (long)nx * ((R + 31) / 32) * R;
    if (passes & SCULPT_FILTER_PASS_A) {
        SC_HIP(hipMemsetAsync(v.hd, 0, sizeof(FilterHeader), st));
        auto kern = (flags & SCULPT_FILTER_COARSE_FP16) ? density_coarse_kernel<tf16x8> : density_coarse_kernel<tbf16x8>;
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        int threads = 1024;
        size_t lds_launch = lds_c;
#ifdef SCULPT_EXPERIMENTS
        // tools/try_coresident.py (builds with SCULPT_EXTRA_HIPCC_FLAGS=-DSCULPT_EXPERIMENTS only): SCULPT_COARSE_WG="threads,lds_kb"
        // -- a narrower workgroup with its LDS padded so that only one fits a CU
        if (const char *e = getenv("SCULPT_COARSE_WG")) {
            int t = 0, kb = 0;
            if (sscanf(e, "%d,%d", &t, &kb) == 2 && (t == 256 || t == 512 || t == 1024)) {
                threads = t;
                lds_launch = std::max(lds_c, (size_t)kb * 1024);
                SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
            }
        }
#endif
        const int grid = (int)std::min<long>((ntiles + 15) / 16, num_cus());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_launch, st, blob, FA, FB, FC, R, nx, density_bias, out_add, level_log,
                           margin, mark_all ? 1 : 0, out, v.sign, v.mark, v.audit,
                           audit_hash((uint32_t)R * 2654435761u + (uint32_t)x_begin, (uint32_t)nx), v.hd);
        SC_LAUNCH_CHECK();
    }
    auto list_pass = [&](int stage) -> int {
        auto kern = density_list_l3k_kernel<1024>;
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x));
        const int grid = (int)std::min<long>((ntiles + 15) / 16, num_cus());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds_x, st, blob, FA, FB, FC, R, density_bias, out_add,
                           v.list, v.hd, out, stage, v.sign, v.audit, (long)nx * R * R);
        SC_LAUNCH_CHECK();
        return 0;
    };
    const long rows = (long)nx * R, words = rows * ((R + 31) / 32);
    if (passes & SCULPT_FILTER_PASS_B) {   // the marked points: exact values, and with them a certain sign everywhere
        hipLaunchKernelGGL(filter_points_kernel<0>, dim3(cdiv(words, 1024)), dim3(1024), 0, st, v.cell, v.sign, v.mark, R, nx, v.list, v.hd);
        SC_LAUNCH_CHECK();
        if (int rc = list_pass(0)) return rc;
    }
    if (passes & SCULPT_FILTER_PASS_C) {   // the values marching cubes reads
        hipLaunchKernelGGL(filter_cells_kernel, dim3(cdiv(words, 1024)), dim3(1024), 0, st, v.sign, R, nx, v.cell, v.hd);
        hipLaunchKernelGGL(filter_points_kernel<1>, dim3(cdiv(words, 1024)), dim3(1024), 0, st, v.cell, v.sign, v.mark, R, nx, v.list, v.hd);
        SC_LAUNCH_CHECK();
        if (int rc = list_pass(1)) return rc;
    }
    return 0;
}

size_t sculpt_density_filter_sign_offset(int R, int nx) {
    if (R < 2 || nx < 1) return 0;
    FilterView v;
    char base[1];
    filter_layout(R, nx, base, &v);
    return (size_t)(reinterpret_cast<char *>(v.sign) - base);
}

int sculpt_density_filter_stats(const void *filter_workspace, int32_t *stats12, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    SC_REQUIRE(filter_workspace && stats12, "density_filter_stats: null argument");
    SC_HIP(hipMemcpyAsync(stats12, filter_workspace, SCULPT_FILTER_STATS_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
