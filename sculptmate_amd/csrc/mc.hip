// Marching cubes (Lewiner) on gfx950, output identical -- including vertex and face ORDER -- to
// skimage.measure.marching_cubes(vol, level) with default arguments, which is what the reference
// calls on the CPU at TripoSR/tsr/models/isosurface.py:46-48 (MarchingCubeHelper.forward :41-54).
//
// The sequential algorithm visits cells with the slowest array axis outermost and creates a
// vertex the first time a cell touches a lattice edge.  The parallel restatement:
//   * cell order            = C-order linear index of the cell grid (n0-1, n1-1, n2-1)
//   * owner of an edge      = the first of the (up to 4) cells sharing it in that order; every
//                             Lewiner tiling uses exactly the sign-changing edges of its cell
//                             (checked over all tables in tests/test_mc_tables.py), so ownership is
//                             purely geometric
//   * vertex id             = exclusive scan over cells of "#vertices owned", plus the rank of the
//                             edge among the owner's owned vertices in first-appearance order of the
//                             owner's triangle list (centre vertex 12 is always owned)
//   * face offset           = exclusive scan over cells of "#triangles"
// Passes (each one thread per cell, 256 cells per workgroup, wave-level prefix scans):
//   count : classify, per-block (ntri, nown) sums, data min/max
//   scan  : exclusive scan of the block sums (single workgroup), totals
//   verts : re-classify, block scan, owners write vertex positions + lattice-edge -> id map
//   faces : re-classify, block scan, every active cell writes its triangles through the map
// HBM traffic: 3 reads of the volume (4 B/voxel each) + 12 B/vertex + 12|24 B/face + sparse map.
#include <float.h>
#include <math.h>
#include <string.h>

#include "common.h"
#include "mc_luts.h"

// The CPU implementation rounds every double operation separately (x86-64, no FMA contraction);
// keep the device arithmetic identical.
#pragma clang fp contract(off)

namespace sculpt {

// A product that must be rounded on its own (the CPU code has no FMA): the empty asm makes the value
// opaque to the optimiser, so no later add can be fused with it.  (HIP's __dmul_rn/__dadd_rn are
// plain operators parsed under the default contract=fast and DO get fused after inlining.)
__device__ __forceinline__ double rounded(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

static constexpr double MC_EPS = 2.220446049250313e-16;  // skimage: np.spacing(1.0)
static constexpr int MC_BLOCK = 256;

// device copies of the tables
__constant__ signed char d_tiling_flat[MC_TILING_FLAT_SIZE];
__constant__ unsigned short d_tiling_base[MC_NUM_TILINGS];
__constant__ unsigned char d_tiling_rowlen[MC_NUM_TILINGS];
__constant__ unsigned char d_tiling_inner[MC_NUM_TILINGS];
__constant__ signed char d_cases[512];
__constant__ signed char d_cases_classic[256 * 16];
__constant__ signed char d_test3[24], d_test4[8], d_test6[48 * 3], d_test7[16 * 5], d_test10[6 * 3],
    d_test12[24 * 4], d_test13[2 * 7], d_subconfig13[64];

static bool g_tables_uploaded[64] = {false};

static int upload_tables() {
    int dev = 0;
    SC_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && g_tables_uploaded[dev]) return 0;
#define UP(sym, src) SC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sym), src, sizeof(src)))
    UP(d_tiling_flat, mc_tiling_flat);
    UP(d_tiling_base, mc_tiling_base);
    UP(d_tiling_rowlen, mc_tiling_rowlen);
    UP(d_tiling_inner, mc_tiling_inner);
    UP(d_cases, mc_cases);
    UP(d_cases_classic, mc_cases_classic);
    UP(d_test3, mc_test3);
    UP(d_test4, mc_test4);
    UP(d_test6, mc_test6);
    UP(d_test7, mc_test7);
    UP(d_test10, mc_test10);
    UP(d_test12, mc_test12);
    UP(d_test13, mc_test13);
    UP(d_subconfig13, mc_subconfig13);
#undef UP
    if (dev >= 0 && dev < 64) g_tables_uploaded[dev] = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// classification (per cell); v[8] = corner values minus level (double), Lewiner corner order
// ---------------------------------------------------------------------------------------------
__device__ bool test_face(const double *v, int face) {
    const int af = face < 0 ? -face : face;
    double A, B, C, D;
    switch (af) {
        case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
        case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
        case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
        case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
        case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
        default: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
    }
    // no contraction: A*C and B*D are rounded separately in the CPU implementation
    const double ac = rounded(A * C), bd = rounded(B * D);
    const double acbd = ac - bd;
    if (acbd > -MC_EPS && acbd < MC_EPS) return face >= 0;
    return rounded(rounded((double)face * A) * acbd) >= 0;
}

__device__ bool test_internal(const double *v, int mc_case, int config, int subconfig, int s) {
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    int test = 0;
#define MUL(a, b) rounded((a) * (b))
#define ADD(a, b) ((a) + (b))
#define SUB(a, b) ((a) - (b))
    if (mc_case == 4 || mc_case == 10) {
        const double a = SUB(MUL(SUB(v[4], v[0]), SUB(v[6], v[2])), MUL(SUB(v[7], v[3]), SUB(v[5], v[1])));
        const double b = SUB(SUB(ADD(MUL(v[2], SUB(v[4], v[0])), MUL(v[0], SUB(v[6], v[2]))),
                                 MUL(v[1], SUB(v[7], v[3]))),
                             MUL(v[3], SUB(v[5], v[1])));
        t = -b / ADD(MUL(2.0, a), MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = ADD(v[0], MUL(SUB(v[4], v[0]), t));
        Bt = ADD(v[3], MUL(SUB(v[7], v[3]), t));
        Ct = ADD(v[2], MUL(SUB(v[6], v[2]), t));
        Dt = ADD(v[1], MUL(SUB(v[5], v[1]), t));
    } else {
        int edge;
        if (mc_case == 6) edge = d_test6[config * 3 + 2];
        else if (mc_case == 7) edge = d_test7[config * 5 + 4];
        else if (mc_case == 12) edge = d_test12[config * 4 + 3];
        else edge = d_tiling_flat[d_tiling_base[MC_T_13_5_1] + (config * 4 + subconfig) * 18];
        // reference edge e from corner a to corner b; the three "parallel" edges (p0,p1),(q0,q1),(r0,r1)
        int ea, eb, p0, p1, q0, q1, r0, r1;
        switch (edge) {
            case 0: ea = 0; eb = 1; p0 = 3; p1 = 2; q0 = 7; q1 = 6; r0 = 4; r1 = 5; break;
            case 1: ea = 1; eb = 2; p0 = 0; p1 = 3; q0 = 4; q1 = 7; r0 = 5; r1 = 6; break;
            case 2: ea = 2; eb = 3; p0 = 1; p1 = 0; q0 = 5; q1 = 4; r0 = 6; r1 = 7; break;
            case 3: ea = 3; eb = 0; p0 = 2; p1 = 1; q0 = 6; q1 = 5; r0 = 7; r1 = 4; break;
            case 4: ea = 4; eb = 5; p0 = 7; p1 = 6; q0 = 3; q1 = 2; r0 = 0; r1 = 1; break;
            case 5: ea = 5; eb = 6; p0 = 4; p1 = 7; q0 = 0; q1 = 3; r0 = 1; r1 = 2; break;
            case 6: ea = 6; eb = 7; p0 = 5; p1 = 4; q0 = 1; q1 = 0; r0 = 2; r1 = 3; break;
            case 7: ea = 7; eb = 4; p0 = 6; p1 = 5; q0 = 2; q1 = 1; r0 = 3; r1 = 0; break;
            case 8: ea = 0; eb = 4; p0 = 3; p1 = 7; q0 = 2; q1 = 6; r0 = 1; r1 = 5; break;
            case 9: ea = 1; eb = 5; p0 = 0; p1 = 4; q0 = 3; q1 = 7; r0 = 2; r1 = 6; break;
            case 10: ea = 2; eb = 6; p0 = 1; p1 = 5; q0 = 0; q1 = 4; r0 = 3; r1 = 7; break;
            default: ea = 3; eb = 7; p0 = 2; p1 = 6; q0 = 1; q1 = 5; r0 = 0; r1 = 4; break;  // 11
        }
        t = v[ea] / ADD(SUB(v[ea], v[eb]), MC_EPS);
        At = 0;
        Bt = ADD(v[p0], MUL(SUB(v[p1], v[p0]), t));
        Ct = ADD(v[q0], MUL(SUB(v[q1], v[q0]), t));
        Dt = ADD(v[r0], MUL(SUB(v[r1], v[r0]), t));
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    const double det = SUB(MUL(At, Ct), MUL(Bt, Dt));
#undef MUL
#undef ADD
#undef SUB
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        // scikit-image's Cython port falls off the end (returns 0) when the inner test fails
        case 5: return (det < MC_EPS) ? (s > 0) : false;
        case 10: return (det >= MC_EPS) ? (s > 0) : false;
        default: return s < 0;
    }
}

// returns tiling offset into d_tiling_flat and number of index entries (3*ntri); 0 if inactive
struct Tiling {
    int ofs;
    int len;
};

__device__ Tiling classify(const double *v, bool classic) {
    int index = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) index |= (v[k] > 0.0) ? (1 << k) : 0;
    Tiling out;
    out.ofs = 0;
    out.len = 0;
    if (index == 0 || index == 255) return out;
    if (classic) {
        // classic rows live in their own table; flag with negative offset
        int n = 0;
        while (n < 16 && d_cases_classic[16 * index + n] != -1) ++n;
        out.ofs = -(16 * index) - 1;
        out.len = n;
        return out;
    }
    const int c = d_cases[2 * index], cfg = d_cases[2 * index + 1];
    int table = -1, sub = 0, sc = 0;
    switch (c) {
        case 1: table = MC_T_1; break;
        case 2: table = MC_T_2; break;
        case 3: table = test_face(v, d_test3[cfg]) ? MC_T_3_2 : MC_T_3_1; break;
        case 4: table = test_internal(v, c, cfg, 0, d_test4[cfg]) ? MC_T_4_1 : MC_T_4_2; break;
        case 5: table = MC_T_5; break;
        case 6:
            if (test_face(v, d_test6[cfg * 3 + 0])) table = MC_T_6_2;
            else table = test_internal(v, c, cfg, 0, d_test6[cfg * 3 + 1]) ? MC_T_6_1_1 : MC_T_6_1_2;
            break;
        case 7:
            if (test_face(v, d_test7[cfg * 5 + 0])) sc += 1;
            if (test_face(v, d_test7[cfg * 5 + 1])) sc += 2;
            if (test_face(v, d_test7[cfg * 5 + 2])) sc += 4;
            switch (sc) {
                case 0: table = MC_T_7_1; break;
                case 1: table = MC_T_7_2; sub = 0; break;
                case 2: table = MC_T_7_2; sub = 1; break;
                case 3: table = MC_T_7_3; sub = 0; break;
                case 4: table = MC_T_7_2; sub = 2; break;
                case 5: table = MC_T_7_3; sub = 1; break;
                case 6: table = MC_T_7_3; sub = 2; break;
                default: table = test_internal(v, c, cfg, sc, d_test7[cfg * 5 + 3]) ? MC_T_7_4_2 : MC_T_7_4_1; break;
            }
            break;
        case 8: table = MC_T_8; break;
        case 9: table = MC_T_9; break;
        case 10:
            if (test_face(v, d_test10[cfg * 3 + 0])) table = test_face(v, d_test10[cfg * 3 + 1]) ? MC_T_10_1_1_ : MC_T_10_2;
            else if (test_face(v, d_test10[cfg * 3 + 1])) table = MC_T_10_2_;
            else table = test_internal(v, c, cfg, 0, d_test10[cfg * 3 + 2]) ? MC_T_10_1_1 : MC_T_10_1_2;
            break;
        case 11: table = MC_T_11; break;
        case 12:
            if (test_face(v, d_test12[cfg * 4 + 0])) table = test_face(v, d_test12[cfg * 4 + 1]) ? MC_T_12_1_1_ : MC_T_12_2;
            else if (test_face(v, d_test12[cfg * 4 + 1])) table = MC_T_12_2_;
            else table = test_internal(v, c, cfg, 0, d_test12[cfg * 4 + 2]) ? MC_T_12_1_1 : MC_T_12_1_2;
            break;
        case 13:
            for (int k = 0; k < 6; ++k)
                if (test_face(v, d_test13[cfg * 7 + k])) sc += 1 << k;
            sc = d_subconfig13[sc];
            if (sc == 0) table = MC_T_13_1;
            else if (sc <= 6) { table = MC_T_13_2; sub = sc - 1; }
            else if (sc <= 18) { table = MC_T_13_3; sub = sc - 7; }
            else if (sc <= 22) { table = MC_T_13_4; sub = sc - 19; }
            else if (sc <= 26) {
                sub = sc - 23;
                table = test_internal(v, c, cfg, sub, d_test13[cfg * 7 + 6]) ? MC_T_13_5_1 : MC_T_13_5_2;
            } else if (sc <= 38) { table = MC_T_13_3_; sub = sc - 27; }
            else if (sc <= 44) { table = MC_T_13_2_; sub = sc - 39; }
            else if (sc == 45) table = MC_T_13_1_;
            break;
        case 14: table = MC_T_14; break;
        default: break;
    }
    if (table < 0) return out;
    const int rowlen = d_tiling_rowlen[table];
    out.ofs = d_tiling_base[table] + (cfg * d_tiling_inner[table] + sub) * rowlen;
    out.len = rowlen;
    return out;
}

__device__ __forceinline__ int tiling_entry(const Tiling &t, int i) {
    return (t.ofs < 0) ? d_cases_classic[(-t.ofs - 1) + i] : d_tiling_flat[t.ofs + i];
}

// geometry of the cell grid
struct Grid {
    int n0, n1, n2;   // voxels
    int c0, c1, c2;   // cells
    long ncells;
    int halo_low;     // slab mode: lattice plane 0 belongs to the previous slab (its x/y edges are not owned here)
    int z_off;        // slab mode: global index of this slab's lattice plane 0 (added to axis-0 coordinates)
};

__device__ __forceinline__ void load_cell(const float *__restrict__ vol, const Grid &g, int z, int y, int x,
                                          double level, double *v) {
    const long sy = g.n2, sz = (long)g.n1 * g.n2;
    const float *p = vol + z * sz + y * sy + x;
    v[0] = (double)p[0] - level;
    v[1] = (double)p[1] - level;
    v[2] = (double)p[sy + 1] - level;
    v[3] = (double)p[sy] - level;
    v[4] = (double)p[sz] - level;
    v[5] = (double)p[sz + 1] - level;
    v[6] = (double)p[sz + sy + 1] - level;
    v[7] = (double)p[sz + sy] - level;
}

// Workgroup -> cells: a workgroup covers up to 256 consecutive x of ONE (z,y) row, so the cell
// coordinates need one scalar division per workgroup instead of three 64-bit divisions per thread, the
// corner loads of a wave are contiguous, and workgroup order == cell order (needed by the scans).
__device__ __forceinline__ bool cell_of_block(const Grid &g, int &x, int &y, int &z) {
    const int bpr = (g.c2 + MC_BLOCK - 1) / MC_BLOCK;          // workgroups per row
    const unsigned row = blockIdx.x / (unsigned)bpr;             // wave-uniform
    const int seg = (int)(blockIdx.x - row * (unsigned)bpr);
    x = seg * MC_BLOCK + (int)threadIdx.x;
    y = (int)(row % (unsigned)g.c1);
    z = (int)(row / (unsigned)g.c1);
    return x < g.c2;
}

// The emit passes run over the ACTIVE cells only, packed: global active index i -> (workgroup b of the classify pass, rank
// k inside it) through the two-level exclusive scan of the per-workgroup active counts (act_ofs local to a group of 1024
// workgroups, gact_base per group).  "Largest index whose offset is <= i" lands on the non-empty workgroup among equals.
struct ActiveIndex {
    const unsigned *act_ofs, *gact_base;
    int nblocks, ngroups;
};
__device__ __forceinline__ void locate_active(const ActiveIndex &ai, unsigned i, int &b, int &k) {
    int lo = 0, hi = ai.ngroups;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ai.gact_base[mid] <= i) lo = mid; else hi = mid;
    }
    const unsigned r = i - ai.gact_base[lo];
    int b0 = lo << 10, b1 = min(ai.nblocks, b0 + 1024);
    while (b1 - b0 > 1) {
        const int mid = (b0 + b1) >> 1;
        if (ai.act_ofs[mid] <= r) b0 = mid; else b1 = mid;
    }
    b = b0;
    k = (int)(r - ai.act_ofs[b0]);
}
// first cell (x0, y, z) of classify workgroup b
__device__ __forceinline__ void origin_of_block(const Grid &g, int b, int &x0, int &y, int &z) {
    const int bpr = (g.c2 + MC_BLOCK - 1) / MC_BLOCK;
    const unsigned row = (unsigned)b / (unsigned)bpr;
    x0 = (b - (int)(row * (unsigned)bpr)) * MC_BLOCK;
    y = (int)(row % (unsigned)g.c1);
    z = (int)(row / (unsigned)g.c1);
}

// does cell (x,y,z) own edge e (is it the first cell, in sweep order, that touches it)?
__device__ __forceinline__ bool owns_edge(int e, int x, int y, int z, int halo_low) {
    if (halo_low && z == 0 && e < 4) return false;  // owned by the last cell layer of the previous slab
    switch (e) {
        case 0: return y == 0 && z == 0;
        case 1: return z == 0;
        case 2: return z == 0;
        case 3: return x == 0 && z == 0;
        case 4: return y == 0;
        case 5: return true;
        case 6: return true;
        case 7: return x == 0;
        case 8: return x == 0 && y == 0;
        case 9: return y == 0;
        case 10: return true;
        case 11: return x == 0;
        default: return true;  // 12: centre vertex
    }
}

// lattice-edge slot of edge e of cell (x,y,z): axis plane (0 = along x, 1 = along y, 2 = along z)
// and the lattice point the edge starts at
__device__ __forceinline__ long edge_slot(int e, int x, int y, int z, const Grid &g) {
    int axis, lx = x, ly = y, lz = z;
    switch (e) {
        case 0: axis = 0; break;
        case 1: axis = 1; lx += 1; break;
        case 2: axis = 0; ly += 1; break;
        case 3: axis = 1; break;
        case 4: axis = 0; lz += 1; break;
        case 5: axis = 1; lx += 1; lz += 1; break;
        case 6: axis = 0; ly += 1; lz += 1; break;
        case 7: axis = 1; lz += 1; break;
        case 8: axis = 2; break;
        case 9: axis = 2; lx += 1; break;
        case 10: axis = 2; lx += 1; ly += 1; break;
        default: axis = 2; ly += 1; break;  // 11
    }
    const long nvox = (long)g.n0 * g.n1 * g.n2;
    return axis * nvox + ((long)lz * g.n1 + ly) * g.n2 + lx;
}

// number of triangles / owned vertices of a classified cell, packed (ntri | nown << 16)
__device__ int cell_counts(const Tiling &t, int x, int y, int z, int halo_low) {
    if (t.len == 0) return 0;
    unsigned seen = 0;
    int nown = 0;
    for (int i = 0; i < t.len; ++i) {
        const int e = tiling_entry(t, i);
        if (!(seen >> e & 1u)) {
            seen |= 1u << e;
            nown += owns_edge(e, x, y, z, halo_low) ? 1 : 0;
        }
    }
    return (t.len / 3) | (nown << 16);
}

// exclusive scan of `val` over the 256-thread block; returns exclusive prefix, total in *total
__device__ int block_exclusive_scan(int val, int *total) {
    __shared__ int wsum[MC_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MC_BLOCK / 64; ++w) {
        const int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - val;
}

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float ord2f(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct McHeader {            // first 64 bytes of the workspace
    unsigned long long total_tri;
    unsigned long long total_vert;
    unsigned min_ord, max_ord;
    unsigned nan_seen;  // some corner value was NaN (set by the classify pass)
    unsigned total_active;  // number of active cells (= CellRec records) of the whole grid
    unsigned pad[8];
};

// One record per ACTIVE cell, stored compactly per workgroup (slot b*256 + rank, rank in cell order):
//   w0 = x_local | len << 8 | classic << 15 | ofs << 16      (the chosen tiling: no re-classification later)
//   w1 = exclusive prefix inside the workgroup of (ntri | nown << 16)
struct CellRec { unsigned w0, w1; };

// Pass 1 over the full grid: sign pattern of every cell (float compares only), ballot + popcount
// compaction of the active cells of the workgroup into LDS, then the expensive part (Lewiner face /
// interior tests in fp64, triangle and owned-vertex counts, their prefix) runs on DENSE lanes.
__global__ __launch_bounds__(MC_BLOCK) void mc_classify_kernel(const float *__restrict__ vol, Grid g, float levelf,
                                                               double level, int classic, CellRec *__restrict__ recs,
                                                               int *__restrict__ block_counts,
                                                               int *__restrict__ block_nact,
                                                               float2 *__restrict__ block_minmax,
                                                               McHeader *__restrict__ hdr) {
    __shared__ unsigned char s_list[MC_BLOCK];
    __shared__ int s_wcnt[MC_BLOCK / 64];
    __shared__ float s_mn[MC_BLOCK / 64], s_mx[MC_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mn = FLT_MAX, mx = -FLT_MAX;
    int x, y, z;
    bool active = false;
    const bool valid = cell_of_block(g, x, y, z);
    if (valid) {
        const long sy = g.n2, sz = (long)g.n1 * g.n2;
        const float *p = vol + z * sz + y * sy + x;
        const float f[8] = {p[0], p[1], p[sy + 1], p[sy], p[sz], p[sz + 1], p[sz + sy + 1], p[sz + sy]};
        int idx = 0;
        bool nan = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mn = fminf(mn, f[k]);
            mx = fmaxf(mx, f[k]);
            nan |= f[k] != f[k];
            idx |= (f[k] > levelf) ? (1 << k) : 0;  // == ((double)f - level > 0): levelf = largest float <= level
        }
        active = idx != 0 && idx != 255;
        if (nan) hdr->nan_seen = 1u;  // fminf/fmaxf drop NaN silently; a plain racing store of 1 is enough
    }
    const unsigned long long bal = __ballot(active);
    const int wrank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    int wbase = 0, nact = 0;
#pragma unroll
    for (int w = 0; w < MC_BLOCK / 64; ++w) {
        if (w < wave) wbase += s_wcnt[w];
        nact += s_wcnt[w];
    }
    if (nact == 0) {  // workgroup-uniform: most row segments are far from the surface -- no list, no scan
        if (threadIdx.x == 0) {
            for (int w = 0; w < MC_BLOCK / 64; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
            block_counts[blockIdx.x] = 0;
            block_nact[blockIdx.x] = 0;
            block_minmax[blockIdx.x] = make_float2(mn, mx);
        }
        return;
    }
    if (active) s_list[wbase + wrank] = (unsigned char)threadIdx.x;
    __syncthreads();
    // dense part
    int packed = 0;
    unsigned w0 = 0;
    if ((int)threadIdx.x < nact) {
        const int xl = s_list[threadIdx.x];
        const int cx = x - (int)threadIdx.x + xl;  // same row segment
        double v[8];
        load_cell(vol, g, z, y, cx, level, v);
        const Tiling t = classify(v, classic != 0);
        packed = cell_counts(t, cx, y, z, g.halo_low);
        const unsigned ofs = t.ofs < 0 ? (unsigned)(-t.ofs - 1) : (unsigned)t.ofs;
        w0 = (unsigned)xl | ((unsigned)t.len << 8) | ((t.ofs < 0 ? 1u : 0u) << 15) | (ofs << 16);
    }
    int total;
    const int pre = block_exclusive_scan(packed, &total);
    if ((int)threadIdx.x < nact) {
        CellRec r;
        r.w0 = w0; r.w1 = (unsigned)pre;
        recs[(long)blockIdx.x * MC_BLOCK + threadIdx.x] = r;
    }
    if (threadIdx.x == 0) {
        for (int w = 0; w < MC_BLOCK / 64; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
        block_counts[blockIdx.x] = total;
        block_nact[blockIdx.x] = nact;
        block_minmax[blockIdx.x] = make_float2(mn, mx);
    }
}

// two-level exclusive scan of the packed per-workgroup counts (tri low 16 | vert high 16 of an int):
// level 1: every workgroup scans 1024 entries and publishes its 64-bit total (tri | vert << 32)
__global__ __launch_bounds__(1024) void mc_scan1_kernel(const int *__restrict__ block_counts,
                                                        const float2 *__restrict__ block_minmax, int nblocks,
                                                        unsigned *__restrict__ tri_ofs, unsigned *__restrict__ vert_ofs,
                                                        unsigned long long *__restrict__ group_tot,
                                                        float2 *__restrict__ group_minmax,
                                                        const int *__restrict__ block_nact, unsigned *__restrict__ act_ofs,
                                                        unsigned *__restrict__ group_act) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned wasum[16];
    __shared__ float s_mn[16], s_mx[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int pk = i < nblocks ? block_counts[i] : 0;
    float mn = FLT_MAX, mx = -FLT_MAX;
    if (i < nblocks) { const float2 mm = block_minmax[i]; mn = mm.x; mx = mm.y; }
    const unsigned long long val = (unsigned long long)(pk & 0xffff) | ((unsigned long long)(pk >> 16) << 32);
    const unsigned na = i < nblocks ? (unsigned)block_nact[i] : 0u;
    unsigned long long inc = val;
    unsigned ainc = na;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long up = __shfl_up(inc, d, 64);
        const unsigned aup = __shfl_up(ainc, d, 64);
        if (lane >= d) { inc += up; ainc += aup; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 63) { wsum[wave] = inc; wasum[wave] = ainc; }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    unsigned long long wbase = 0, tot = 0;
    unsigned awbase = 0, atot = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { wbase += wsum[w]; awbase += wasum[w]; }
        tot += wsum[w];
        atot += wasum[w];
    }
    const unsigned long long ex = wbase + inc - val;
    if (i < nblocks) {
        tri_ofs[i] = (unsigned)(ex & 0xffffffffull);   // local to the group; the group base is added by the consumers
        vert_ofs[i] = (unsigned)(ex >> 32);
        act_ofs[i] = awbase + ainc - na;
    }
    if (threadIdx.x == 0) {
        for (int w = 0; w < 16; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
        group_tot[blockIdx.x] = tot;
        group_minmax[blockIdx.x] = make_float2(mn, mx);
        group_act[blockIdx.x] = atot;
    }
}

// level 2: one workgroup turns the group totals into exclusive group bases (in place) and the grand totals
__global__ __launch_bounds__(1024) void mc_scan2_kernel(unsigned long long *__restrict__ group_tot,
                                                        const float2 *__restrict__ group_minmax, int ngroups,
                                                        McHeader *__restrict__ hdr, unsigned *__restrict__ group_act) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    __shared__ unsigned wasum[16];
    __shared__ unsigned acarry_s;
    if (threadIdx.x == 0) acarry_s = 0;
    __shared__ float s_mn[16], s_mx[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int base = 0; base < ngroups; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned long long val = i < ngroups ? group_tot[i] : 0ull;
        const unsigned aval = i < ngroups ? group_act[i] : 0u;
        if (i < ngroups) { const float2 mm = group_minmax[i]; mn = fminf(mn, mm.x); mx = fmaxf(mx, mm.y); }
        unsigned long long inc = val;
        unsigned ainc = aval;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long up = __shfl_up(inc, d, 64);
            const unsigned aup = __shfl_up(ainc, d, 64);
            if (lane >= d) { inc += up; ainc += aup; }
        }
        if (lane == 63) { wsum[wave] = inc; wasum[wave] = ainc; }
        __syncthreads();
        unsigned long long wbase = carry_s, tot = 0;
        unsigned awbase = acarry_s, atot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { wbase += wsum[w]; awbase += wasum[w]; }
            tot += wsum[w];
            atot += wasum[w];
        }
        if (i < ngroups) { group_tot[i] = wbase + inc - val; group_act[i] = awbase + ainc - aval; }
        __syncthreads();
        if (threadIdx.x == 0) { carry_s += tot; acarry_s += atot; }
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 16; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
        hdr->total_tri = carry_s & 0xffffffffull;
        hdr->total_vert = carry_s >> 32;
        hdr->total_active = acarry_s;
        hdr->min_ord = f2ord(mn);
        hdr->max_ord = f2ord(mx);
    }
}

// weights of the two end points of an edge -> parametric position along the edge (double)
__device__ __forceinline__ double edge_frac(double v_near, double v_far) {
    const double t_near = 1.0 / (MC_EPS + fabs(v_near));
    const double t_far = 1.0 / (MC_EPS + fabs(v_far));
    return t_far / (t_near + t_far);
}

__device__ __forceinline__ Tiling tiling_of(unsigned w0) {
    Tiling t;
    t.len = (int)((w0 >> 8) & 0x7f);
    const int ofs = (int)(w0 >> 16);
    t.ofs = (w0 >> 15 & 1u) ? -(ofs + 1) : ofs;
    return t;
}

// Pass 2, active cells only (dense lanes): the owner of every crossing edge writes the vertex and the
// lattice-edge -> id map.
__global__ __launch_bounds__(MC_BLOCK) void mc_verts_kernel(const float *__restrict__ vol, Grid g, double level,
                                                            const CellRec *__restrict__ recs, ActiveIndex ai,
                                                            const McHeader *__restrict__ hdr,
                                                            const unsigned *__restrict__ vert_ofs,
                                                            const unsigned long long *__restrict__ group_base,
                                                            int *__restrict__ edge_map, float *__restrict__ verts,
                                                            float vdiv, float vmul, float vadd, int affine) {
  const unsigned total_active = hdr->total_active;
  for (unsigned ci = blockIdx.x * MC_BLOCK + threadIdx.x; ci < total_active; ci += gridDim.x * MC_BLOCK) {
    int b, k, x, y, z;
    locate_active(ai, ci, b, k);
    origin_of_block(g, b, x, y, z);
    const CellRec rec = recs[(long)b * MC_BLOCK + k];
    x += (int)(rec.w0 & 0xffu);
    const Tiling t = tiling_of(rec.w0);
    double v[8];
    load_cell(vol, g, z, y, x, level, v);
    unsigned id = (unsigned)(group_base[b >> 10] >> 32) + vert_ofs[b] + (rec.w1 >> 16);
    unsigned seen = 0;
    for (int i = 0; i < t.len; ++i) {
        const int e = tiling_entry(t, i);
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (!owns_edge(e, x, y, z, g.halo_low)) continue;
        double px, py, pz;  // skimage's internal (x,y,z) = (axis2, axis1, axis0)
        if (e == 12) {
            // centre vertex: inverse-|value| weighted mean of the 8 corners, summed in corner order
            double w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = 1.0 / (MC_EPS + fabs(v[k]));
            double ff = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) ff = ff + w[k];
            const double fx = ((w[1] + w[2]) + w[5]) + w[6];
            const double fy = ((w[2] + w[3]) + w[6]) + w[7];
            const double fz = ((w[4] + w[5]) + w[6]) + w[7];
            px = (double)x + fx / ff;
            py = (double)y + fy / ff;
            pz = (double)(z + g.z_off) + fz / ff;
        } else {
            // corner pairs (near = lower lattice coordinate along the edge axis, far = near+1)
            int cn, cf, lx = x, ly = y, lz = z, axis;
            switch (e) {
                case 0: cn = 0; cf = 1; axis = 0; break;
                case 1: cn = 1; cf = 2; axis = 1; lx += 1; break;
                case 2: cn = 3; cf = 2; axis = 0; ly += 1; break;
                case 3: cn = 0; cf = 3; axis = 1; break;
                case 4: cn = 4; cf = 5; axis = 0; lz += 1; break;
                case 5: cn = 5; cf = 6; axis = 1; lx += 1; lz += 1; break;
                case 6: cn = 7; cf = 6; axis = 0; ly += 1; lz += 1; break;
                case 7: cn = 4; cf = 7; axis = 1; lz += 1; break;
                case 8: cn = 0; cf = 4; axis = 2; break;
                case 9: cn = 1; cf = 5; axis = 2; lx += 1; break;
                case 10: cn = 2; cf = 6; axis = 2; lx += 1; ly += 1; break;
                default: cn = 3; cf = 7; axis = 2; ly += 1; break;
            }
            const double fr = edge_frac(v[cn], v[cf]);
            // cell origin + (1.0 or fr): the off-axis offsets are exactly 0.0 or 1.0
            px = (axis == 0) ? (double)x + fr : (double)lx;
            py = (axis == 1) ? (double)y + fr : (double)ly;
            pz = (axis == 2) ? (double)(z + g.z_off) + fr : (double)(lz + g.z_off);
            edge_map[edge_slot(e, x, y, z, g)] = (int)id;
        }
        // output columns (axis0, axis1, axis2) = (z, y, x): skimage's fliplr of its (x,y,z)
        float o0 = (float)pz, o1 = (float)py, o2 = (float)px;
        if (affine) {
            o0 = o0 / vdiv; o1 = o1 / vdiv; o2 = o2 / vdiv;               // v_pos / (R - 1)  isosurface.py:53
            o0 = rounded(o0 * vmul) + vadd;                                // scale_tensor    system.py:185-189
            o1 = rounded(o1 * vmul) + vadd;
            o2 = rounded(o2 * vmul) + vadd;
        }
        verts[3 * (size_t)id + 0] = o0;
        verts[3 * (size_t)id + 1] = o1;
        verts[3 * (size_t)id + 2] = o2;
        ++id;
    }
  }
}

// Pass 3, active cells only: triangles through the lattice-edge -> id map (no volume reads at all).
template <typename IdxT>
__global__ __launch_bounds__(MC_BLOCK) void mc_faces_kernel(Grid g, const CellRec *__restrict__ recs, ActiveIndex ai,
                                                            const McHeader *__restrict__ hdr,
                                                            const unsigned *__restrict__ tri_ofs,
                                                            const unsigned *__restrict__ vert_ofs,
                                                            const unsigned long long *__restrict__ group_base,
                                                            const int *__restrict__ edge_map, IdxT *__restrict__ faces,
                                                            int ref_order) {
  const unsigned total_active = hdr->total_active;
  for (unsigned ci = blockIdx.x * MC_BLOCK + threadIdx.x; ci < total_active; ci += gridDim.x * MC_BLOCK) {
    int b, k, x, y, z;
    locate_active(ai, ci, b, k);
    origin_of_block(g, b, x, y, z);
    const CellRec rec = recs[(long)b * MC_BLOCK + k];
    x += (int)(rec.w0 & 0xffu);
    const Tiling t = tiling_of(rec.w0);
    const unsigned long long gb = group_base[b >> 10];
    const unsigned tri0 = (unsigned)(gb & 0xffffffffull) + tri_ofs[b] + (rec.w1 & 0xffffu);
    const unsigned vown0 = (unsigned)(gb >> 32) + vert_ofs[b] + (rec.w1 >> 16);
    // id of the centre vertex = own base + rank among owned vertices in first-appearance order
    int centre_id = -1;
    {
        unsigned seen = 0;
        int rank = 0;
        for (int i = 0; i < t.len; ++i) {
            const int e = tiling_entry(t, i);
            if (seen >> e & 1u) continue;
            seen |= 1u << e;
            if (e == 12) { centre_id = (int)vown0 + rank; break; }
            rank += owns_edge(e, x, y, z, g.halo_low) ? 1 : 0;
        }
    }
    const int ntri = t.len / 3;
    for (int k = 0; k < ntri; ++k) {
        int id[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = tiling_entry(t, 3 * k + j);
            if (e == 12) id[j] = centre_id;
            else if (g.halo_low && z == 0 && e < 4) {
                // vertex lives in the previous slab: encode the edge of lattice plane 0 as -(1 + slot),
                // slot = axis*(n1*n2) + ly*n2 + lx; resolved after the gather (sculptmate_amd/slab.py)
                const int axis = e & 1, lx = x + (e == 1), ly = y + (e == 2);
                id[j] = -(1 + axis * g.n1 * g.n2 + ly * g.n2 + lx);
            } else id[j] = edge_map[edge_slot(e, x, y, z, g)];
        }
        // internal (a,b,c); skimage 'descent' flips to (c,b,a); the reference then takes [1,0,2] -> (b,c,a)
        IdxT *f = faces + 3 * (size_t)(tri0 + k);
        if (ref_order) { f[0] = (IdxT)id[1]; f[1] = (IdxT)id[2]; f[2] = (IdxT)id[0]; }
        else { f[0] = (IdxT)id[2]; f[1] = (IdxT)id[1]; f[2] = (IdxT)id[0]; }
    }
  }
}

// lattice-edge -> vertex-id map of the LAST lattice plane (x and y edges), -1 where the edge has no crossing
__global__ __launch_bounds__(256) void mc_top_plane_kernel(const float *__restrict__ vol, Grid g, double level,
                                                           const int *__restrict__ edge_map, int *__restrict__ out) {
    const int plane = g.n1 * g.n2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * plane) return;
    const int axis = i / plane, rem = i % plane, ly = rem / g.n2, lx = rem % g.n2;
    const int lz = g.n0 - 1;
    const int lx2 = lx + (axis == 0), ly2 = ly + (axis == 1);
    int v = -1;
    if (lx2 < g.n2 && ly2 < g.n1) {
        const float *p = vol + (long)lz * plane;
        const bool a = ((double)p[ly * g.n2 + lx] - level) > 0.0, b = ((double)p[ly2 * g.n2 + lx2] - level) > 0.0;
        if (a != b) v = edge_map[(long)axis * g.n0 * plane + (long)lz * plane + ly * g.n2 + lx];
    }
    out[i] = v;
}

static int make_grid(int n0, int n1, int n2, Grid *g) {
    SC_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "marching_cubes: input array must be at least 2x2x2");
    g->n0 = n0; g->n1 = n1; g->n2 = n2;
    g->c0 = n0 - 1; g->c1 = n1 - 1; g->c2 = n2 - 1;
    g->ncells = (long)g->c0 * g->c1 * g->c2;
    g->halo_low = 0;
    g->z_off = 0;
    SC_REQUIRE((long)g->c0 * g->c1 * cdiv(g->c2, MC_BLOCK) < 0x7fffffffL, "marching_cubes: volume too large");
    return 0;
}

struct WsLayout {
    size_t off_counts, off_nact, off_minmax, off_tri, off_vert, off_gtot, off_gmm, off_recs, off_map, off_aofs, off_gact, total;
    int ngroups;
    int nblocks;
};
static WsLayout ws_layout(const Grid &g) {
    WsLayout w;
    w.nblocks = g.c0 * g.c1 * cdiv(g.c2, MC_BLOCK);
    size_t o = 64;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    w.ngroups = cdiv(w.nblocks, 1024);
    w.off_counts = o; o = al(o + sizeof(int) * w.nblocks);
    w.off_nact = o;   o = al(o + sizeof(int) * w.nblocks);
    w.off_minmax = o; o = al(o + sizeof(float2) * w.nblocks);
    w.off_gtot = o;   o = al(o + sizeof(unsigned long long) * w.ngroups);
    w.off_gmm = o;    o = al(o + sizeof(float2) * w.ngroups);
    w.off_recs = o;   o = al(o + sizeof(CellRec) * (size_t)w.nblocks * MC_BLOCK);
    w.off_tri = o;    o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_vert = o;   o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_map = o;    o = al(o + sizeof(int) * 3 * (size_t)g.n0 * g.n1 * g.n2);
    w.off_aofs = o;   o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_gact = o;   o = al(o + sizeof(unsigned) * w.ngroups);
    w.total = o;
    return w;
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

size_t sculpt_mc_workspace_bytes(int n0, int n1, int n2) {
    Grid g;
    if (make_grid(n0, n1, n2, &g)) return 0;
    return ws_layout(g).total;
}

int sculpt_mc_count(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                    int64_t *n_verts_host, int64_t *n_faces_host, float *minmax_host, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    Grid g;
    if (int rc = make_grid(n0, n1, n2, &g)) return rc;
    g.halo_low = (flags & SCULPT_MC_SLAB_HALO_LOW) ? 1 : 0;
    SC_REQUIRE(vol && workspace && n_verts_host && n_faces_host, "mc_count: null argument");
    if (int rc = upload_tables()) return rc;
    const WsLayout w = ws_layout(g);
    char *ws = reinterpret_cast<char *>(workspace);
    McHeader *hdr = reinterpret_cast<McHeader *>(ws);
    McHeader init;
    memset(&init, 0, sizeof(init));
    init.min_ord = 0xffffffffu;
    init.max_ord = 0u;
    SC_HIP(hipMemcpyAsync(hdr, &init, sizeof(init), hipMemcpyHostToDevice, st));
    const int classic = (flags & SCULPT_MC_USE_CLASSIC) ? 1 : 0;
    // float f > double level  <=>  f > (largest float <= level): the sign pass needs no fp64
    float levelf = (float)level;
    if ((double)levelf > level) levelf = nextafterf(levelf, -INFINITY);
    hipLaunchKernelGGL(mc_classify_kernel, dim3(w.nblocks), dim3(MC_BLOCK), 0, st, vol, g, levelf, level, classic,
                       reinterpret_cast<CellRec *>(ws + w.off_recs), reinterpret_cast<int *>(ws + w.off_counts),
                       reinterpret_cast<int *>(ws + w.off_nact), reinterpret_cast<float2 *>(ws + w.off_minmax), hdr);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_scan1_kernel, dim3(w.ngroups), dim3(1024), 0, st, reinterpret_cast<const int *>(ws + w.off_counts),
                       reinterpret_cast<const float2 *>(ws + w.off_minmax), w.nblocks,
                       reinterpret_cast<unsigned *>(ws + w.off_tri), reinterpret_cast<unsigned *>(ws + w.off_vert),
                       reinterpret_cast<unsigned long long *>(ws + w.off_gtot), reinterpret_cast<float2 *>(ws + w.off_gmm),
                       reinterpret_cast<const int *>(ws + w.off_nact), reinterpret_cast<unsigned *>(ws + w.off_aofs),
                       reinterpret_cast<unsigned *>(ws + w.off_gact));
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_scan2_kernel, dim3(1), dim3(1024), 0, st, reinterpret_cast<unsigned long long *>(ws + w.off_gtot),
                       reinterpret_cast<const float2 *>(ws + w.off_gmm), w.ngroups, hdr,
                       reinterpret_cast<unsigned *>(ws + w.off_gact));
    SC_LAUNCH_CHECK();
    McHeader res;
    SC_HIP(hipMemcpyAsync(&res, hdr, sizeof(res), hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    *n_verts_host = (int64_t)res.total_vert;
    *n_faces_host = (int64_t)res.total_tri;
    // skimage: "Surface level must be within volume data range." (ValueError)
    const float mn = ord2f(res.min_ord), mx = ord2f(res.max_ord);
    if (minmax_host) { minmax_host[0] = mn; minmax_host[1] = mx; }
    if (res.nan_seen) {
        set_error("marching_cubes: the volume contains NaN");
        return SCULPT_ERR_MC_NAN;
    }
    if (flags & SCULPT_MC_SLAB) return 0;  // a slab may be empty; the caller decides globally
    if ((double)level < (double)mn || (double)level > (double)mx) {
        set_error("Surface level must be within volume data range.");
        return SCULPT_ERR_MC_LEVEL;
    }
    if (res.total_vert == 0) {
        set_error("No surface found at the given iso value.");
        return SCULPT_ERR_MC_EMPTY;
    }
    SC_REQUIRE(res.total_vert < 0x7fffffffull && res.total_tri < 0x55555555ull, "marching_cubes: mesh too large for 32-bit ids");
    return 0;
}

int sculpt_mc_emit(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                   float vert_div, float vert_mul, float vert_add, int axis0_offset, float *verts, void *faces,
                   int *top_plane_map, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    Grid g;
    if (int rc = make_grid(n0, n1, n2, &g)) return rc;
    g.halo_low = (flags & SCULPT_MC_SLAB_HALO_LOW) ? 1 : 0;
    g.z_off = axis0_offset;
    SC_REQUIRE(vol && workspace && verts && faces, "mc_emit: null argument");
    const WsLayout w = ws_layout(g);
    char *ws = reinterpret_cast<char *>(workspace);
    const int classic = (flags & SCULPT_MC_USE_CLASSIC) ? 1 : 0;
    const int ref = (flags & SCULPT_MC_REFERENCE_ORDER) ? 1 : 0;
    const unsigned *tri = reinterpret_cast<const unsigned *>(ws + w.off_tri);
    const unsigned *vrt = reinterpret_cast<const unsigned *>(ws + w.off_vert);
    int *emap = reinterpret_cast<int *>(ws + w.off_map);
    const CellRec *recs = reinterpret_cast<const CellRec *>(ws + w.off_recs);
    const McHeader *hdr = reinterpret_cast<const McHeader *>(ws);
    const ActiveIndex ai{reinterpret_cast<const unsigned *>(ws + w.off_aofs), reinterpret_cast<const unsigned *>(ws + w.off_gact),
                         w.nblocks, w.ngroups};
    const unsigned long long *gbase = reinterpret_cast<const unsigned long long *>(ws + w.off_gtot);
    // packed active cells, grid-stride (the count lives in the workspace header: no second read-back)
    const int egrid = std::min(w.nblocks, num_cus() * 16);
    hipLaunchKernelGGL(mc_verts_kernel, dim3(egrid), dim3(MC_BLOCK), 0, st, vol, g, (double)level, recs, ai, hdr, vrt,
                       gbase, emap, verts, vert_div, vert_mul, vert_add, ref);
    SC_LAUNCH_CHECK();
    if (flags & SCULPT_MC_FACES_I64)
        hipLaunchKernelGGL(mc_faces_kernel<long long>, dim3(egrid), dim3(MC_BLOCK), 0, st, g, recs, ai, hdr, tri, vrt,
                           gbase, emap, reinterpret_cast<long long *>(faces), ref);
    else
        hipLaunchKernelGGL(mc_faces_kernel<int>, dim3(egrid), dim3(MC_BLOCK), 0, st, g, recs, ai, hdr, tri, vrt, gbase,
                           emap, reinterpret_cast<int *>(faces), ref);
    SC_LAUNCH_CHECK();
    if (top_plane_map) {
        hipLaunchKernelGGL(mc_top_plane_kernel, dim3(cdiv(2L * n1 * n2, 256)), dim3(256), 0, st, vol, g, (double)level,
                           emap, top_plane_map);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
