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
//   emit  : active cells only (one record per active cell from the count pass), one workgroup per brick of rows: owners write
//           vertex positions, every cell writes its triangles; the id of a vertex on edge e is the OWNER's vertex base + the
//           rank of e among the owner's vertices, read from the owner's record (a neighbour cell at offset {0,-1}^3, found
//           through the rows' chunk tables, in LDS): no lattice-wide edge -> id map (round 6; it was int32 [R^3][4], 268 MB at 256^3)
// Workspace: per-row arrays (counts, offsets, 256-bit active masks) + 8 bytes per ACTIVE cell in a pool whose capacity the caller
// chooses (sculpt_mc_workspace_bytes_for); a count pass that runs out of pool reports SCULPT_ERR_MC_WORKSPACE and the number of
// active cells, and the caller repeats it with a larger workspace.  23 MB at 256^3 and 181 MB at 512^3 with the default pool
// (one active cell per 8 cells), where the dense records + map of round 5 took 403 MB and 3.2 GB.
#include <float.h>
#include <math.h>
#include <string.h>

#include <stdlib.h>

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
// the same table with two entries per byte (every entry is an edge number 0..12): the emit pass keeps it in LDS
static constexpr int MC_TNIB16 = (MC_TILING_FLAT_SIZE / 2 + 1 + 15) / 16;
__device__ __attribute__((aligned(16))) unsigned char d_tiling_nib[MC_TNIB16 * 16];
__constant__ unsigned short d_tiling_base[MC_NUM_TILINGS];
__constant__ unsigned char d_tiling_rowlen[MC_NUM_TILINGS];
__constant__ unsigned char d_tiling_inner[MC_NUM_TILINGS];
__constant__ signed char d_cases[512];
__constant__ signed char d_cases_classic[256 * 16];
__constant__ signed char d_test3[24], d_test4[8], d_test6[48 * 3], d_test7[16 * 5], d_test10[6 * 3],
    d_test12[24 * 4], d_test13[2 * 7], d_subconfig13[64];

enum {
    LUT_CASES = 0, LUT_TEST3 = 512, LUT_TEST4 = 536, LUT_TEST6 = 544, LUT_TEST7 = 688, LUT_TEST10 = 768, LUT_TEST12 = 786,
    LUT_TEST13 = 882, LUT_SUB13 = 896, LUT_ROWLEN = 960, LUT_INNER = 998, LUT_T1351 = 1036, LUT_BASE = 1048,
    LUT_ROWBASE = 1124, LUT_ROWMASK = 1200, LUT_MAX_ROWS = 768, LUT_ROWRANK = LUT_ROWMASK + 2 * LUT_MAX_ROWS,
    LUT_BYTES = LUT_ROWRANK + LUT_MAX_ROWS
};
__device__ __attribute__((aligned(16))) unsigned char d_lut_blob[LUT_BYTES];

static bool g_tables_uploaded[64] = {false};

static int upload_tables() {
    int dev = 0;
    SC_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && g_tables_uploaded[dev]) return 0;
#define UP(sym, src) SC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sym), src, sizeof(src)))
    UP(d_tiling_flat, mc_tiling_flat);
    {
        static unsigned char nib[MC_TNIB16 * 16];
        memset(nib, 0, sizeof(nib));
        for (int i = 0; i < MC_TILING_FLAT_SIZE; ++i) {
            SC_REQUIRE(mc_tiling_flat[i] >= 0 && mc_tiling_flat[i] < 16, "marching_cubes: tiling entry %d out of range", (int)mc_tiling_flat[i]);
            nib[i >> 1] |= (unsigned char)(mc_tiling_flat[i] << ((i & 1) * 4));
        }
        SC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_tiling_nib), nib, sizeof(nib)));
    }
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
    {   // the LDS blob of the classify pass: small tables + per-row edge masks (bit e = the row's triangles use edge e)
        static unsigned char blob[LUT_BYTES];
        memset(blob, 0, sizeof(blob));
        memcpy(blob + LUT_CASES, mc_cases, 512);
        memcpy(blob + LUT_TEST3, mc_test3, sizeof(mc_test3));
        memcpy(blob + LUT_TEST4, mc_test4, sizeof(mc_test4));
        memcpy(blob + LUT_TEST6, mc_test6, sizeof(mc_test6));
        memcpy(blob + LUT_TEST7, mc_test7, sizeof(mc_test7));
        memcpy(blob + LUT_TEST10, mc_test10, sizeof(mc_test10));
        memcpy(blob + LUT_TEST12, mc_test12, sizeof(mc_test12));
        memcpy(blob + LUT_TEST13, mc_test13, sizeof(mc_test13));
        memcpy(blob + LUT_SUB13, mc_subconfig13, sizeof(mc_subconfig13));
        static_assert(sizeof(mc_test3) == 24 && sizeof(mc_test4) == 8 && sizeof(mc_test6) == 144 && sizeof(mc_test7) == 80 &&
                          sizeof(mc_test10) == 18 && sizeof(mc_test12) == 96 && sizeof(mc_test13) == 14 && sizeof(mc_subconfig13) == 64,
                      "LUT blob layout");
        unsigned short *base = reinterpret_cast<unsigned short *>(blob + LUT_BASE);
        unsigned short *rowbase = reinterpret_cast<unsigned short *>(blob + LUT_ROWBASE);
        unsigned short *rowmask = reinterpret_cast<unsigned short *>(blob + LUT_ROWMASK);
        unsigned char *rowrank = blob + LUT_ROWRANK;
        int nrows = 0;
        for (int t = 0; t < MC_NUM_TILINGS; ++t) {
            blob[LUT_ROWLEN + t] = mc_tiling_rowlen[t];
            blob[LUT_INNER + t] = mc_tiling_inner[t];
            base[t] = mc_tiling_base[t];
            rowbase[t] = (unsigned short)nrows;
            const int end = (t + 1 < MC_NUM_TILINGS) ? mc_tiling_base[t + 1] : MC_TILING_FLAT_SIZE;
            const int rows = (end - mc_tiling_base[t]) / mc_tiling_rowlen[t];
            for (int r = 0; r < rows; ++r) {
                unsigned m = 0;
                for (int i = 0; i < mc_tiling_rowlen[t]; ++i) m |= 1u << mc_tiling_flat[mc_tiling_base[t] + r * mc_tiling_rowlen[t] + i];
                SC_REQUIRE(nrows < LUT_MAX_ROWS && m < 0x2000u, "marching_cubes: tiling tables do not fit the LUT blob");
                // An INTERIOR cell (x, y, z > 0) owns exactly the vertices on edges 5, 6, 10 and the centre vertex 12 (owns_edge):
                // the rank of each of the three edges among those four, in first-appearance order of the row's triangle list --
                // what a neighbour needs, beside the owner's vertex base, to name the vertex (edge_vertex_id).  2 bits each.
                {
                    unsigned seen = 0, rk = 0;
                    int n_owned = 0;
                    for (int i = 0; i < mc_tiling_rowlen[t]; ++i) {
                        const int e = mc_tiling_flat[mc_tiling_base[t] + r * mc_tiling_rowlen[t] + i];
                        if (seen >> e & 1u) continue;
                        seen |= 1u << e;
                        if (e == 5) rk |= (unsigned)n_owned;
                        if (e == 6) rk |= (unsigned)n_owned << 2;
                        if (e == 10) rk |= (unsigned)n_owned << 4;
                        if (e == 5 || e == 6 || e == 10 || e == 12) ++n_owned;
                    }
                    rowrank[nrows] = (unsigned char)rk;
                }
                rowmask[nrows++] = (unsigned short)m;
            }
        }
        for (int i = 0; i < 8; ++i) blob[LUT_T1351 + i] = (unsigned char)mc_tiling_flat[mc_tiling_base[MC_T_13_5_1] + i * 18];
        SC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_lut_blob), blob, sizeof(blob)));
    }
    if (dev >= 0 && dev < 64) g_tables_uploaded[dev] = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The small look-up tables of the classification, two ways: straight from __constant__ memory (TabConst), or from a
// 3.4 KiB blob staged in LDS (TabLds, the classify pass): lanes index the tables with different cells' cases, so a
// __constant__ access is a per-lane vector load with an L2 round trip, and a cell walks through 5-10 DEPENDENT ones.
// Blob layout (bytes): the tables back to back + per-tiling-row edge masks (see build_lut_blob).
// ---------------------------------------------------------------------------------------------
struct TabConst {
    __device__ int cases(int i) const { return d_cases[i]; }
    __device__ int test3(int i) const { return d_test3[i]; }
    __device__ int test4(int i) const { return d_test4[i]; }
    __device__ int test6(int i) const { return d_test6[i]; }
    __device__ int test7(int i) const { return d_test7[i]; }
    __device__ int test10(int i) const { return d_test10[i]; }
    __device__ int test12(int i) const { return d_test12[i]; }
    __device__ int test13(int i) const { return d_test13[i]; }
    __device__ int sub13(int i) const { return d_subconfig13[i]; }
    __device__ int rowlen(int t) const { return d_tiling_rowlen[t]; }
    __device__ int inner(int t) const { return d_tiling_inner[t]; }
    __device__ int base(int t) const { return d_tiling_base[t]; }
    __device__ int t1351(int i) const { return d_tiling_flat[d_tiling_base[MC_T_13_5_1] + i * 18]; }
};
struct TabLds {
    const unsigned char *b;  // LDS copy of d_lut_blob
    __device__ int sc(int o) const { return (int)(signed char)b[o]; }
    __device__ int cases(int i) const { return sc(LUT_CASES + i); }
    __device__ int test3(int i) const { return sc(LUT_TEST3 + i); }
    __device__ int test4(int i) const { return sc(LUT_TEST4 + i); }
    __device__ int test6(int i) const { return sc(LUT_TEST6 + i); }
    __device__ int test7(int i) const { return sc(LUT_TEST7 + i); }
    __device__ int test10(int i) const { return sc(LUT_TEST10 + i); }
    __device__ int test12(int i) const { return sc(LUT_TEST12 + i); }
    __device__ int test13(int i) const { return sc(LUT_TEST13 + i); }
    __device__ int sub13(int i) const { return sc(LUT_SUB13 + i); }
    __device__ int rowlen(int t) const { return b[LUT_ROWLEN + t]; }
    __device__ int inner(int t) const { return b[LUT_INNER + t]; }
    __device__ int base(int t) const { return reinterpret_cast<const unsigned short *>(b + LUT_BASE)[t]; }
    __device__ int t1351(int i) const { return sc(LUT_T1351 + i); }
    __device__ int rowbase(int t) const { return reinterpret_cast<const unsigned short *>(b + LUT_ROWBASE)[t]; }
    __device__ unsigned rowmask(int row) const { return reinterpret_cast<const unsigned short *>(b + LUT_ROWMASK)[row]; }
    __device__ unsigned rowrank(int row) const { return b[LUT_ROWRANK + row]; }
};

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

template <class TAB>
__device__ bool test_internal(const TAB &T, const double *v, int mc_case, int config, int subconfig, int s) {
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
        if (mc_case == 6) edge = T.test6(config * 3 + 2);
        else if (mc_case == 7) edge = T.test7(config * 5 + 4);
        else if (mc_case == 12) edge = T.test12(config * 4 + 3);
        else edge = T.t1351(config * 4 + subconfig);
        // reference edge e from corner a to corner b; the three "parallel" edges (p0,p1),(q0,q1),(r0,r1).
        // The corner VALUES are picked inside the switch (static indices only): with v[ea] ... v[r1] indexed by run-time
        // corner numbers the compiler keeps v[] in scratch memory, and every access of the classification -- the static
        // ones too -- becomes a ~1 us scratch round trip (the classify pass spent 100 of its 190 us there).
        double Ea, Eb, P0, P1, Q0, Q1, R0, R1;
#define PICK(a, b, c, d, e, f, gq, h) Ea = v[a]; Eb = v[b]; P0 = v[c]; P1 = v[d]; Q0 = v[e]; Q1 = v[f]; R0 = v[gq]; R1 = v[h]
        switch (edge) {
            case 0: PICK(0, 1, 3, 2, 7, 6, 4, 5); break;
            case 1: PICK(1, 2, 0, 3, 4, 7, 5, 6); break;
            case 2: PICK(2, 3, 1, 0, 5, 4, 6, 7); break;
            case 3: PICK(3, 0, 2, 1, 6, 5, 7, 4); break;
            case 4: PICK(4, 5, 7, 6, 3, 2, 0, 1); break;
            case 5: PICK(5, 6, 4, 7, 0, 3, 1, 2); break;
            case 6: PICK(6, 7, 5, 4, 1, 0, 2, 3); break;
            case 7: PICK(7, 4, 6, 5, 2, 1, 3, 0); break;
            case 8: PICK(0, 4, 3, 7, 2, 6, 1, 5); break;
            case 9: PICK(1, 5, 0, 4, 3, 7, 2, 6); break;
            case 10: PICK(2, 6, 1, 5, 0, 4, 3, 7); break;
            default: PICK(3, 7, 2, 6, 1, 5, 0, 4); break;  // 11
        }
#undef PICK
        t = Ea / ADD(SUB(Ea, Eb), MC_EPS);
        At = 0;
        Bt = ADD(P0, MUL(SUB(P1, P0), t));
        Ct = ADD(Q0, MUL(SUB(Q1, Q0), t));
        Dt = ADD(R0, MUL(SUB(R1, R0), t));
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

// row_out (optional): index of the chosen tiling row among all rows of all tables (TabLds::rowmask), -1 for classic
template <class TAB>
__device__ Tiling classify(const TAB &T, const double *v, bool classic, int *row_out = nullptr) {
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
        if (row_out) *row_out = -1;
        return out;
    }
    const int c = T.cases(2 * index), cfg = T.cases(2 * index + 1);
    int table = -1, sub = 0, sc = 0;
    switch (c) {
        case 1: table = MC_T_1; break;
        case 2: table = MC_T_2; break;
        case 3: table = test_face(v, T.test3(cfg)) ? MC_T_3_2 : MC_T_3_1; break;
        case 4: table = test_internal(T, v, c, cfg, 0, T.test4(cfg)) ? MC_T_4_1 : MC_T_4_2; break;
        case 5: table = MC_T_5; break;
        case 6:
            if (test_face(v, T.test6(cfg * 3 + 0))) table = MC_T_6_2;
            else table = test_internal(T, v, c, cfg, 0, T.test6(cfg * 3 + 1)) ? MC_T_6_1_1 : MC_T_6_1_2;
            break;
        case 7:
            if (test_face(v, T.test7(cfg * 5 + 0))) sc += 1;
            if (test_face(v, T.test7(cfg * 5 + 1))) sc += 2;
            if (test_face(v, T.test7(cfg * 5 + 2))) sc += 4;
            switch (sc) {
                case 0: table = MC_T_7_1; break;
                case 1: table = MC_T_7_2; sub = 0; break;
                case 2: table = MC_T_7_2; sub = 1; break;
                case 3: table = MC_T_7_3; sub = 0; break;
                case 4: table = MC_T_7_2; sub = 2; break;
                case 5: table = MC_T_7_3; sub = 1; break;
                case 6: table = MC_T_7_3; sub = 2; break;
                default: table = test_internal(T, v, c, cfg, sc, T.test7(cfg * 5 + 3)) ? MC_T_7_4_2 : MC_T_7_4_1; break;
            }
            break;
        case 8: table = MC_T_8; break;
        case 9: table = MC_T_9; break;
        case 10:
            if (test_face(v, T.test10(cfg * 3 + 0))) table = test_face(v, T.test10(cfg * 3 + 1)) ? MC_T_10_1_1_ : MC_T_10_2;
            else if (test_face(v, T.test10(cfg * 3 + 1))) table = MC_T_10_2_;
            else table = test_internal(T, v, c, cfg, 0, T.test10(cfg * 3 + 2)) ? MC_T_10_1_1 : MC_T_10_1_2;
            break;
        case 11: table = MC_T_11; break;
        case 12:
            if (test_face(v, T.test12(cfg * 4 + 0))) table = test_face(v, T.test12(cfg * 4 + 1)) ? MC_T_12_1_1_ : MC_T_12_2;
            else if (test_face(v, T.test12(cfg * 4 + 1))) table = MC_T_12_2_;
            else table = test_internal(T, v, c, cfg, 0, T.test12(cfg * 4 + 2)) ? MC_T_12_1_1 : MC_T_12_1_2;
            break;
        case 13:
            for (int k = 0; k < 6; ++k)
                if (test_face(v, T.test13(cfg * 7 + k))) sc += 1 << k;
            sc = T.sub13(sc);
            if (sc == 0) table = MC_T_13_1;
            else if (sc <= 6) { table = MC_T_13_2; sub = sc - 1; }
            else if (sc <= 18) { table = MC_T_13_3; sub = sc - 7; }
            else if (sc <= 22) { table = MC_T_13_4; sub = sc - 19; }
            else if (sc <= 26) {
                sub = sc - 23;
                table = test_internal(T, v, c, cfg, sub, T.test13(cfg * 7 + 6)) ? MC_T_13_5_1 : MC_T_13_5_2;
            } else if (sc <= 38) { table = MC_T_13_3_; sub = sc - 27; }
            else if (sc <= 44) { table = MC_T_13_2_; sub = sc - 39; }
            else if (sc == 45) table = MC_T_13_1_;
            break;
        case 14: table = MC_T_14; break;
        default: break;
    }
    if (table < 0) return out;
    const int rowlen = T.rowlen(table);
    const int r = cfg * T.inner(table) + sub;
    out.ofs = T.base(table) + r * rowlen;
    out.len = rowlen;
    if (row_out) *row_out = table * 1024 + r;  // (table, row in table); TabLds turns it into a global row index
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

// The lattice edge that is edge e (0..11) of cell (x, y, z): its axis (0 = along x, 1 = along y, 2 = along z) and the lattice point
// it starts at.
__device__ __forceinline__ void lattice_edge_of(int e, int x, int y, int z, int &axis, int &lx, int &ly, int &lz) {
    // edges 0-7: x / y alternating (bottom 0-3, top 4-7), 8-11: z;  +1 in x for 1, 5, 9, 10; in y for 2, 6, 10, 11; in z for 4-7
    axis = e < 8 ? (e & 1) : 2;
    lx = x + (int)((0x622u >> e) & 1u);
    ly = y + (int)((0xC44u >> e) & 1u);
    lz = z + (int)((0x0F0u >> e) & 1u);
}

// bit e set: cell (x, y, z) owns the vertex on its edge e (owns_edge as a mask; bit 12 = the centre vertex)
__device__ __forceinline__ unsigned owned_mask(int x, int y, int z, int halo_low) {
    unsigned own = 0x1460u;                               // edges 5, 6, 10 and the centre vertex 12: always
    if (x == 0) own |= 0x0880u;                           // 7, 11
    if (y == 0) own |= 0x0210u;                           // 4, 9
    if (z == 0) own |= 0x0006u;                           // 1, 2
    if (y == 0 && z == 0) own |= 0x0001u;                 // 0
    if (x == 0 && z == 0) own |= 0x0008u;                 // 3
    if (x == 0 && y == 0) own |= 0x0100u;                 // 8
    if (halo_low && z == 0) own &= ~0x000fu;              // slab mode: plane 0's x/y edges belong to the previous slab
    return own;
}

struct CellRec;

// What the emit passes need to name the vertex on a lattice edge WITHOUT a lattice-wide edge -> id map (round 6).  The vertex
// belongs to the edge's owner -- the first, in sweep order, of the up to four cells around it: the one at the lowest
// coordinates, i.e. at offset {0, -1} from the edge's start point along the two axes across the edge -- and its id is
//     (vertex base of the owner's block) + (owned vertices of the earlier cells of the block: the owner's record) + (rank of the
//      edge among the owner's own vertices, in first-appearance order of its triangle list).
// The owner's record is the k-th of its block's, k = number of active cells before it in the row segment: one 16-byte entry of
// the block's chunk table (RowChunk) and a popcount.  mc_emit_brick_kernel does all this on LDS copies of the rows around a
// brick; the function below is the same from global memory (owners in the previous row SEGMENT, the slab top plane).  The rank comes from two bits of the record for an interior owner on the Lewiner tables (it owns
// only the vertices on edges 5, 6, 10 and the centre: LUT_ROWRANK), from a walk over its triangle list otherwise (cells on the
// low faces of the volume, classic tables).
// one 64-cell chunk of a block (a row segment of 256 cells): which of its cells are active, and where the first one's record is
struct RowChunk {
    unsigned long long mask;
    unsigned rec_index;   // index into the record pool of the chunk's first active cell (= the block's first one + the active cells
    unsigned pad;         // of the earlier chunks): record of cell xl = recs[rec_index + popcount(mask below bit xl % 64)]
};
struct EdgeIds {
    const CellRec *recs;
    const RowChunk *chunks;                 // [nblocks][4]
    const unsigned *vert_ofs;               // [nblocks] exclusive scan inside a group of 1024 blocks
    const unsigned long long *group_base;   // [ngroups] (tri | vert << 32)
    int bpr;                                // blocks per row of cells
};
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
    unsigned rec_capacity;  // records the pool of this workspace holds (set by the host with the header)
    unsigned pool_overflow; // a sub-pool ran out: no record was written past it, and the emit passes write nothing
    unsigned pad[6];
};
static_assert(sizeof(McHeader) == 64, "header is 64 bytes");

// One record per ACTIVE cell.  The active cells of a row segment of 256 cells ("block" b: the row-per-workgroup unit of the scans)
// are consecutive, in cell order, in the pool; the brick that classified the row took the space with one atomic add.  The pool is
// split into up to MC_SUBPOOLS equal parts with a counter each (a brick takes from the part its hashed index names): 13 000 bricks
// adding to ONE counter serialise in the L2 (classify pass 71 -> 116 us with two such counters), 200 per counter on 64 different
// lines do not.  A pool that holds every cell of the grid is never split (it cannot overflow, whatever the spread).
//   w0 = x_local | len << 8 | classic << 15 | ofs << 16      (the chosen tiling: no re-classification later)
//   w1 = exclusive prefix inside the block of ntri (bits 0-11) and of the owned vertices (bits 16-27), and -- for a cell with
//        x, y, z > 0 on the Lewiner tables -- the ranks of the vertices on edges 5 / 6 / 10 among the cell's own (bits 12-13 /
//        14-15 / 28-29; see LUT_ROWRANK); bit 30 = the classic flag again: w1 alone names an interior owner's vertices
struct CellRec { unsigned w0, w1; };
static constexpr int MC_SUBPOOLS = 64, MC_CTR_STRIDE = 16;   // one counter per 64-byte line
__device__ __forceinline__ unsigned rec_tri(const CellRec &r) { return r.w1 & 0xfffu; }
__device__ __forceinline__ unsigned rec_vert(const CellRec &r) { return (r.w1 >> 16) & 0xfffu; }
__device__ __forceinline__ unsigned rank_bits_to_w1(unsigned rk) { return ((rk & 15u) << 12) | (((rk >> 4) & 3u) << 28); }

// ---------------------------------------------------------------------------------------------
// Pass 1.  One workgroup classifies a BRICK of MC_TZ x MC_TY rows of cells (16 "blocks" of 256 cells of one row: the unit of
// the per-row counts, offsets and active masks the scans and the emit passes work on):
//   1. the (MC_TZ+1) x (MC_TY+1) lattice rows the brick touches are staged ONCE into LDS with coalesced 1-KiB row
//      loads (the row-per-workgroup form issued 8 loads per cell and fetched every row from 4 workgroups on up to
//      4 XCDs: 3x the volume in L2 misses); workgroups of one XCD take a contiguous range of bricks, so the rows two
//      neighbouring bricks share along y hit that XCD's L2;
//   2. sign patterns from LDS, one wave per row (no cross-wave exchange): ballot + popcount append the active
//      cells of the row to its list;
//   3. the expensive part -- Lewiner face / interior tests in fp64, triangle and owned-vertex counts -- runs over
//      the active cells of the WHOLE brick packed onto consecutive lanes (a row segment has ~13 active cells on a
//      typical surface: 1/20 of a workgroup; a brick ~200), corners read from LDS;
//   4. per-row exclusive prefix of the counts (one wave per row) and the record stores, into a piece of the record pool the
//      brick takes with one atomic add (the order of the pieces is irrelevant: the row's chunk table names its piece).
static constexpr int MC_TZ = 4, MC_TY = 4, MC_ROWS = MC_TZ * MC_TY;
static constexpr int MC_SRC_LD = 260;  // floats per staged lattice row (257 used)

__global__ __launch_bounds__(MC_BLOCK) void mc_classify_brick_kernel(const float *__restrict__ vol, Grid g, float levelf,
                                                                     double level, int classic, int nby, int bpr,
                                                                     CellRec *__restrict__ recs,
                                                                     RowChunk *__restrict__ chunks,
                                                                     unsigned *__restrict__ pool_ctr, int nsub, unsigned sub_cap,
                                                                     int *__restrict__ block_counts,
                                                                     int *__restrict__ block_nact,
                                                                     float2 *__restrict__ block_minmax,
                                                                     McHeader *__restrict__ hdr,
                                                                     const unsigned *__restrict__ signbits, int sign_ld) {
    constexpr int NSRC = (MC_TZ + 1) * (MC_TY + 1);
    constexpr int NCH = MC_BLOCK / 64;  // 64-value chunks of a row (= waves of the workgroup)
    // 38 KiB in all: four workgroups per CU
    __shared__ float s_src[NSRC * MC_SRC_LD];
    __shared__ unsigned long long s_mask[NSRC][NCH + 1];  // bit-planes "value > level" of the staged rows (+ the 257th value)
    __shared__ unsigned char s_list[MC_ROWS][MC_BLOCK];
    __shared__ unsigned char s_pk[MC_ROWS * MC_BLOCK];    // triangles | owned vertices << 4 of the packed active cells
    __shared__ int s_cnt[MC_ROWS];
    __shared__ unsigned long long s_act[MC_ROWS][MC_BLOCK / 64];   // active cells of the rows, 64 per word
    __shared__ unsigned s_pool;   // first record of this brick in the pool; ~0u: the pool is full
    __shared__ float s_mn[MC_BLOCK / 64], s_mx[MC_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) unsigned char s_lut[LUT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(LUT_BYTES % 16 == 0 && LUT_BYTES / 16 <= MC_BLOCK, "one 16-byte piece of the LUT blob per thread");
    if (tid < LUT_BYTES / 16) reinterpret_cast<uint4 *>(s_lut)[tid] = reinterpret_cast<const uint4 *>(d_lut_blob)[tid];
    // brick of this workgroup (XCD-contiguous order; bricks run seg fastest, then y, then z)
    const int brick = xcd_tile(blockIdx.x, gridDim.x);
    const int seg = brick % bpr, by = (brick / bpr) % nby, bz = brick / (bpr * nby);
    const int z0 = bz * MC_TZ, y0 = by * MC_TY, x0 = seg * MC_BLOCK;
    const int nz = min(MC_TZ, g.c0 - z0), ny = min(MC_TY, g.c1 - y0), nx = min(MC_BLOCK, g.c2 - x0);  // cells
    // ---- 1. stage the lattice rows: row (dz, dy), dz <= nz, dy <= ny, values x0 .. x0 + nx; the compare "value > level"
    // of a wave's 64 values IS its ballot, so the sign pass below works on 64-bit planes instead of on floats
    float mn = FLT_MAX, mx = -FLT_MAX;
    bool nan = false;
    // SIGNED form (sculpt_mc_count_launch_signed): the caller already holds the planes "value > level" of the whole lattice (the
    // filtered density grid's sign words, 32 values of the fastest axis per word): the masks come from 25 x 9 words instead of
    // 25 x 257 floats, and a brick without an active cell -- most of the volume -- never touches the volume at all; bricks with
    // active cells stage their rows as before (values for the Lewiner tests).  No data range is collected in this form.
    const bool use_planes = signbits != nullptr;   // kernel-uniform
    auto stage_rows = [&](bool masks) {
        float val[NSRC], edge = 0.f;
        const long sy = g.n2, sz = (long)g.n1 * g.n2;
        const float *base = vol + z0 * sz + y0 * sy + x0;
#pragma unroll
        for (int r = 0; r < NSRC; ++r) {
            const int dz = r / (MC_TY + 1), dy = r % (MC_TY + 1);
            const bool in = dz <= nz && dy <= ny;  // workgroup-uniform
            // lanes beyond the row read its last value again (finite, in range): min / max / NaN need no extra mask
            val[r] = in ? base[dz * sz + dy * sy + min(tid, nx)] : 0.f;
        }
        if (tid < NSRC) {  // the 257th value of every row (only present when the segment is full)
            const int dz = tid / (MC_TY + 1), dy = tid % (MC_TY + 1);
            if (dz <= nz && dy <= ny) edge = base[dz * sz + dy * sy + nx];
        }
#pragma unroll
        for (int r = 0; r < NSRC; ++r) {
            const int dz = r / (MC_TY + 1), dy = r % (MC_TY + 1);
            if (dz <= nz && dy <= ny) {
                mn = fminf(mn, val[r]);
                mx = fmaxf(mx, val[r]);
                nan |= val[r] != val[r];
            }
            s_src[r * MC_SRC_LD + tid] = val[r];
            if (masks) {
                const unsigned long long bal = __ballot(val[r] > levelf);  // == ((double)f - level > 0): levelf = largest float <= level
                if (lane == 0) s_mask[r][wave] = bal;
            }
        }
        if (tid < NSRC) {
            const int dz = tid / (MC_TY + 1), dy = tid % (MC_TY + 1);
            if (dz <= nz && dy <= ny) {
                mn = fminf(mn, edge);
                mx = fmaxf(mx, edge);
                nan |= edge != edge;
            }
            s_src[tid * MC_SRC_LD + nx] = edge;  // nx == 256: the extra column; nx < 256: rewrites the same value
            if (masks) s_mask[tid][NCH] = (edge > levelf) ? 1ull : 0ull;
        }
    };
    if (!use_planes) {
        stage_rows(true);
    } else if (tid < NSRC * (NCH + 1)) {
        const int r = tid / (NCH + 1), c = tid % (NCH + 1);
        const int dz = r / (MC_TY + 1), dy = r % (MC_TY + 1);
        unsigned long long m = 0ull;
        if (dz <= nz && dy <= ny) {
            const unsigned *w = signbits + ((long)(z0 + dz) * g.n1 + (y0 + dy)) * sign_ld;
            const int w0 = (x0 >> 5) + 2 * c;   // x0 is a multiple of MC_BLOCK = 256: word aligned; bits past the row are 0
            const unsigned lo = w0 < sign_ld ? w[w0] : 0u, hi = w0 + 1 < sign_ld ? w[w0 + 1] : 0u;
            m = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
        s_mask[r][c] = c < NCH ? m : (m & 1ull);   // chunk NCH: the 257th value of a full segment
    }
    if (nan) hdr->nan_seen = 1u;  // fminf/fmaxf drop NaN silently; a plain racing store of 1 is enough
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
    __syncthreads();
    // ---- 2. active cells, one wave per row of cells: a cell is active unless its 8 corner bits are equal.  Corner bits of
    // the 64 cells of a chunk = the planes of the 4 lattice rows at x (m) and at x + 1 (m shifted, bit 63 from the next chunk)
    for (int lr = wave; lr < MC_ROWS; lr += MC_BLOCK / 64) {
        const int dz = lr / MC_TY, dy = lr % MC_TY;
        int n = 0;
        if (dz < nz && dy < ny) {
            const int r00 = dz * (MC_TY + 1) + dy;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                unsigned long long any = 0ull, all = ~0ull;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = r00 + (q & 1) + (q >> 1) * (MC_TY + 1);
                    const unsigned long long m = s_mask[r][c], m1 = (m >> 1) | (s_mask[r][c + 1] << 63);
                    any |= m | m1;
                    all &= m & m1;
                }
                const int left = nx - c * 64;  // cells of this chunk inside the grid
                const unsigned long long valid = left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
                const unsigned long long act = any & ~all & valid;
                if (lane == 0) s_act[lr][c] = act;   // -> the row's chunk table (step 4)
                if ((act >> lane) & 1ull) s_list[lr][n + __popcll(act & ((1ull << lane) - 1ull))] = (unsigned char)(c * 64 + lane);
                n += __popcll(act);
            }
        }
        if (lane == 0) s_cnt[lr] = n;
    }
    __syncthreads();
    // ---- 3. dense classification over the packed active cells of the brick
    int pre[MC_ROWS + 1];
    pre[0] = 0;
#pragma unroll
    for (int r = 0; r < MC_ROWS; ++r) pre[r + 1] = pre[r] + s_cnt[r];
    const int total_active = pre[MC_ROWS];
    if (total_active > 0) {   // workgroup-uniform
        if (tid == 0) {       // this brick's piece of the record pool
            const unsigned sub = (((unsigned)brick * 2654435761u) >> 12) % (unsigned)nsub;
            unsigned at = atomicAdd(&pool_ctr[sub * MC_CTR_STRIDE], (unsigned)total_active);
            if (at + (unsigned)total_active > sub_cap) { hdr->pool_overflow = 1u; at = ~0u; } else at += sub * sub_cap;
            s_pool = at;
        }
        if (use_planes) {     // this brick needs its values after all
            stage_rows(false);
            if (nan) hdr->nan_seen = 1u;
        }
        __syncthreads();
    }
    const TabLds T{s_lut};
    for (int p = tid; p < total_active; p += MC_BLOCK) {
        int lr = 0;
#pragma unroll
        for (int r = 1; r < MC_ROWS; ++r) lr += (p >= pre[r]) ? 1 : 0;
        int start = 0;
#pragma unroll
        for (int r = 0; r < MC_ROWS; ++r) start = (r == lr) ? pre[r] : start;
        const int dz = lr / MC_TY, dy = lr % MC_TY;
        const int xl = s_list[lr][p - start];
        const float *r00 = s_src + (dz * (MC_TY + 1) + dy) * MC_SRC_LD + xl, *r01 = r00 + MC_SRC_LD;
        const float *r10 = r00 + (MC_TY + 1) * MC_SRC_LD, *r11 = r10 + MC_SRC_LD;
        // Lewiner corner order (load_cell): 0 (z,y,x) 1 (z,y,x+1) 2 (z,y+1,x+1) 3 (z,y+1,x) 4..7 the same at z+1
        double v[8];
        v[0] = (double)r00[0] - level; v[1] = (double)r00[1] - level; v[2] = (double)r01[1] - level; v[3] = (double)r01[0] - level;
        v[4] = (double)r10[0] - level; v[5] = (double)r10[1] - level; v[6] = (double)r11[1] - level; v[7] = (double)r11[0] - level;
        int row = -1;
        const Tiling t = classify(T, v, classic != 0, &row);
        const int cx = x0 + xl, cy = y0 + dy, cz = z0 + dz;
        int counts;
        unsigned rk = 0;
        if (row < 0) {
            counts = cell_counts(t, cx, cy, cz, g.halo_low);  // classic tables (or inactive): the entry walk
        } else {
            // owned vertices = edges of the row's triangles (bit mask) that no earlier cell of the sweep touches (owns_edge)
            unsigned own = 0x1460u;                               // edges 5, 6, 10 and the centre vertex 12: always
            if (cx == 0) own |= 0x0880u;                          // 7, 11
            if (cy == 0) own |= 0x0210u;                          // 4, 9
            if (cz == 0) own |= 0x0006u;                          // 1, 2
            if (cy == 0 && cz == 0) own |= 0x0001u;               // 0
            if (cx == 0 && cz == 0) own |= 0x0008u;               // 3
            if (cx == 0 && cy == 0) own |= 0x0100u;               // 8
            if (g.halo_low && cz == 0) own &= ~0x000fu;           // slab mode: plane 0's x/y edges belong to the previous slab
            const int grow = T.rowbase(row >> 10) + (row & 1023);
            const unsigned used = T.rowmask(grow);
            counts = (t.len / 3) | (__popc(used & own) << 16);
            rk = T.rowrank(grow);   // meaningful for interior cells only; edge_vertex_id walks the entries for the others
        }
        s_pk[p] = (unsigned char)((counts & 15) | ((counts >> 16) << 4));  // <= 12 triangles, <= 13 owned vertices
        s_list[lr][p - start] = (unsigned char)rk;   // (this thread's own slot: x_local has been read) -> step 4
        const unsigned ofs = t.ofs < 0 ? (unsigned)(-t.ofs - 1) : (unsigned)t.ofs;
        if (s_pool != ~0u)
            recs[s_pool + p].w0 = (unsigned)xl | ((unsigned)t.len << 8) | ((t.ofs < 0 ? 1u : 0u) << 15) | (ofs << 16);
    }
    __syncthreads();
    // ---- 4. per-row exclusive prefix (one wave per row) and the records of the row's virtual block
    float bmn = FLT_MAX, bmx = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < MC_BLOCK / 64; ++w) { bmn = fminf(bmn, s_mn[w]); bmx = fmaxf(bmx, s_mx[w]); }
    for (int lr = wave; lr < MC_ROWS; lr += MC_BLOCK / 64) {
        const int dz = lr / MC_TY, dy = lr % MC_TY;
        if (dz >= nz || dy >= ny) continue;
        const long blk = ((long)(z0 + dz) * g.c1 + (y0 + dy)) * bpr + seg;
        int start = 0;
#pragma unroll
        for (int r = 0; r < MC_ROWS; ++r) start = (r == lr) ? pre[r] : start;
        const int n = s_cnt[lr];
        int carry = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const int b = k < n ? s_pk[start + k] : 0;
            const int val = (b & 15) | ((b >> 4) << 16);
            int inc = val;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(inc, d, 64);
                if (lane >= d) inc += up;
            }
            if (k < n && s_pool != ~0u)
                recs[s_pool + start + k].w1 = (unsigned)(carry + inc - val) | rank_bits_to_w1(s_list[lr][k]) | (classic ? 1u << 30 : 0u);
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) {
            block_counts[blk] = carry;
            block_nact[blk] = n;
        }
        if (lane < NCH) {   // the row's chunk table
            unsigned before = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) before += (c < lane && n > 0) ? (unsigned)__popcll(s_act[lr][c]) : 0u;
            RowChunk ch;
            ch.mask = n > 0 ? s_act[lr][lane] : 0ull;
            ch.rec_index = total_active > 0 ? s_pool + (unsigned)start + before : 0u;
            ch.pad = 0;
            chunks[blk * NCH + lane] = ch;
            // the brick's min / max on its first row, neutral elements on the others (the scans only reduce them)
            block_minmax[blk] = (lr == 0 && !use_planes) ? make_float2(bmn, bmx) : make_float2(FLT_MAX, -FLT_MAX);
        }
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

// rank of the vertex on edge eo among the vertices cell (ox, oy, oz) owns (record rec), in first-appearance order of its triangles
__device__ __forceinline__ bool rank_in_w1(unsigned w1, int ox, int oy, int oz) { return ox > 0 && oy > 0 && oz > 0 && !(w1 >> 30 & 1u); }
__device__ __forceinline__ int rank_from_w1(unsigned w1, int eo) {   // an interior owner on the Lewiner tables: eo is 5, 6 or 10
    return (int)((eo == 5 ? w1 >> 12 : eo == 6 ? w1 >> 14 : w1 >> 28) & 3u);
}
__device__ __forceinline__ int owned_rank(const CellRec &rec, int eo, int ox, int oy, int oz, int halo_low) {
    if (rank_in_w1(rec.w1, ox, oy, oz)) return rank_from_w1(rec.w1, eo);
    const Tiling t = tiling_of(rec.w0);
    unsigned seen = 0;
    int rank = 0;
    for (int i = 0; i < t.len; ++i) {
        const int e = tiling_entry(t, i);
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (e == eo) break;
        rank += owns_edge(e, ox, oy, oz, halo_low) ? 1 : 0;
    }
    return rank;
}

// owner cell (ox, oy, oz) of the lattice edge (axis, lx, ly, lz) and the edge's number eo inside the owner
__device__ __forceinline__ int owner_of(int axis, int lx, int ly, int lz, int &ox, int &oy, int &oz) {
    ox = lx; oy = ly; oz = lz;
    if (axis == 0) {
        oy = max(ly - 1, 0); oz = max(lz - 1, 0);
        const int dy = ly - oy, dz = lz - oz;
        return dz ? (dy ? 6 : 4) : (dy ? 2 : 0);
    }
    if (axis == 1) {
        ox = max(lx - 1, 0); oz = max(lz - 1, 0);
        const int dx = lx - ox, dz = lz - oz;
        return dz ? (dx ? 5 : 7) : (dx ? 1 : 3);
    }
    ox = max(lx - 1, 0); oy = max(ly - 1, 0);
    const int dx = lx - ox, dy = ly - oy;
    return dy ? (dx ? 10 : 11) : (dx ? 9 : 8);
}

// id of the vertex on the lattice edge (axis, lx, ly, lz); the edge must carry one (its end points differ in sign), and in slab
// mode with a lower halo it must not lie in lattice plane 0 along x / y (those vertices belong to the previous slab).
__device__ int lattice_edge_vertex_id(const EdgeIds &E, const Grid &g, int axis, int lx, int ly, int lz) {
    int ox, oy, oz;
    const int eo = owner_of(axis, lx, ly, lz, ox, oy, oz);
    const int ob = (oz * g.c1 + oy) * E.bpr + (ox >> 8), xl = ox & 255;
    const RowChunk ch = E.chunks[(long)ob * 4 + (xl >> 6)];
    const CellRec rec = E.recs[ch.rec_index + (unsigned)__popcll(ch.mask & ((1ull << (xl & 63)) - 1ull))];
    const unsigned base = (unsigned)(E.group_base[ob >> 10] >> 32) + E.vert_ofs[ob] + rec_vert(rec);
    return (int)(base + (unsigned)owned_rank(rec, eo, ox, oy, oz, g.halo_low));
}
__device__ __forceinline__ int edge_vertex_id(const EdgeIds &E, const Grid &g, int e, int x, int y, int z) {
    int axis, lx, ly, lz;
    lattice_edge_of(e, x, y, z, axis, lx, ly, lz);
    return lattice_edge_vertex_id(E, g, axis, lx, ly, lz);
}

// The emit pass: vertices and triangles of the active cells, one workgroup per EMIT BRICK of MC_EZ x MC_EY = 8 x 6 rows x 64 cells
// (grid-stride; a brick without an active cell costs one look at its rows' chunk-table entries).
//   * A cell writes the vertices it OWNS (owns_edge) at (vertex base of its block) + (owned vertices of the earlier cells of
//     the block: its record) + rank, and its triangles at the same kind of offset.
//   * The id of a vertex on an edge of cell (x, y, z) is the owner's base + the rank of the edge among the owner's vertices
//     (lattice_edge_vertex_id); the owner is a cell at offset {0, -1} along each axis, i.e. in one of the (MC_EZ + 1) x
//     (MC_EY + 1) rows around the brick.  The workgroup copies the second word of those rows' records -- all an interior
//     owner's ids need -- into LDS once (coalesced; 63 rows), and a look-up is two LDS reads: no lattice-wide
//     edge -> id map (round 5: int32 [R^3][4], written and read with one divergent global access per vertex reference).
//     Owners outside the copy (x - 1 of the brick's first column) or on a low face of the volume go through global memory.
//   * 64 cells along x, not a whole 256-cell row segment: a surface lying flat along x fills a brick, and the kernel ends
//     with its slowest workgroup (whole segments: 8000-cell bricks, the average wave resident for 1/7 of the kernel's time).
//   * The triangle tables sit in LDS (two entries per byte: 6 KiB, loaded once per workgroup): a cell walks 10-40 entries of
//     its tiling, each a divergent read.
//   * The kernel is bound by the bricks in flight (every step of a brick waits on the one before: rows -> records -> cells),
//     i.e. by its LDS and registers: 25 KiB and 80 registers a workgroup / lane, six workgroups per CU (38 KiB, 117
//     registers, four per CU: 136 us instead of 124 with 4 x 8-row bricks; 128 threads instead of 256 on the same LDS: 190 us;
//     1536 / 2048 record words instead of 1024: 113 us instead of 105).
// Volume reads: the 8 corners of every cell that owns a vertex, requested together.
// rows (z, y) and cells along x of an emit brick.  (EZ, EY) with (EZ + 1)(EY + 1) <= 64, bench volume: (8, 6) 104-106 us, (15, 3)
// 108, (7, 7) 112, (9, 5) 114, (6, 8) 117, (4, 8) 125, (12, 3) 125, (4, 11) 129, (8, 4) 137, (2, 8) 152, (4, 4) 169
static constexpr int MC_EZ = 8, MC_EY = 6, MC_EX = 64;
static constexpr int MC_EROWS = MC_EZ * MC_EY, MC_EHROWS = (MC_EZ + 1) * (MC_EY + 1);
static constexpr int MC_EREC = 1024;                          // records (their w1) in LDS; a brick of a closed surface has 100-400 with
                                                              // the rows around it (more than this: look-ups through global memory)
static_assert(MC_EHROWS <= 64, "the rows around an emit brick are handled by the lanes of one wave");
static_assert(MC_EX == 64 && MC_BLOCK % MC_EX == 0, "an emit brick spans one 64-cell chunk of the rows' chunk tables");

template <typename IdxT>
__global__ __launch_bounds__(MC_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void mc_emit_brick_kernel(const float *__restrict__ vol, Grid g, double level,
                                                                 const CellRec *__restrict__ recs, EdgeIds E,
                                                                 const McHeader *__restrict__ hdr,
                                                                 const unsigned *__restrict__ tri_ofs, int nby, int nbx, int nbricks,
                                                                 float *__restrict__ verts, float vdiv, float vmul, float vadd,
                                                                 int affine, IdxT *__restrict__ faces, int ref_order,
                                                                 unsigned long long cap_vert, unsigned long long cap_tri) {
    // speculative emit (sculpt_mc_emit_capped): the buffers were sized before the counts were read back; a mesh that does not fit
    // writes nothing at all (the caller sees the counts, allocates and emits again) ... and so does a count pass whose record pool
    // was too small (the caller repeats it with a larger workspace)
    if (hdr->total_vert > cap_vert || hdr->total_tri > cap_tri || hdr->pool_overflow) return;
    __shared__ unsigned s_w1[MC_EREC];                         // w1 of the records of the rows around the brick, row after row
    __shared__ RowChunk s_chunk[MC_EHROWS];                    // the rows' chunk-table entries for this brick's 64 columns
    __shared__ int s_start[MC_EHROWS + 1];                     // first record of row r in s_w1
    __shared__ unsigned s_vabs[MC_EHROWS], s_tabs[MC_EHROWS];  // absolute vertex / triangle base of the row's block
    __shared__ int s_own[MC_EROWS + 1];                        // prefix of the active cells of the brick's own rows
    __shared__ __attribute__((aligned(16))) unsigned char s_nib[MC_TNIB16 * 16];
    __shared__ int s_ids[13 * MC_BLOCK];                       // per thread: the vertex ids on the 12 edges + centre of its cell
    const int tid = threadIdx.x;
    for (int i = tid; i < MC_TNIB16; i += blockDim.x) reinterpret_cast<uint4 *>(s_nib)[i] = reinterpret_cast<const uint4 *>(d_tiling_nib)[i];
    auto entry = [&](const Tiling &t, int i) -> int {
        if (t.ofs < 0) return d_cases_classic[(-t.ofs - 1) + i];
        const int q = t.ofs + i;
        return (s_nib[q >> 1] >> ((q & 1) * 4)) & 15;
    };
    const long sy = g.n2, sz = (long)g.n1 * g.n2;
    for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
        const int bx = brick % nbx, by = (brick / nbx) % nby, bz = brick / (nbx * nby);
        const int z0 = bz * MC_EZ, y0 = by * MC_EY, x0 = bx * MC_EX;
        const int seg = x0 / MC_BLOCK, chunk = (x0 % MC_BLOCK) / MC_EX;
        __syncthreads();   // the previous brick's look-ups are done (and the tables are there)
        // ---- the rows around the brick: chunk-table entries, bases; prefix sums inside the first wave
        if (tid < 64) {
            const int dz = tid / (MC_EY + 1) - 1, dy = tid % (MC_EY + 1) - 1;
            const int z = z0 + dz, y = y0 + dy;
            int n = 0;
            if (tid < MC_EHROWS && z >= 0 && z < g.c0 && y >= 0 && y < g.c1) {
                const int blk = (z * g.c1 + y) * E.bpr + seg;
                const RowChunk ch = E.chunks[(long)blk * (MC_BLOCK / 64) + chunk];
                n = __popcll(ch.mask);
                if (n > 0) {
                    const unsigned long long gb = E.group_base[blk >> 10];
                    s_vabs[tid] = (unsigned)(gb >> 32) + E.vert_ofs[blk];
                    s_tabs[tid] = (unsigned)(gb & 0xffffffffull) + tri_ofs[blk];
                    s_chunk[tid] = ch;
                }
            }
            const bool own = tid < MC_EHROWS && dz >= 0 && dy >= 0;
            int inc = n, oinc = own ? n : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(inc, d, 64), oup = __shfl_up(oinc, d, 64);
                if (tid >= d) { inc += up; oinc += oup; }
            }
            if (tid < MC_EHROWS) s_start[tid + 1] = inc;
            if (own) s_own[dz * MC_EY + dy + 1] = oinc;
            if (tid == 0) { s_start[0] = 0; s_own[0] = 0; }
        }
        __syncthreads();
        const int n_own = s_own[MC_EROWS];
        if (n_own == 0) continue;   // workgroup-uniform: no surface in this brick
        // ---- w1 of the records of the rows -> LDS (when they fit)
        const int total = s_start[MC_EHROWS];
        const bool in_lds = total <= MC_EREC;
        for (int i = tid; i < (in_lds ? total : 0); i += blockDim.x) {
            int lo = 0, hi = MC_EHROWS;   // row of record i: the last r with s_start[r] <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_start[mid] <= i) lo = mid; else hi = mid;
            }
            s_w1[i] = recs[s_chunk[lo].rec_index + (unsigned)(i - s_start[lo])].w1;   // (row lo has records: its entry is there)
        }
        __syncthreads();
        // ---- the brick's own active cells
        for (int p = tid; p < n_own; p += blockDim.x) {
            int lo = 0, hi = MC_EROWS;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_own[mid] <= p) lo = mid; else hi = mid;
            }
            const int dz = lo / MC_EY, dy = lo % MC_EY, fr = (dz + 1) * (MC_EY + 1) + (dy + 1);
            const int kself = p - s_own[lo];
            const CellRec rec = recs[s_chunk[fr].rec_index + (unsigned)kself];   // (consecutive lanes: consecutive records)
            const int x = seg * MC_BLOCK + (int)(rec.w0 & 0xffu), y = y0 + dy, z = z0 + dz;
            const Tiling t = tiling_of(rec.w0);
            const unsigned tri0 = s_tabs[fr] + rec_tri(rec);
            const unsigned vown0 = s_vabs[fr] + rec_vert(rec);
            const float *cell = vol + z * sz + y * sy + x;
            // -- one walk over the triangle list: the edges it uses, and the owned ones in first-appearance order (4 bits each)
            const unsigned own = owned_mask(x, y, z, g.halo_low);
            unsigned used = 0;
            unsigned long long own_list = 0;
            int n_owned = 0;
            for (int i = 0; i < t.len; ++i) {
                const int e = entry(t, i);
                if (used >> e & 1u) continue;
                used |= 1u << e;
                if (own >> e & 1u) { own_list |= (unsigned long long)e << (4 * n_owned); ++n_owned; }
            }
            int *ids = s_ids + tid;   // ids[e * MC_BLOCK]: the vertex id on edge e of this cell (12: the centre vertex)
            // -- the vertices this cell owns (every lane runs the same code: the walk above took the divergence).  The 8 corner
            // values are requested together, before the first is needed: one exposed load latency per cell, not one per vertex.
            if (n_owned > 0) {
                double v[8];
                v[0] = (double)cell[0] - level; v[1] = (double)cell[1] - level;
                v[2] = (double)cell[sy + 1] - level; v[3] = (double)cell[sy] - level;
                v[4] = (double)cell[sz] - level; v[5] = (double)cell[sz + 1] - level;
                v[6] = (double)cell[sz + sy + 1] - level; v[7] = (double)cell[sz + sy] - level;
                // corner (dx, dy, dz) of the cell (static selects: a run-time index would put v[] in scratch memory)
                auto corner = [&](int dx, int dy, int dz) -> double {
                    const double a = dx ? v[1] : v[0], b = dx ? v[2] : v[3], c = dx ? v[5] : v[4], d = dx ? v[6] : v[7];
                    const double lo = dy ? b : a, hi = dy ? d : c;
                    return dz ? hi : lo;
                };
                for (int j = 0; j < n_owned; ++j) {
                    const int e = (int)((own_list >> (4 * j)) & 15ull);
                    const unsigned id = vown0 + (unsigned)j;
                    ids[e * MC_BLOCK] = (int)id;
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
                        // near = the edge's start point (lower lattice coordinate along its axis), far = the next point along it
                        int axis, lx, ly, lz;
                        lattice_edge_of(e, x, y, z, axis, lx, ly, lz);
                        const int dx = lx - x, dy = ly - y, dz = lz - z;
                        const double v_near = corner(dx, dy, dz);
                        const double v_far = corner(dx | (axis == 0), dy | (axis == 1), dz | (axis == 2));
                        const double frc = edge_frac(v_near, v_far);
                        // cell origin + (1.0 or frc): the off-axis offsets are exactly 0.0 or 1.0
                        px = (axis == 0) ? (double)x + frc : (double)lx;
                        py = (axis == 1) ? (double)y + frc : (double)ly;
                        pz = (axis == 2) ? (double)(z + g.z_off) + frc : (double)(lz + g.z_off);
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
                }
            }
            // -- the vertices on its other edges: one look-up per edge, whatever the number of triangles that share it
            for (unsigned rest = used & ~own; rest; rest &= rest - 1) {
                const int e = __ffs(rest) - 1;
                int id;
                if (g.halo_low && z == 0 && e < 4) {
                    // vertex lives in the previous slab: encode the edge of lattice plane 0 as -(1 + slot),
                    // slot = axis*(n1*n2) + ly*n2 + lx; resolved after the gather (sculptmate_amd/slab.py)
                    const int axis = e & 1, lx = x + (e == 1), ly = y + (e == 2);
                    id = -(1 + axis * g.n1 * g.n2 + ly * g.n2 + lx);
                } else {
                    int axis, lx, ly, lz, ox, oy, oz;
                    lattice_edge_of(e, x, y, z, axis, lx, ly, lz);
                    const int eo = owner_of(axis, lx, ly, lz, ox, oy, oz);
                    const int oxl = ox - x0;
                    if (oxl >= 0 && in_lds) {
                        const int orow = (oz - z0 + 1) * (MC_EY + 1) + (oy - y0 + 1);
                        const unsigned ow1 = s_w1[s_start[orow] + __popcll(s_chunk[orow].mask & ((1ull << oxl) - 1ull))];
                        if (rank_in_w1(ow1, ox, oy, oz)) id = (int)(s_vabs[orow] + ((ow1 >> 16) & 0xfffu)) + rank_from_w1(ow1, eo);
                        else id = lattice_edge_vertex_id(E, g, axis, lx, ly, lz);   // an owner on a low face / classic tables: the walk
                    } else {
                        id = lattice_edge_vertex_id(E, g, axis, lx, ly, lz);       // the owner sits left of the brick (or: too many records)
                    }
                }
                ids[e * MC_BLOCK] = id;
            }
            // -- its triangles
            const int ntri = t.len / 3;
            for (int k = 0; k < ntri; ++k) {
                const int a = ids[entry(t, 3 * k) * MC_BLOCK], b2 = ids[entry(t, 3 * k + 1) * MC_BLOCK], c = ids[entry(t, 3 * k + 2) * MC_BLOCK];
                // internal (a,b,c); skimage 'descent' flips to (c,b,a); the reference then takes [1,0,2] -> (b,c,a)
                IdxT *f = faces + 3 * (size_t)(tri0 + k);
                if (ref_order) { f[0] = (IdxT)b2; f[1] = (IdxT)c; f[2] = (IdxT)a; }
                else { f[0] = (IdxT)c; f[1] = (IdxT)b2; f[2] = (IdxT)a; }
            }
        }
    }
}

// lattice-edge -> vertex-id map of the LAST lattice plane (x and y edges), -1 where the edge has no crossing
__global__ __launch_bounds__(256) void mc_top_plane_kernel(const float *__restrict__ vol, Grid g, double level,
                                                           EdgeIds E, const McHeader *__restrict__ hdr, int *__restrict__ out) {
    if (hdr->pool_overflow) return;
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
        if (a != b) v = lattice_edge_vertex_id(E, g, axis, lx, ly, lz);
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
    size_t off_counts, off_nact, off_minmax, off_tri, off_vert, off_gtot, off_gmm, off_recs, off_chunks, off_ctr, off_aofs, off_gact, total;
    int ngroups;
    int nblocks;
    long rec_capacity, nbricks;
    int nsub;                 // parts of the record pool (one counter each)
};
// records of the default pool: one active cell per 8 cells (a closed surface at 256^3 has ~1 per 17), at least 65 536
static long default_rec_capacity(const Grid &g) { return std::min(g.ncells, std::max(g.ncells / 8, 65536L)); }
// (the record pool comes LAST: every other offset is independent of its capacity)
static WsLayout ws_layout(const Grid &g, long rec_capacity) {
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
    w.off_tri = o;    o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_vert = o;   o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_chunks = o; o = al(o + sizeof(RowChunk) * (MC_BLOCK / 64) * (size_t)w.nblocks);
    w.nbricks = (long)cdiv(g.c2, MC_BLOCK) * cdiv(g.c1, MC_TY) * cdiv(g.c0, MC_TZ);
    w.off_aofs = o;   o = al(o + sizeof(unsigned) * w.nblocks);
    w.off_gact = o;   o = al(o + sizeof(unsigned) * w.ngroups);
    w.rec_capacity = std::max(1L, std::min(rec_capacity, g.ncells));
    w.nsub = w.rec_capacity >= g.ncells ? 1 : (int)std::max(1L, std::min<long>(MC_SUBPOOLS, w.nbricks / 16));
    w.off_ctr = o;    o = al(o + sizeof(unsigned) * MC_SUBPOOLS * MC_CTR_STRIDE);
    w.off_recs = o;   o = al(o + sizeof(CellRec) * (size_t)w.rec_capacity);
    w.total = o;
    return w;
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

size_t sculpt_mc_workspace_bytes(int n0, int n1, int n2) {
    Grid g;
    if (make_grid(n0, n1, n2, &g)) return 0;
    return ws_layout(g, default_rec_capacity(g)).total;
}

size_t sculpt_mc_workspace_bytes_for(int n0, int n1, int n2, int64_t max_active_cells) {
    Grid g;
    if (make_grid(n0, n1, n2, &g)) return 0;
    return ws_layout(g, max_active_cells <= 0 ? default_rec_capacity(g) : (long)max_active_cells).total;
}

static int mc_count_launch(const float *vol, const unsigned *signbits, int sign_ld, int n0, int n1, int n2, double level,
                           unsigned flags, long max_active_cells, void *workspace, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    Grid g;
    if (int rc = make_grid(n0, n1, n2, &g)) return rc;
    g.halo_low = (flags & SCULPT_MC_SLAB_HALO_LOW) ? 1 : 0;
    SC_REQUIRE(vol && workspace, "mc_count: null argument");
    if (int rc = upload_tables()) return rc;
    const WsLayout w = ws_layout(g, max_active_cells <= 0 ? default_rec_capacity(g) : max_active_cells);
    char *ws = reinterpret_cast<char *>(workspace);
    McHeader *hdr = reinterpret_cast<McHeader *>(ws);
    McHeader init;
    memset(&init, 0, sizeof(init));
    init.min_ord = 0xffffffffu;
    init.max_ord = 0u;
    init.rec_capacity = (unsigned)std::min<long>(w.rec_capacity, 0xffffffffL);
    SC_HIP(hipMemsetAsync(ws + w.off_ctr, 0, sizeof(unsigned) * MC_SUBPOOLS * MC_CTR_STRIDE, st));
    SC_HIP(hipMemcpyAsync(hdr, &init, sizeof(init), hipMemcpyHostToDevice, st));
    const int classic = (flags & SCULPT_MC_USE_CLASSIC) ? 1 : 0;
    // float f > double level  <=>  f > (largest float <= level): the sign pass needs no fp64
    float levelf = (float)level;
    if ((double)levelf > level) levelf = nextafterf(levelf, -INFINITY);
    {
        const int bpr = cdiv(g.c2, MC_BLOCK), nby = cdiv(g.c1, MC_TY), nbz = cdiv(g.c0, MC_TZ);
        hipLaunchKernelGGL(mc_classify_brick_kernel, dim3(bpr * nby * nbz), dim3(MC_BLOCK), 0, st, vol, g, levelf, level, classic,
                           nby, bpr, reinterpret_cast<CellRec *>(ws + w.off_recs), reinterpret_cast<RowChunk *>(ws + w.off_chunks),
                           reinterpret_cast<unsigned *>(ws + w.off_ctr), w.nsub, (unsigned)(w.rec_capacity / w.nsub),
                           reinterpret_cast<int *>(ws + w.off_counts),
                           reinterpret_cast<int *>(ws + w.off_nact), reinterpret_cast<float2 *>(ws + w.off_minmax), hdr, signbits, sign_ld);
    }
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
    return 0;
}

int sculpt_mc_count_launch(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                           sculpt_stream_t stream) {
    SC_REQUIRE(!(flags & SCULPT_MC_SIGNED), "mc_count: SCULPT_MC_SIGNED belongs to sculpt_mc_count_launch_signed");
    return mc_count_launch(vol, nullptr, 0, n0, n1, n2, level, flags, 0, workspace, stream);
}

int sculpt_mc_count_launch_for(const float *vol, const uint32_t *sign_planes, int words_per_row, int n0, int n1, int n2,
                               double level, unsigned flags, int64_t max_active_cells, void *workspace, sculpt_stream_t stream) {
    SC_REQUIRE(max_active_cells >= 0, "mc_count: negative record capacity");
    if (flags & SCULPT_MC_SIGNED) {
        SC_REQUIRE(sign_planes && words_per_row >= (n2 + 31) / 32, "mc_count_signed: sign planes missing or rows too short (%d words for n2=%d)",
                   words_per_row, n2);
        SC_REQUIRE(!(flags & (SCULPT_MC_SLAB | SCULPT_MC_SLAB_HALO_LOW)), "mc_count_signed: not in slab mode (no data range is collected)");
    } else {
        SC_REQUIRE(!sign_planes, "mc_count: sign planes need SCULPT_MC_SIGNED");
    }
    return mc_count_launch(vol, sign_planes, words_per_row, n0, n1, n2, level, flags, (long)max_active_cells, workspace, stream);
}

int sculpt_mc_count_launch_signed(const float *vol, const uint32_t *sign_planes, int words_per_row, int n0, int n1, int n2,
                                  double level, unsigned flags, void *workspace, sculpt_stream_t stream) {
    SC_REQUIRE(sign_planes && words_per_row >= (n2 + 31) / 32, "mc_count_signed: sign planes missing or rows too short (%d words for n2=%d)",
               words_per_row, n2);
    SC_REQUIRE(!(flags & (SCULPT_MC_SLAB | SCULPT_MC_SLAB_HALO_LOW)), "mc_count_signed: not in slab mode (no data range is collected)");
    return mc_count_launch(vol, sign_planes, words_per_row, n0, n1, n2, level, flags, 0, workspace, stream);
}

int sculpt_mc_count_read(int n0, int n1, int n2, double level, unsigned flags, const void *workspace, int64_t *n_verts_host,
                         int64_t *n_faces_host, float *minmax_host, sculpt_stream_t stream) {
    return sculpt_mc_count_read_ex(n0, n1, n2, level, flags, workspace, n_verts_host, n_faces_host, minmax_host, nullptr, stream);
}

int sculpt_mc_count_read_ex(int n0, int n1, int n2, double level, unsigned flags, const void *workspace, int64_t *n_verts_host,
                            int64_t *n_faces_host, float *minmax_host, int64_t *n_active_host, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    (void)n0; (void)n1; (void)n2;
    SC_REQUIRE(workspace && n_verts_host && n_faces_host, "mc_count: null argument");
    const McHeader *hdr = reinterpret_cast<const McHeader *>(workspace);
    McHeader res;
    SC_HIP(hipMemcpyAsync(&res, hdr, sizeof(res), hipMemcpyDeviceToHost, st));
    SC_HIP(hipStreamSynchronize(st));
    *n_verts_host = (int64_t)res.total_vert;
    *n_faces_host = (int64_t)res.total_tri;
    if (n_active_host) *n_active_host = (int64_t)res.total_active;
    // skimage: "Surface level must be within volume data range." (ValueError)
    const float mn = ord2f(res.min_ord), mx = ord2f(res.max_ord);
    if (minmax_host) { minmax_host[0] = mn; minmax_host[1] = mx; }
    if (res.nan_seen) {
        set_error("marching_cubes: the volume contains NaN");
        return SCULPT_ERR_MC_NAN;
    }
    // (before anything that returns early: an overflow means active cells, i.e. neither of skimage's two errors)
    if (res.pool_overflow) {   // the counts above are right (they do not depend on the records); the emit passes would write nothing
        set_error("marching_cubes: %u active cells do not fit the workspace's record pool (%u): repeat the count with "
                  "sculpt_mc_workspace_bytes_for / sculpt_mc_count_launch_for and a capacity of at least that many",
                  res.total_active, res.rec_capacity);
        return SCULPT_ERR_MC_WORKSPACE;
    }
    if (flags & SCULPT_MC_SLAB) return 0;  // a slab may be empty; the caller decides globally
    // (SIGNED: no data range was collected; an empty result is reported as EMPTY and the caller asks the unsigned form which of
    // skimage's two errors it is)
    if (!(flags & SCULPT_MC_SIGNED) && ((double)level < (double)mn || (double)level > (double)mx)) {
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

int sculpt_mc_count(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                    int64_t *n_verts_host, int64_t *n_faces_host, float *minmax_host, sculpt_stream_t stream) {
    SC_REQUIRE(n_verts_host && n_faces_host, "mc_count: null argument");
    if (int rc = sculpt_mc_count_launch(vol, n0, n1, n2, level, flags, workspace, stream)) return rc;
    return sculpt_mc_count_read(n0, n1, n2, level, flags, workspace, n_verts_host, n_faces_host, minmax_host, stream);
}

int sculpt_mc_emit_capped(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                          float vert_div, float vert_mul, float vert_add, int axis0_offset, float *verts, int64_t cap_verts,
                          void *faces, int64_t cap_faces, int *top_plane_map, sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    SC_REQUIRE(cap_verts >= 0 && cap_faces >= 0, "mc_emit: negative capacity");
    const unsigned long long cv = (unsigned long long)cap_verts, cf = (unsigned long long)cap_faces;
    Grid g;
    if (int rc = make_grid(n0, n1, n2, &g)) return rc;
    g.halo_low = (flags & SCULPT_MC_SLAB_HALO_LOW) ? 1 : 0;
    g.z_off = axis0_offset;
    SC_REQUIRE(vol && workspace && verts && faces, "mc_emit: null argument");
    const WsLayout w = ws_layout(g, 1);   // (the pool comes last: no offset depends on its capacity)
    char *ws = reinterpret_cast<char *>(workspace);
    const int ref = (flags & SCULPT_MC_REFERENCE_ORDER) ? 1 : 0;
    const unsigned *tri = reinterpret_cast<const unsigned *>(ws + w.off_tri);
    const unsigned *vrt = reinterpret_cast<const unsigned *>(ws + w.off_vert);
    const CellRec *recs = reinterpret_cast<const CellRec *>(ws + w.off_recs);
    const McHeader *hdr = reinterpret_cast<const McHeader *>(ws);
    const unsigned long long *gbase = reinterpret_cast<const unsigned long long *>(ws + w.off_gtot);
    const int bpr = cdiv(g.c2, MC_BLOCK), nby = cdiv(g.c1, MC_EY);
    const EdgeIds E{recs, reinterpret_cast<const RowChunk *>(ws + w.off_chunks), vrt, gbase, bpr};
    const int nbx = cdiv(g.c2, MC_EX);
    const long nbricks = (long)nbx * nby * cdiv(g.c0, MC_EZ);
    SC_REQUIRE(nbricks < 0x7fffffffL, "mc_emit: volume too large");
    // one workgroup per brick (a persistent grid of 6 per CU: 193 us against 124; 128 threads per workgroup: 190)
    const int egrid = (int)nbricks, ethreads = MC_BLOCK;
    if (flags & SCULPT_MC_FACES_I64)
        hipLaunchKernelGGL(mc_emit_brick_kernel<long long>, dim3(egrid), dim3(ethreads), 0, st, vol, g, (double)level, recs, E,
                           hdr, tri, nby, nbx, (int)nbricks, verts, vert_div, vert_mul, vert_add, ref,
                           reinterpret_cast<long long *>(faces), ref, cv, cf);
    else
        hipLaunchKernelGGL(mc_emit_brick_kernel<int>, dim3(egrid), dim3(ethreads), 0, st, vol, g, (double)level, recs, E, hdr,
                           tri, nby, nbx, (int)nbricks, verts, vert_div, vert_mul, vert_add, ref, reinterpret_cast<int *>(faces), ref,
                           cv, cf);
    SC_LAUNCH_CHECK();
    if (top_plane_map) {
        hipLaunchKernelGGL(mc_top_plane_kernel, dim3(cdiv(2L * n1 * n2, 256)), dim3(256), 0, st, vol, g, (double)level,
                           E, hdr, top_plane_map);
        SC_LAUNCH_CHECK();
    }
    return 0;
}

int sculpt_mc_emit(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                   float vert_div, float vert_mul, float vert_add, int axis0_offset, float *verts, void *faces,
                   int *top_plane_map, sculpt_stream_t stream) {
    return sculpt_mc_emit_capped(vol, n0, n1, n2, level, flags, workspace, vert_div, vert_mul, vert_add, axis0_offset, verts,
                                 INT64_MAX, faces, INT64_MAX, top_plane_map, stream);
}

}  // extern "C"
