/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called from the
 * product path (sculptmate_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * CPU restatement of the marching-cubes step of the reference hot path:
 *   /root/reference/TripoSR/tsr/models/isosurface.py:41-54  (MarchingCubeHelper.forward)
 *     -> skimage.measure.marching_cubes(vol, 0.0)            (isosurface.py:46-48)
 * The arithmetic lives in a third-party dependency that is NOT under /root/reference:
 * scikit-image (unpinned in the reference: requirements.txt:5, __init__.py:44).  This file
 * restates the published algorithm scikit-image implements -- Lewiner, Lopes, Vieira,
 * Tavares, "Efficient implementation of Marching Cubes' cases with topological guarantees",
 * JGT 8(2) 2003 -- in the form scikit-image's Cython port runs it (sequential sweep with the
 * slowest array axis outermost, vertices shared through two "face layers", vertex position =
 * inverse-|value| weighted mean of the edge end points in double precision, faces flipped for
 * gradient_direction="descent").
 *
 * PARITY PIN: bit-exact (vertices and faces) against scikit-image 0.18.3 run in the build
 * container (/opt/conda/bin/python3.9) on the golden volumes under tests/golden/mc_*.npz
 * (generator: tests/golden/make_mc_goldens.py) and on a live single-cell / random-volume fuzz
 * (tests/test_oracle_mc_vs_skimage.py, skipped where that interpreter is absent).
 *
 * Output convention = what skimage.measure.marching_cubes returns with default arguments:
 *   verts  float32 [nv,3]  columns (axis0, axis1, axis2) in voxel units
 *   faces  int32   [nf,3]  (already flipped for 'descent')
 * The reference's extra steps (faces[:, [1,0,2]], verts/(R-1)) are applied by the caller.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mc_luts.h"

/* skimage: `cdef double FLT_EPSILON = np.spacing(1.0)`  (== DBL_EPSILON) */
static const double MC_EPS = 2.220446049250313e-16;

typedef struct {
    /* growing outputs */
    float *verts; int nv, cap_v;
    int32_t *faces; int nfi, cap_f; /* nfi = number of indices (3*nf) */
    /* face layers: 4 slots per lattice column (x-edge, y-edge, z-edge, centre) */
    int *layer1, *layer2;
    int nx, ny, nz;
    /* current cell */
    int x, y, z;
    double v[8];     /* v0..v7 (Lewiner corner order), iso subtracted */
    double vv[8];    /* indexed dz*4+dy*2+dx */
    int index;
    int v12_done; double v12x, v12y, v12z;
} mc_state;

static int add_vertex(mc_state *s, double x, double y, double z) {
    if (s->nv == s->cap_v) {
        s->cap_v = s->cap_v ? s->cap_v * 2 : 4096;
        s->verts = (float *)realloc(s->verts, sizeof(float) * 3 * (size_t)s->cap_v);
    }
    s->verts[3 * s->nv + 0] = (float)x;
    s->verts[3 * s->nv + 1] = (float)y;
    s->verts[3 * s->nv + 2] = (float)z;
    return s->nv++;
}
static void add_face_index(mc_state *s, int vi) {
    if (s->nfi == s->cap_f) {
        s->cap_f = s->cap_f ? s->cap_f * 2 : 16384;
        s->faces = (int32_t *)realloc(s->faces, sizeof(int32_t) * (size_t)s->cap_f);
    }
    s->faces[s->nfi++] = vi;
}

/* slot of an edge vertex in the face layers; returns pointer to the slot */
static int *layer_slot(mc_state *s, int vi) {
    int i = s->nx * s->y + s->x, j = 0;
    int *layer;
    if (vi < 8) {
        if (vi < 4) layer = s->layer1; else { vi -= 4; layer = s->layer2; }
        if (vi == 1) { i += 1; j = 1; }
        else if (vi == 2) { i += s->nx; }
        else if (vi == 3) { j = 1; }
    } else if (vi < 12) {
        layer = s->layer1; j = 2;
        if (vi == 9) i += 1;
        else if (vi == 10) i += s->nx + 1;
        else if (vi == 11) i += s->nx;
    } else { layer = s->layer1; j = 3; }
    return &layer[4 * i + j];
}

static void center_vertex(mc_state *s) {
    /* corner order v0..v7 with positions (0,0,0)(1,0,0)(1,1,0)(0,1,0)(0,0,1)(1,0,1)(1,1,1)(0,1,1) */
    static const double px[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    static const double py[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    static const double pz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    double fx = 0, fy = 0, fz = 0, ff = 0;
    for (int k = 0; k < 8; ++k) {
        double w = 1.0 / (MC_EPS + fabs(s->v[k]));
        fx += px[k] * w; fy += py[k] * w; fz += pz[k] * w; ff += w;
    }
    s->v12x = (double)s->x + 1.0 * fx / ff;
    s->v12y = (double)s->y + 1.0 * fy / ff;
    s->v12z = (double)s->z + 1.0 * fz / ff;
    s->v12_done = 1;
}

static void add_from_edge(mc_state *s, int vi) {
    int *slot = layer_slot(s, vi);
    if (*slot >= 0) { add_face_index(s, *slot); return; }
    int idx;
    if (vi == 12) {
        if (!s->v12_done) center_vertex(s);
        idx = add_vertex(s, s->v12x, s->v12y, s->v12z);
    } else {
        int dx1 = mc_edge_relx[2 * vi], dx2 = mc_edge_relx[2 * vi + 1];
        int dy1 = mc_edge_rely[2 * vi], dy2 = mc_edge_rely[2 * vi + 1];
        int dz1 = mc_edge_relz[2 * vi], dz2 = mc_edge_relz[2 * vi + 1];
        double t1 = 1.0 / (MC_EPS + fabs(s->vv[dz1 * 4 + dy1 * 2 + dx1]));
        double t2 = 1.0 / (MC_EPS + fabs(s->vv[dz2 * 4 + dy2 * 2 + dx2]));
        double fx = 0, fy = 0, fz = 0, ff = 0;
        fx += (double)dx1 * t1; fy += (double)dy1 * t1; fz += (double)dz1 * t1; ff += t1;
        fx += (double)dx2 * t2; fy += (double)dy2 * t2; fz += (double)dz2 * t2; ff += t2;
        idx = add_vertex(s, (double)s->x + 1.0 * fx / ff, (double)s->y + 1.0 * fy / ff,
                         (double)s->z + 1.0 * fz / ff);
    }
    *slot = idx;
    add_face_index(s, idx);
}

static void add_triangles(mc_state *s, int table, int row, int sub) {
    int rowlen = mc_tiling_rowlen[table];
    const signed char *t = mc_tiling_flat + mc_tiling_base[table] +
                           (row * mc_tiling_inner[table] + sub) * rowlen;
    s->vv[0] = s->v[0]; s->vv[1] = s->v[1]; s->vv[2] = s->v[3]; s->vv[3] = s->v[2];
    s->vv[4] = s->v[4]; s->vv[5] = s->v[5]; s->vv[6] = s->v[7]; s->vv[7] = s->v[6];
    for (int i = 0; i < rowlen; ++i) add_from_edge(s, t[i]);
}

/* Lewiner test_face: does the face contain part of the surface (saddle sign) */
static int test_face(const double *v, int face) {
    int af = face < 0 ? -face : face;
    double A, B, C, D;
    switch (af) {
    case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
    case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
    case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
    case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
    case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
    default: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break; /* 6 */
    }
    double acbd = A * C - B * D;
    if (acbd > -MC_EPS && acbd < MC_EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

/* Lewiner test_interior (Chernyaev's interior ambiguity test) */
static int test_internal(const double *v, int mc_case, int config, int subconfig, int s) {
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
    int test = 0, edge = -1;
    if (mc_case == 4 || mc_case == 10) {
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        if (mc_case == 6) edge = mc_test6[config * 3 + 2];
        else if (mc_case == 7) edge = mc_test7[config * 5 + 4];
        else if (mc_case == 12) edge = mc_test12[config * 4 + 3];
        else /* 13 */ edge = mc_tiling_flat[mc_tiling_base[MC_T_13_5_1] + (config * 4 + subconfig) * 18 + 0];
        switch (edge) {
        case 0: t = v[0] / (v[0] - v[1] + MC_EPS); At = 0; Bt = v[3] + (v[2] - v[3]) * t; Ct = v[7] + (v[6] - v[7]) * t; Dt = v[4] + (v[5] - v[4]) * t; break;
        case 1: t = v[1] / (v[1] - v[2] + MC_EPS); At = 0; Bt = v[0] + (v[3] - v[0]) * t; Ct = v[4] + (v[7] - v[4]) * t; Dt = v[5] + (v[6] - v[5]) * t; break;
        case 2: t = v[2] / (v[2] - v[3] + MC_EPS); At = 0; Bt = v[1] + (v[0] - v[1]) * t; Ct = v[5] + (v[4] - v[5]) * t; Dt = v[6] + (v[7] - v[6]) * t; break;
        case 3: t = v[3] / (v[3] - v[0] + MC_EPS); At = 0; Bt = v[2] + (v[1] - v[2]) * t; Ct = v[6] + (v[5] - v[6]) * t; Dt = v[7] + (v[4] - v[7]) * t; break;
        case 4: t = v[4] / (v[4] - v[5] + MC_EPS); At = 0; Bt = v[7] + (v[6] - v[7]) * t; Ct = v[3] + (v[2] - v[3]) * t; Dt = v[0] + (v[1] - v[0]) * t; break;
        case 5: t = v[5] / (v[5] - v[6] + MC_EPS); At = 0; Bt = v[4] + (v[7] - v[4]) * t; Ct = v[0] + (v[3] - v[0]) * t; Dt = v[1] + (v[2] - v[1]) * t; break;
        case 6: t = v[6] / (v[6] - v[7] + MC_EPS); At = 0; Bt = v[5] + (v[4] - v[5]) * t; Ct = v[1] + (v[0] - v[1]) * t; Dt = v[2] + (v[3] - v[2]) * t; break;
        case 7: t = v[7] / (v[7] - v[4] + MC_EPS); At = 0; Bt = v[6] + (v[5] - v[6]) * t; Ct = v[2] + (v[1] - v[2]) * t; Dt = v[3] + (v[0] - v[3]) * t; break;
        case 8: t = v[0] / (v[0] - v[4] + MC_EPS); At = 0; Bt = v[3] + (v[7] - v[3]) * t; Ct = v[2] + (v[6] - v[2]) * t; Dt = v[1] + (v[5] - v[1]) * t; break;
        case 9: t = v[1] / (v[1] - v[5] + MC_EPS); At = 0; Bt = v[0] + (v[4] - v[0]) * t; Ct = v[3] + (v[7] - v[3]) * t; Dt = v[2] + (v[6] - v[2]) * t; break;
        case 10: t = v[2] / (v[2] - v[6] + MC_EPS); At = 0; Bt = v[1] + (v[5] - v[1]) * t; Ct = v[0] + (v[4] - v[0]) * t; Dt = v[3] + (v[7] - v[3]) * t; break;
        case 11: t = v[3] / (v[3] - v[7] + MC_EPS); At = 0; Bt = v[2] + (v[6] - v[2]) * t; Ct = v[1] + (v[5] - v[1]) * t; Dt = v[0] + (v[4] - v[0]) * t; break;
        default: break;
        }
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    /* scikit-image quirk (differs from Lewiner's C++ `break; return s<0`): in the Cython
     * port the if/elif chain ends here, so a failed inner condition falls off the end of
     * the cdef function and yields 0.  Pinned by the single-cell fuzz against skimage. */
    case 5: if (At * Ct - Bt * Dt < MC_EPS) return s > 0; return 0;
    case 10: if (At * Ct - Bt * Dt >= MC_EPS) return s > 0; return 0;
    default: break; /* 7 11 13 14 15 */
    }
    return s < 0;
}

/*
 * Classify one cell: returns number of triangles (0 if none) and the tiling
 * (table id, row, sub).  v[8] = corner values minus iso (double), Lewiner corner order.
 */
int oracle_mc_classify(const double *v, int *table, int *row, int *sub) {
    int index = 0;
    for (int k = 0; k < 8; ++k) if (v[k] > 0.0) index |= 1 << k;
    int c = mc_cases[2 * index], cfg = mc_cases[2 * index + 1], sc = 0;
    *sub = 0; *row = cfg;
    switch (c) {
    case 0: *table = -1; return 0;
    case 1: *table = MC_T_1; break;
    case 2: *table = MC_T_2; break;
    case 3: *table = test_face(v, mc_test3[cfg]) ? MC_T_3_2 : MC_T_3_1; break;
    case 4: *table = test_internal(v, c, cfg, 0, mc_test4[cfg]) ? MC_T_4_1 : MC_T_4_2; break;
    case 5: *table = MC_T_5; break;
    case 6:
        if (test_face(v, mc_test6[cfg * 3 + 0])) *table = MC_T_6_2;
        else *table = test_internal(v, c, cfg, 0, mc_test6[cfg * 3 + 1]) ? MC_T_6_1_1 : MC_T_6_1_2;
        break;
    case 7:
        if (test_face(v, mc_test7[cfg * 5 + 0])) sc += 1;
        if (test_face(v, mc_test7[cfg * 5 + 1])) sc += 2;
        if (test_face(v, mc_test7[cfg * 5 + 2])) sc += 4;
        switch (sc) {
        case 0: *table = MC_T_7_1; break;
        case 1: *table = MC_T_7_2; *sub = 0; break;
        case 2: *table = MC_T_7_2; *sub = 1; break;
        case 3: *table = MC_T_7_3; *sub = 0; break;
        case 4: *table = MC_T_7_2; *sub = 2; break;
        case 5: *table = MC_T_7_3; *sub = 1; break;
        case 6: *table = MC_T_7_3; *sub = 2; break;
        default: *table = test_internal(v, c, cfg, sc, mc_test7[cfg * 5 + 3]) ? MC_T_7_4_2 : MC_T_7_4_1; break;
        }
        break;
    case 8: *table = MC_T_8; break;
    case 9: *table = MC_T_9; break;
    case 10:
        if (test_face(v, mc_test10[cfg * 3 + 0])) {
            *table = test_face(v, mc_test10[cfg * 3 + 1]) ? MC_T_10_1_1_ : MC_T_10_2;
        } else {
            if (test_face(v, mc_test10[cfg * 3 + 1])) *table = MC_T_10_2_;
            else *table = test_internal(v, c, cfg, 0, mc_test10[cfg * 3 + 2]) ? MC_T_10_1_1 : MC_T_10_1_2;
        }
        break;
    case 11: *table = MC_T_11; break;
    case 12:
        if (test_face(v, mc_test12[cfg * 4 + 0])) {
            *table = test_face(v, mc_test12[cfg * 4 + 1]) ? MC_T_12_1_1_ : MC_T_12_2;
        } else {
            if (test_face(v, mc_test12[cfg * 4 + 1])) *table = MC_T_12_2_;
            else *table = test_internal(v, c, cfg, 0, mc_test12[cfg * 4 + 2]) ? MC_T_12_1_1 : MC_T_12_1_2;
        }
        break;
    case 13:
        for (int k = 0; k < 6; ++k) if (test_face(v, mc_test13[cfg * 7 + k])) sc += 1 << k;
        sc = mc_subconfig13[sc];
        if (sc == 0) *table = MC_T_13_1;
        else if (sc <= 6) { *table = MC_T_13_2; *sub = sc - 1; }
        else if (sc <= 18) { *table = MC_T_13_3; *sub = sc - 7; }
        else if (sc <= 22) { *table = MC_T_13_4; *sub = sc - 19; }
        else if (sc <= 26) {
            *sub = sc - 23;
            *table = test_internal(v, c, cfg, *sub, mc_test13[cfg * 7 + 6]) ? MC_T_13_5_1 : MC_T_13_5_2;
        }
        else if (sc <= 38) { *table = MC_T_13_3_; *sub = sc - 27; }
        else if (sc <= 44) { *table = MC_T_13_2_; *sub = sc - 39; }
        else if (sc == 45) *table = MC_T_13_1_;
        else { *table = -1; return 0; } /* "Impossible case 13?" */
        break;
    case 14: *table = MC_T_14; break;
    default: *table = -1; return 0;
    }
    return mc_tiling_rowlen[*table] / 3;
}

/*
 * vol: float32 C-order [n0][n1][n2].  Returns 0 ok; 1 = level outside data range
 * (skimage ValueError); 2 = no surface (skimage RuntimeError); 3 = bad shape.
 * Outputs are malloc'd; free with oracle_mc_free.
 */
int oracle_marching_cubes(const float *vol, int n0, int n1, int n2, double level, int use_classic,
                          float **verts_out, int *nv_out, int32_t **faces_out, int *nf_out) {
    *verts_out = NULL; *faces_out = NULL; *nv_out = 0; *nf_out = 0;
    if (n0 < 2 || n1 < 2 || n2 < 2) return 3;
    {
        float mn = vol[0], mx = vol[0];
        size_t n = (size_t)n0 * n1 * n2;
        for (size_t i = 1; i < n; ++i) { if (vol[i] < mn) mn = vol[i]; if (vol[i] > mx) mx = vol[i]; }
        if (level < (double)mn || level > (double)mx) return 1;
    }
    mc_state s; memset(&s, 0, sizeof(s));
    s.nx = n2; s.ny = n1; s.nz = n0;
    size_t ls = (size_t)4 * s.nx * s.ny;
    s.layer1 = (int *)malloc(sizeof(int) * ls);
    s.layer2 = (int *)malloc(sizeof(int) * ls);
    for (size_t i = 0; i < ls; ++i) { s.layer1[i] = -1; s.layer2[i] = -1; }
    const size_t sy = (size_t)n2, sz = (size_t)n1 * n2;
    for (int z = 0; z < n0 - 1; ++z) {
        /* new_z_value: swap layers, clear the upper one */
        int *tmp = s.layer1; s.layer1 = s.layer2; s.layer2 = tmp;
        for (size_t i = 0; i < ls; ++i) s.layer2[i] = -1;
        for (int y = 0; y < n1 - 1; ++y) {
            for (int x = 0; x < n2 - 1; ++x) {
                const float *p = vol + z * sz + y * sy + x;
                s.x = x; s.y = y; s.z = z; s.v12_done = 0;
                s.v[0] = (double)p[0] - level;       s.v[1] = (double)p[1] - level;
                s.v[2] = (double)p[sy + 1] - level;  s.v[3] = (double)p[sy] - level;
                s.v[4] = (double)p[sz] - level;      s.v[5] = (double)p[sz + 1] - level;
                s.v[6] = (double)p[sz + sy + 1] - level; s.v[7] = (double)p[sz + sy] - level;
                if (use_classic) {
                    int index = 0;
                    for (int k = 0; k < 8; ++k) if (s.v[k] > 0.0) index |= 1 << k;
                    const signed char *t = mc_cases_classic + 16 * index;
                    int nt = 0; while (t[3 * nt] != -1) ++nt;
                    if (nt) {
                        s.vv[0] = s.v[0]; s.vv[1] = s.v[1]; s.vv[2] = s.v[3]; s.vv[3] = s.v[2];
                        s.vv[4] = s.v[4]; s.vv[5] = s.v[5]; s.vv[6] = s.v[7]; s.vv[7] = s.v[6];
                        for (int i = 0; i < 3 * nt; ++i) add_from_edge(&s, t[i]);
                    }
                } else {
                    int table, row, sub;
                    if (oracle_mc_classify(s.v, &table, &row, &sub) > 0) add_triangles(&s, table, row, sub);
                }
            }
        }
    }
    free(s.layer1); free(s.layer2);
    if (s.nv == 0) { free(s.verts); free(s.faces); return 2; }
    /* wrapper: vertices fliplr -> (z,y,x) = (axis0,axis1,axis2); faces fliplr (descent) */
    for (int i = 0; i < s.nv; ++i) { float t = s.verts[3 * i]; s.verts[3 * i] = s.verts[3 * i + 2]; s.verts[3 * i + 2] = t; }
    for (int i = 0; i + 2 < s.nfi; i += 3) { int32_t t = s.faces[i]; s.faces[i] = s.faces[i + 2]; s.faces[i + 2] = t; }
    *verts_out = s.verts; *nv_out = s.nv; *faces_out = s.faces; *nf_out = s.nfi / 3;
    return 0;
}

void oracle_mc_free(void *p) { free(p); }
