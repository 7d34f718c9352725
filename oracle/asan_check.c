/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * AddressSanitizer / UBSan driver for the C oracle (sanitizers run on the CPU build only -- GPU ASan is
 * not available on the pool).  `make -C oracle asan` builds the three oracle sources together with this file under
 * -fsanitize=address,undefined and runs it: marching cubes on ragged / minimal / noisy / constant volumes, the triplane
 * query on border and out-of-range points, the dense lattice on a partial range, the UV baker on a tiny mesh.
 * Exit code 0 = every call returned what it should and the sanitizers stayed silent. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_marching_cubes(const float *vol, int n0, int n1, int n2, double level, int use_classic, float **verts_out,
                          int *nv_out, int32_t **faces_out, int *nf_out);
void oracle_mc_free(void *p);
void oracle_query_triplane(const float *planes, int C, int H, int W, const float *pts, int64_t N, float radius,
                           float density_bias, int n_layers, const int *dims, const float *const *Wt, const float *const *bs,
                           float *density, float *features, float *density_act, float *color);
void oracle_density_grid(const float *planes, int C, int H, int W, int R, float radius, float density_bias, int n_layers,
                         const int *dims, const float *const *Wt, const float *const *bs, int64_t begin, int64_t end, float *out);
void oracle_grid_point(int64_t flat, int R, float radius, float *p);
void oracle_bake_rasterize(const float *uv, size_t nv, const int *idx, size_t nf, int res, float *out);
void oracle_bake_interpolate(const float *attr, const int *idx, const float *rast, int res, float *out);
void oracle_set_threads(int n);

static uint64_t rng = 0x9e3779b97f4a7c15ull;
static float frand(void) {  /* xorshift64*, uniform in [-1, 1) */
    rng ^= rng >> 12; rng ^= rng << 25; rng ^= rng >> 27;
    return (float)((double)((rng * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "asan_check: %s failed (line %d)\n", #cond, __LINE__); return 1; } } while (0)

static int run_mc(int n0, int n1, int n2, int kind, int classic, int expect) {
    const size_t n = (size_t)n0 * n1 * n2;
    float *vol = (float *)malloc(sizeof(float) * n);  /* exact size: an out-of-range corner read is an ASan error */
    for (size_t i = 0; i < n; ++i) {
        const int z = (int)(i / ((size_t)n1 * n2)), y = (int)(i / n2 % n1), x = (int)(i % n2);
        if (kind == 0) vol[i] = frand();                                           /* noise: every ambiguous case */
        else if (kind == 1) vol[i] = 1.0f;                                         /* constant: level outside the range */
        else if (kind == 2) vol[i] = (float)((x + y + z) % 3) - 1.0f;              /* integers: degenerate saddles */
        else vol[i] = 0.6f * (float)(n0 < n1 ? n0 : n1) * 0.5f - sqrtf((float)((x - n2 / 2) * (x - n2 / 2) + (y - n1 / 2) * (y - n1 / 2) + (z - n0 / 2) * (z - n0 / 2)));
    }
    float *v = NULL; int32_t *f = NULL; int nv = 0, nf = 0;
    const int rc = oracle_marching_cubes(vol, n0, n1, n2, 0.0, classic, &v, &nv, &f, &nf);
    free(vol);
    CHECK(rc == expect);
    if (rc == 0) {
        CHECK(nv > 0 && nf > 0);
        for (int i = 0; i < 3 * nf; ++i) CHECK(f[i] >= 0 && f[i] < nv);
        for (int i = 0; i < nv; ++i) {
            CHECK(v[3 * i] >= 0.f && v[3 * i] <= (float)(n0 - 1) && v[3 * i + 1] >= 0.f && v[3 * i + 1] <= (float)(n1 - 1));
            CHECK(v[3 * i + 2] >= 0.f && v[3 * i + 2] <= (float)(n2 - 1));
        }
    }
    oracle_mc_free(v); oracle_mc_free(f);
    return 0;
}

int main(void) {
    oracle_set_threads(2);
    /* ---- marching cubes */
    if (run_mc(2, 2, 2, 0, 0, 0)) return 1;
    if (run_mc(9, 8, 7, 0, 0, 0)) return 1;
    if (run_mc(17, 5, 33, 0, 0, 0)) return 1;
    if (run_mc(12, 12, 12, 2, 0, 0)) return 1;
    if (run_mc(24, 20, 28, 3, 0, 0)) return 1;
    if (run_mc(10, 10, 10, 0, 1, 0)) return 1;    /* classic tables */
    if (run_mc(6, 6, 6, 1, 0, 1)) return 1;        /* level outside the data range */
    if (run_mc(1, 4, 4, 0, 0, 3)) return 1;        /* bad shape */
    /* ---- triplane query + dense lattice: full-size decoder (120 -> 64 x 9 -> 4), planes 3 x 40 x 64 x 64 */
    enum { C = 40, H = 64, W = 64, NL = 10 };
    int dims[NL + 1]; dims[0] = 3 * C; for (int i = 1; i < NL; ++i) dims[i] = 64; dims[NL] = 4;
    float *Wt[NL], *bs[NL];
    for (int l = 0; l < NL; ++l) {
        Wt[l] = (float *)malloc(sizeof(float) * dims[l] * dims[l + 1]);
        bs[l] = (float *)malloc(sizeof(float) * dims[l + 1]);
        for (int i = 0; i < dims[l] * dims[l + 1]; ++i) Wt[l][i] = frand() * sqrtf(6.0f / (float)dims[l]);
        for (int i = 0; i < dims[l + 1]; ++i) bs[l][i] = 0.1f * frand();
    }
    float *planes = (float *)malloc(sizeof(float) * 3 * C * H * W);
    for (size_t i = 0; i < (size_t)3 * C * H * W; ++i) planes[i] = frand();
    enum { N = 1500 };
    float *pts = (float *)malloc(sizeof(float) * 3 * N);
    for (int i = 0; i < 3 * N; ++i) pts[i] = 0.87f * frand();
    const float corner[] = {0.87f, 0.87f, 0.87f, -0.87f, -0.87f, -0.87f, 0.87f, -0.87f, 0.f, 3.f, -5.f, 0.2f, NAN, 0.f, 0.f};
    memcpy(pts, corner, sizeof(corner));           /* exact borders, far outside, NaN */
    float *d = (float *)malloc(sizeof(float) * N), *ft = (float *)malloc(sizeof(float) * 3 * N);
    float *da = (float *)malloc(sizeof(float) * N), *col = (float *)malloc(sizeof(float) * 3 * N);
    oracle_query_triplane(planes, C, H, W, pts, N, 0.87f, -1.0f, NL, dims, (const float *const *)Wt, (const float *const *)bs, d, ft, da, col);
    for (int i = 5; i < N; ++i) CHECK(isfinite(d[i]) && da[i] > 0.f && col[3 * i] >= 0.f && col[3 * i] <= 1.f);
    enum { R = 9 };
    float *grid = (float *)malloc(sizeof(float) * 200);
    oracle_density_grid(planes, C, H, W, R, 0.87f, -1.0f, NL, dims, (const float *const *)Wt, (const float *const *)bs, 529, 729, grid);
    for (int i = 0; i < 200; ++i) CHECK(grid[i] > 0.f);
    float p[3];
    oracle_grid_point(728, R, 0.87f, p);
    CHECK(p[0] == 0.87f && p[1] == 0.87f && p[2] == 0.87f);
    /* ---- UV baker: two triangles covering part of the texture */
    const float uv[] = {0.1f, 0.1f, 0.9f, 0.1f, 0.9f, 0.9f, 0.1f, 0.9f};
    const int idx[] = {0, 1, 2, 0, 2, 3};
    const float attr[] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 1};
    enum { RES = 23 };
    float *rast = (float *)malloc(sizeof(float) * 4 * RES * RES), *tex = (float *)malloc(sizeof(float) * 3 * RES * RES);
    oracle_bake_rasterize(uv, 4, idx, 2, RES, rast);
    oracle_bake_interpolate(attr, idx, rast, RES, tex);
    int hit = 0;
    for (int i = 0; i < RES * RES; ++i) { CHECK(rast[4 * i + 3] >= -1.f && rast[4 * i + 3] < 2.f); hit += rast[4 * i + 3] >= 0.f; }
    CHECK(hit > RES * RES / 2);
    for (int l = 0; l < NL; ++l) { free(Wt[l]); free(bs[l]); }
    free(planes); free(pts); free(d); free(ft); free(da); free(col); free(grid); free(rast); free(tex);
    printf("asan_check ok\n");
    return 0;
}
