/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called from the
 * product path (sculptmate_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * CPU restatement (plain C, fp32 like the reference) of the triplane density/colour query:
 *   TriplaneNeRFRenderer.query_triplane   /root/reference/TripoSR/tsr/models/nerf_renderer.py:41-91
 *   scale_tensor                          /root/reference/TripoSR/tsr/utils.py:222-231
 *   F.grid_sample(bilinear, align_corners=False, zeros padding)   nerf_renderer.py:61-66
 *       (torch semantics: unnormalise ix = ((g+1)*W-1)/2, 4 taps, out-of-range taps contribute 0)
 *   NeRFMLP.forward                       /root/reference/TripoSR/tsr/models/network_utils.py:116-124
 *       Linear(120,64)+SiLU, 8x[Linear(64,64)+SiLU], Linear(64,4)      network_utils.py:48-79
 *   MarchingCubeHelper.grid_vertices      /root/reference/TripoSR/tsr/models/isosurface.py:25-39
 *
 * PARITY PIN: checked against tests/golden/query_triplane.npz, produced by importing the
 * reference's own modules in the build container (tests/golden/make_reference_goldens.py).
 * Tolerance (fp32, different summation order than torch's sgemm): see tests/test_oracle_query.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* grid lattice coordinate in [0,1]: torch.linspace(0,1,R)[i] (isosurface.py:28-32).
 * torch's CPU kernel evaluates start+step*i for the first half and end-step*(R-1-i) for the
 * second (its SIMD path may differ from this scalar form by 1 ulp; documented in DESIGN.md). */
static inline float lin01(int i, int R) {
    float step = (1.0f - 0.0f) / (float)(R - 1);
    int half = R / 2;
    return (i < half) ? 0.0f + step * (float)i : 1.0f - step * (float)(R - i - 1);
}

/* grid point i (flat index ix*R*R + iy*R + iz) -> position in (-radius, radius):
 * scale_tensor(g, (0,1), (-radius, radius))  (system.py:177-181) */
void oracle_grid_point(int64_t flat, int R, float radius, float *p) {
    int iz = (int)(flat % R), iy = (int)((flat / R) % R), ix = (int)(flat / ((int64_t)R * R));
    float span = (float)((double)radius - (double)(-radius)); /* python double, cast at the multiply */
    float g[3] = {lin01(ix, R), lin01(iy, R), lin01(iz, R)};
    for (int k = 0; k < 3; ++k) {
        float d = (g[k] - 0.0f) / (1.0f - 0.0f);
        p[k] = d * span + (-radius);
    }
}

static inline float silu(float x) { return x / (1.0f + expf(-x)); }

/*
 * planes [3][C][H][W] fp32, pts [N][3] (in (-radius, radius)), MLP: n_layers linear layers,
 * W[l] is [dout_l][din_l] row major (torch.nn.Linear.weight), b[l] is [dout_l].
 * dims = {din_0, dout_0(=din_1), ..., dout_last}  (n_layers+1 entries; max width 128)
 * Outputs (any may be NULL): density [N], features [N][3], density_act [N], color [N][3].
 */
static void query_range(const float *planes, int C, int H, int W, const float *pts, int64_t N0,
                        int64_t N1, float radius, float density_bias, int n_layers, const int *dims,
                        const float *const *Wt, const float *const *bs, float *density,
                        float *features, float *density_act, float *color) {
    const float lo = -radius;
    const float span = (float)((double)radius - (double)(-radius));
    for (int64_t n = N0; n < N1; ++n) {
        float q[3];
        for (int k = 0; k < 3; ++k) {
            /* scale_tensor(p, (-r, r), (-1, 1))  nerf_renderer.py:52-54 */
            float d = (pts[3 * n + k] - lo) / span;
            q[k] = d * (1.0f - (-1.0f)) + (-1.0f);
        }
        float a[128], t[128];
        /* indices2D = (x,y), (x,z), (y,z)   nerf_renderer.py:57-60; grid[...,0]->W, [...,1]->H */
        const int ia[3] = {0, 0, 1}, ib[3] = {1, 2, 2};
        for (int pl = 0; pl < 3; ++pl) {
            float gx = q[ia[pl]], gy = q[ib[pl]];
            float fx = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
            float fy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
            float x0f = floorf(fx), y0f = floorf(fy);
            float wx = fx - x0f, ex = 1.0f - wx, wy = fy - y0f, ey = 1.0f - wy;
            float nw = ey * ex, ne = ey * wx, sw = wy * ex, se = wy * wx;
            long x0 = (long)x0f, y0 = (long)y0f, x1 = x0 + 1, y1 = y0 + 1;
            int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
            int vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
            for (int c = 0; c < C; ++c) {
                const float *P = planes + ((size_t)pl * C + c) * H * W;
                float v = 0.0f;
                if (vy0 && vx0) v += P[y0 * W + x0] * nw;
                if (vy0 && vx1) v += P[y0 * W + x1] * ne;
                if (vy1 && vx0) v += P[y1 * W + x0] * sw;
                if (vy1 && vx1) v += P[y1 * W + x1] * se;
                a[pl * C + c] = v; /* concat: feature = plane*C + channel (nerf_renderer.py:68) */
            }
        }
        for (int l = 0; l < n_layers; ++l) {
            int din = dims[l], dout = dims[l + 1];
            for (int o = 0; o < dout; ++o) {
                const float *w = Wt[l] + (size_t)o * din;
                float s = 0.0f;
                for (int i = 0; i < din; ++i) s += w[i] * a[i];
                s += bs[l][o];
                t[o] = (l + 1 < n_layers) ? silu(s) : s;
            }
            for (int o = 0; o < dout; ++o) a[o] = t[o];
        }
        if (density) density[n] = a[0];
        if (features) { features[3 * n] = a[1]; features[3 * n + 1] = a[2]; features[3 * n + 2] = a[3]; }
        if (density_act) density_act[n] = expf(a[0] + density_bias);
        if (color) for (int k = 0; k < 3; ++k) color[3 * n + k] = 1.0f / (1.0f + expf(-a[1 + k]));
    }
}

void oracle_query_triplane(const float *planes, int C, int H, int W, const float *pts, int64_t N,
                           float radius, float density_bias, int n_layers, const int *dims,
                           const float *const *Wt, const float *const *bs, float *density,
                           float *features, float *density_act, float *color) {
    const int64_t CH = 1024;
#pragma omp parallel for schedule(dynamic)
    for (int64_t c = 0; c < (N + CH - 1) / CH; ++c) {
        int64_t n0 = c * CH, n1 = n0 + CH < N ? n0 + CH : N;
        query_range(planes, C, H, W, pts, n0, n1, radius, density_bias, n_layers, dims, Wt, bs,
                    density, features, density_act, color);
    }
}

/* dense grid: density_act for flat range [begin, end) of the R^3 lattice  (system.py:171-183) */
void oracle_density_grid(const float *planes, int C, int H, int W, int R, float radius,
                         float density_bias, int n_layers, const int *dims, const float *const *Wt,
                         const float *const *bs, int64_t begin, int64_t end, float *out) {
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = begin; i < end; ++i) {
        float p[3];
        oracle_grid_point(i, R, radius, p);
        query_range(planes, C, H, W, p, 0, 1, radius, density_bias, n_layers, dims, Wt, bs,
                    NULL, NULL, out + (i - begin), NULL);
    }
}

#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int oracle_max_threads(void) { return omp_get_max_threads(); }
#else
void oracle_set_threads(int n) { (void)n; }
int oracle_max_threads(void) { return 1; }
#endif
