/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never used by sculptmate_amd/).
 *
 * CPU restatement of the UV-space texture baker of StableFast:
 *   rasterize_cpu    /root/reference/StableFast/sf3d/texture_baker/common.py:123-142 (+ :104-121 barycentrics)
 *   interpolate_cpu  /root/reference/StableFast/sf3d/texture_baker/common.py:214-229
 * The reference itself calls a Windows-only texture_baker.dll (baker.py:31-57, 91-118) whose source is
 * not in the repository; common.py is the only in-repo statement of its semantics.  "First hit wins" in
 * common.py depends on its BVH traversal order; this restatement (and the HIP kernel) take the LOWEST
 * triangle index, which is the same thing wherever UV triangles do not overlap.
 * PARITY PIN: tests/golden/baker.npz, produced by running the reference's common.py
 * (tests/golden/make_reference_goldens.py baker).  Where charts overlap: parity unpinned.
 */
#include <stddef.h>

typedef struct { float u, v, w; } bary_t;

static bary_t barycentric(float px, float py, float ax, float ay, float bx, float by, float cx, float cy) {
    const float e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay, qx = px - ax, qy = py - ay;
    const float d00 = e1x * e1x + e1y * e1y;
    const float d01 = e1x * e2x + e1y * e2y;
    const float d11 = e2x * e2x + e2y * e2y;
    const float d20 = qx * e1x + qy * e1y;
    const float d21 = qx * e2x + qy * e2y;
    const float denom = d00 * d11 - d01 * d01;
    bary_t b;
    b.v = (d11 * d20 - d01 * d21) / denom;
    b.w = (d00 * d21 - d01 * d20) / denom;
    b.u = 1.0f - b.v - b.w;
    return b;
}

void oracle_bake_rasterize(const float *uv, size_t nv, const int *idx, size_t nf, int res, float *out) {
    (void)nv;
    const float R = (float)res;
    for (int y = 0; y < res; ++y)
        for (int x = 0; x < res; ++x) {
            const float px = (float)x / R, py = 1.0f - (float)y / R;
            float *o = out + 4 * ((size_t)y * res + x);
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = -1.f;
            for (size_t t = 0; t < nf; ++t) {
                const int i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
                bary_t b = barycentric(px, py, uv[2 * i0], uv[2 * i0 + 1], uv[2 * i1], uv[2 * i1 + 1], uv[2 * i2], uv[2 * i2 + 1]);
                if (b.u >= 0.f && b.v >= 0.f && b.w >= 0.f) { o[0] = b.u; o[1] = b.v; o[2] = b.w; o[3] = (float)t; break; }
            }
        }
}

void oracle_bake_interpolate(const float *attr, const int *idx, const float *rast, int res, float *out) {
    for (size_t p = 0; p < (size_t)res * res; ++p) {
        const float *r = rast + 4 * p;
        float *o = out + 3 * p;
        o[0] = o[1] = o[2] = 0.f;
        if (r[3] < 0.f) continue;
        const int t = (int)r[3];
        const int i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
        for (int k = 0; k < 3; ++k) o[k] = attr[3 * i0 + k] * r[0] + attr[3 * i1 + k] * r[1] + attr[3 * i2 + k] * r[2];
    }
}
