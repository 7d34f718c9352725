// Box-projection UV unwrapping for StableFast-3D meshes on gfx950.
//
// Replaces (reference file:line), one entry point per stage of Unwrapper.forward (StableFast/sf3d/uv_unwrapper/unwrap.py:625-697):
//   sculpt_uv_moments          the statistics behind _align_mesh_with_main_axis (:546-623; the reference runs a randomised
//                              torch.pca_lowrank, the host side here takes the exact principal axes of these moments)
//   sculpt_uv_box_project      rotation into the principal frame + _box_assign_vertex_to_cube_face (:16-122)
//   sculpt_uv_chart_tangents   _calculate_tangents (:239-305) + the per-chart mean tangents of
//                              _rotate_uv_slices_consistent_space (:326-355), including its F.normalize(x, -1) quirk
//                              (the second positional argument of F.normalize is the order p: a p = -1 "norm")
//   sculpt_uv_rotate_charts    the rotation and the joint min / max stretch of every chart (:357-381)
//   sculpt_uv_assign_atlas     assign_faces_uv_to_atlas_index -- in the reference a function of uv_unwrapper.dll whose source
//                              is not available (:124-175).  Own algorithm, same contract (chart c -> c, c + 6 or 12):
//                              a per-chart z-buffer in UV space (64-bit atomicMax of depth | triangle id); a triangle that
//                              loses one of its interior samples to a triangle in front of it moves to the overlap slice
//                              c + 6; the same test among the overlap slices sends the losers to 12 ("remaining", every
//                              triangle in its own square).
//   sculpt_uv_place            _handle_slice_uvs, _handle_remaining_uvs, _find_slice_offset_and_scale,
//                              _distribute_individual_uvs_in_atlas (:177-237, 383-527)
//
// All of it is HBM-bound gather / scatter work: one thread per face or vertex, 32-bit atomics on ordered-float keys for
// the min / max reductions (order independent, exact), double atomics for the few sums, ballot + popcount for ranks.
#include <float.h>
#include <math.h>

#include <vector>

#include "common.h"

namespace sculpt {

namespace {

constexpr int UVB = 256;

__device__ __forceinline__ unsigned uv_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float uv_ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// words of the per-call statistics block (32-bit each)
enum : int {
    ST_BBOX_MIN = 0,   // 3 ordered floats
    ST_BBOX_MAX = 3,   // 3
    ST_DIV = 6,        // 3: max |coordinate along the chart axis| per corner slot
    ST_CH_MIN = 9,     // 6: joint min of the rotated chart
    ST_CH_MAX = 15,    // 6
    ST_SL_UMIN = 21,   // 6: slices 6..11
    ST_SL_UMAX = 27,
    ST_SL_VMIN = 33,
    ST_SL_VMAX = 39,
    ST_REMAINING = 45,  // count of faces with index >= 12
    ST_WORDS = 48
};

__global__ void uv_stats_init_kernel(unsigned *st) {
    const int i = threadIdx.x;
    if (i >= ST_WORDS) return;
    unsigned v = 0u;
    const bool is_min = (i >= ST_BBOX_MIN && i < ST_BBOX_MIN + 3) || (i >= ST_CH_MIN && i < ST_CH_MIN + 6) ||
                        (i >= ST_SL_UMIN && i < ST_SL_UMIN + 6) || (i >= ST_SL_VMIN && i < ST_SL_VMIN + 6);
    const bool is_max = (i >= ST_BBOX_MAX && i < ST_BBOX_MAX + 3) || (i >= ST_DIV && i < ST_DIV + 3) ||
                        (i >= ST_CH_MAX && i < ST_CH_MAX + 6) || (i >= ST_SL_UMAX && i < ST_SL_UMAX + 6) ||
                        (i >= ST_SL_VMAX && i < ST_SL_VMAX + 6);
    if (is_min) v = uv_f2ord(FLT_MAX);
    if (is_max) v = uv_f2ord(-FLT_MAX);
    st[i] = v;
}

template <typename IdxT>
__device__ __forceinline__ void load_face(const IdxT *faces, long f, int &a, int &b, int &c) {
    a = (int)faces[3 * f];
    b = (int)faces[3 * f + 1];
    c = (int)faces[3 * f + 2];
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < UVB / 64; ++w) t += sh[w];
    return t;
}

// ---------------------------------------------------------------------------------------------- moments
__global__ __launch_bounds__(UVB) void uv_moments_kernel(const float *__restrict__ v, long nv, double *__restrict__ out) {
    __shared__ double sh[UVB / 64];
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // x y z xx xy xz yy yz zz
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        const double x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
        s[0] += x; s[1] += y; s[2] += z;
        s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double t = block_sum(s[k], sh);
        if (threadIdx.x == 0) atomicAdd(&out[k], t);
    }
}

// ---------------------------------------------------------------------------------------------- rotation + bbox
struct Rot3 { float m[9]; };

__global__ __launch_bounds__(UVB) void uv_rotate_mesh_kernel(const float *__restrict__ pos, const float *__restrict__ nrm, long nv, Rot3 R,
                                                             float *__restrict__ rpos, float *__restrict__ rnrm, unsigned *__restrict__ st) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        const float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        const float n[3] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = __fadd_rn(__fadd_rn(__fmul_rn(R.m[3 * r], p[0]), __fmul_rn(R.m[3 * r + 1], p[1])), __fmul_rn(R.m[3 * r + 2], p[2]));
            const float b = __fadd_rn(__fadd_rn(__fmul_rn(R.m[3 * r], n[0]), __fmul_rn(R.m[3 * r + 1], n[1])), __fmul_rn(R.m[3 * r + 2], n[2]));
            rpos[3 * i + r] = a;
            rnrm[3 * i + r] = b;
            mn[r] = fminf(mn[r], a);
            mx[r] = fmaxf(mx[r], a);
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = mn[r], b = mx[r];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a = fminf(a, __shfl_xor(a, d, 64));
            b = fmaxf(b, __shfl_xor(b, d, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            const volatile unsigned *sv = st;
            if (uv_f2ord(a) < sv[ST_BBOX_MIN + r]) atomicMin(&st[ST_BBOX_MIN + r], uv_f2ord(a));
            if (uv_f2ord(b) > sv[ST_BBOX_MAX + r]) atomicMax(&st[ST_BBOX_MAX + r], uv_f2ord(b));
        }
    }
}

// ---------------------------------------------------------------------------------------------- box projection
__device__ __forceinline__ float unit_coord(float p, float lo, float hi) {
    return __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(__fsub_rn(p, lo), __fsub_rn(hi, lo))), 1.0f);
}

template <typename IdxT>
__global__ __launch_bounds__(UVB) void uv_box_project_kernel(const float *__restrict__ rpos, const float *__restrict__ rnrm,
                                                             const IdxT *__restrict__ faces, long nf, float *__restrict__ face_uv,
                                                             int *__restrict__ chart, unsigned *__restrict__ st) {
    float lo[3], hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { lo[r] = uv_ord2f(st[ST_BBOX_MIN + r]); hi[r] = uv_ord2f(st[ST_BBOX_MAX + r]); }
    float dmax[3] = {0.f, 0.f, 0.f};
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        int vi[3];
        load_face(faces, f, vi[0], vi[1], vi[2]);
        float tri[3][3], ns[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int r = 0; r < 3; ++r) tri[k][r] = unit_coord(rpos[3 * (long)vi[k] + r], lo[r], hi[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) ns[r] = __fadd_rn(__fadd_rn(rnrm[3 * (long)vi[0] + r], rnrm[3 * (long)vi[1] + r]), rnrm[3 * (long)vi[2] + r]);
        const float len = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ns[0], ns[0]), __fmul_rn(ns[1], ns[1])), __fmul_rn(ns[2], ns[2])));
        const float inv = fmaxf(len, 1e-6f);
        const float fn[3] = {__fdiv_rn(ns[0], inv), __fdiv_rn(ns[1], inv), __fdiv_rn(ns[2], inv)};
        // argmax over (+x, -x, +y, -y, +z, -z) of the component along that direction; first maximum wins
        int c = 0;
        float best = fn[0];
        const float cand[6] = {fn[0], -fn[0], fn[1], -fn[1], fn[2], -fn[2]};
#pragma unroll
        for (int k = 1; k < 6; ++k)
            if (cand[k] > best) { best = cand[k]; c = k; }
        const int ax = c >> 1;
        const int us = ax == 0 ? 1 : 0, vs = ax == 2 ? 1 : 2;
        chart[f] = c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float vv = tri[k][vs];
            face_uv[6 * f + 2 * k] = tri[k][us];
            face_uv[6 * f + 2 * k + 1] = c == 4 ? vv : -vv;
            dmax[k] = fmaxf(dmax[k], fabsf(tri[k][ax]));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = dmax[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a = fmaxf(a, __shfl_xor(a, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&st[ST_DIV + k], uv_f2ord(a));
    }
}

__global__ __launch_bounds__(UVB) void uv_box_finish_kernel(float *__restrict__ face_uv, long nf, const unsigned *__restrict__ st) {
    const float div[3] = {uv_ord2f(st[ST_DIV]), uv_ord2f(st[ST_DIV + 1]), uv_ord2f(st[ST_DIV + 2])};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf * 6; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)((i % 6) >> 1);
        const float x = __fmul_rn(__fadd_rn(__fdiv_rn(face_uv[i], div[k]), 1.0f), 0.5f);
        face_uv[i] = fminf(fmaxf(x, 0.f), 1.f);
    }
}

// ---------------------------------------------------------------------------------------------- tangents
template <typename IdxT>
__global__ __launch_bounds__(UVB) void uv_face_tangent_kernel(const float *__restrict__ rpos, const IdxT *__restrict__ faces, long nf,
                                                              const float *__restrict__ face_uv, float *__restrict__ acc /* [nv][4] */) {
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        int vi[3];
        load_face(faces, f, vi[0], vi[1], vi[2]);
        const float *t = face_uv + 6 * f;
        const float du1 = t[2] - t[0], dv1 = t[3] - t[1], du2 = t[4] - t[0], dv2 = t[5] - t[1];
        const float den = fmaxf(__fsub_rn(__fmul_rn(du1, dv2), __fmul_rn(dv1, du2)), 1e-6f);  // clip(1e-6): negatives too
        float tg[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float p0 = rpos[3 * (long)vi[0] + r];
            const float d1 = rpos[3 * (long)vi[1] + r] - p0, d2 = rpos[3 * (long)vi[2] + r] - p0;
            tg[r] = __fdiv_rn(__fsub_rn(__fmul_rn(d1, dv2), __fmul_rn(d2, dv1)), den);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float *a = acc + 4 * (long)vi[k];
            atomicAdd(a, tg[0]);
            atomicAdd(a + 1, tg[1]);
            atomicAdd(a + 2, tg[2]);
            atomicAdd(a + 3, 1.0f);
        }
    }
}

// acc [nv][4] (sum, count) -> (tangent perpendicular to the normal, unused)
__global__ __launch_bounds__(UVB) void uv_vertex_tangent_kernel(float *__restrict__ acc, const float *__restrict__ rnrm, long nv) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        float *a = acc + 4 * i;
        const float cnt = a[3];
        float t[3] = {a[0] / cnt, a[1] / cnt, a[2] / cnt};
        float len = fmaxf(sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), 1e-12f);
        t[0] /= len; t[1] /= len; t[2] /= len;
        const float n[3] = {rnrm[3 * i], rnrm[3 * i + 1], rnrm[3 * i + 2]};
        const float d = t[0] * n[0] + t[1] * n[1] + t[2] * n[2];
        t[0] -= d * n[0]; t[1] -= d * n[1]; t[2] -= d * n[2];
        len = fmaxf(sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), 1e-12f);
        a[0] = t[0] / len; a[1] = t[1] / len; a[2] = t[2] / len;
    }
}

__device__ __forceinline__ void expected_tangent(const float *p, const float *n, float *e) {
    const float s[3] = {-p[1], p[0], 0.f};
    const float c1[3] = {s[1] * n[2] - s[2] * n[1], s[2] * n[0] - s[0] * n[2], s[0] * n[1] - s[1] * n[0]};       // side x n
    const float c2[3] = {n[1] * c1[2] - n[2] * c1[1], n[2] * c1[0] - n[0] * c1[2], n[0] * c1[1] - n[1] * c1[0]};  // n x (side x n)
    // F.normalize(x, -1): p = -1 "norm" = 1 / (1/|x| + 1/|y| + 1/|z|), clamped at 1e-12
    const float inv = 1.0f / fabsf(c2[0]) + 1.0f / fabsf(c2[1]) + 1.0f / fabsf(c2[2]);
    const float nm = fmaxf(1.0f / inv, 1e-12f);
    e[0] = c2[0] / nm; e[1] = c2[1] / nm; e[2] = c2[2] / nm;
}

// sums [6][7] doubles: sum of the corner tangents (3), of the expected tangents (3), number of corners
template <typename IdxT>
__global__ __launch_bounds__(UVB) void uv_chart_sums_kernel(const float *__restrict__ rpos, const float *__restrict__ rnrm,
                                                            const IdxT *__restrict__ faces, long nf, const int *__restrict__ chart,
                                                            const float *__restrict__ vt /* [nv][4] */, double *__restrict__ sums) {
    __shared__ double sh[UVB / 64];
    double loc[6][7];
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int k = 0; k < 7; ++k) loc[c][k] = 0.0;
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        int vi[3];
        load_face(faces, f, vi[0], vi[1], vi[2]);
        const int c = chart[f] % 6;
        double a[7] = {0, 0, 0, 0, 0, 0, 3.0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long v = vi[k];
            float e[3];
            expected_tangent(rpos + 3 * v, rnrm + 3 * v, e);
            a[0] += vt[4 * v]; a[1] += vt[4 * v + 1]; a[2] += vt[4 * v + 2];
            a[3] += e[0]; a[4] += e[1]; a[5] += e[2];
        }
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
            if (cc == c) {
#pragma unroll
                for (int k = 0; k < 7; ++k) loc[cc][k] += a[k];
            }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double t = block_sum(loc[c][k], sh);
            if (threadIdx.x == 0 && t != 0.0) atomicAdd(&sums[c * 7 + k], t);
        }
}

// ---------------------------------------------------------------------------------------------- chart rotation
struct Rot2x6 { float c[6], s[6]; };

__global__ __launch_bounds__(UVB) void uv_rotate_chart_kernel(float *__restrict__ face_uv, const int *__restrict__ chart, long nf, Rot2x6 R,
                                                              unsigned *__restrict__ st) {
    float mn[6], mx[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { mn[c] = FLT_MAX; mx[c] = -FLT_MAX; }
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        const int c = chart[f] % 6;
        float co = 0.f, si = 0.f;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
            if (cc == c) { co = R.c[cc]; si = R.s[cc]; }
        float lo = FLT_MAX, hi = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = __fsub_rn(__fmul_rn(face_uv[6 * f + 2 * k], 2.0f), 1.0f);
            const float y = __fsub_rn(__fmul_rn(face_uv[6 * f + 2 * k + 1], 2.0f), 1.0f);
            const float u = __fadd_rn(__fmul_rn(co, x), __fmul_rn(-si, y));
            const float v = __fadd_rn(__fmul_rn(si, x), __fmul_rn(co, y));
            face_uv[6 * f + 2 * k] = u;
            face_uv[6 * f + 2 * k + 1] = v;
            lo = fminf(lo, fminf(u, v));
            hi = fmaxf(hi, fmaxf(u, v));
        }
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
            if (cc == c) { mn[cc] = fminf(mn[cc], lo); mx[cc] = fmaxf(mx[cc], hi); }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float a = mn[c], b = mx[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a = fminf(a, __shfl_xor(a, d, 64));
            b = fmaxf(b, __shfl_xor(b, d, 64));
        }
        if ((threadIdx.x & 63) == 0 && a <= b) {
            const volatile unsigned *sv = st;
            if (uv_f2ord(a) < sv[ST_CH_MIN + c]) atomicMin(&st[ST_CH_MIN + c], uv_f2ord(a));
            if (uv_f2ord(b) > sv[ST_CH_MAX + c]) atomicMax(&st[ST_CH_MAX + c], uv_f2ord(b));
        }
    }
}

__global__ __launch_bounds__(UVB) void uv_rescale_chart_kernel(float *__restrict__ face_uv, const int *__restrict__ chart, long nf,
                                                               const unsigned *__restrict__ st) {
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        const int c = chart[f] % 6;
        const float lo = uv_ord2f(st[ST_CH_MIN + c]), hi = uv_ord2f(st[ST_CH_MAX + c]);
        const float span = __fsub_rn(hi, lo);
#pragma unroll
        for (int k = 0; k < 6; ++k) face_uv[6 * f + k] = __fdiv_rn(__fsub_rn(face_uv[6 * f + k], lo), span);
    }
}

// ---------------------------------------------------------------------------------------------- atlas assignment (own algorithm)
// Sample points of a triangle: the pixel centres strictly inside it (barycentrics > EDGE_EPS, so the pixels on an edge
// shared by two neighbours belong to neither).  A triangle without any (smaller than a pixel) draws nothing and is tested
// at its centroid against the triangle that owns that pixel.
constexpr float EDGE_EPS = 1e-4f;

struct TriRaster {
    float x[3], y[3], inv_area;
    int x0, x1, y0, y1;
    bool degenerate;
};

__device__ __forceinline__ TriRaster make_raster(const float *t, int res) {
    TriRaster r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.x[k] = t[2 * k] * res; r.y[k] = t[2 * k + 1] * res; }
    const float area = (r.x[1] - r.x[0]) * (r.y[2] - r.y[0]) - (r.x[2] - r.x[0]) * (r.y[1] - r.y[0]);
    r.degenerate = fabsf(area) < 1e-12f;
    r.inv_area = r.degenerate ? 0.f : 1.0f / area;
    const float xmin = fminf(r.x[0], fminf(r.x[1], r.x[2])), xmax = fmaxf(r.x[0], fmaxf(r.x[1], r.x[2]));
    const float ymin = fminf(r.y[0], fminf(r.y[1], r.y[2])), ymax = fmaxf(r.y[0], fmaxf(r.y[1], r.y[2]));
    r.x0 = max(0, (int)floorf(xmin - 0.5f));
    r.x1 = min(res - 1, (int)ceilf(xmax - 0.5f));
    r.y0 = max(0, (int)floorf(ymin - 0.5f));
    r.y1 = min(res - 1, (int)ceilf(ymax - 0.5f));
    return r;
}

__device__ __forceinline__ bool inside_strict(const TriRaster &r, float px, float py) {
    if (r.degenerate) return false;
    const float l1 = ((px - r.x[0]) * (r.y[2] - r.y[0]) - (r.x[2] - r.x[0]) * (py - r.y[0])) * r.inv_area;
    const float l2 = ((r.x[1] - r.x[0]) * (py - r.y[0]) - (px - r.x[0]) * (r.y[1] - r.y[0])) * r.inv_area;
    return l1 > EDGE_EPS && l2 > EDGE_EPS && (1.0f - l1 - l2) > EDGE_EPS;
}

__device__ __forceinline__ unsigned long long zkey(float depth, long f) {
    return ((unsigned long long)uv_f2ord(depth) << 32) | (unsigned long long)(0xffffffffu - (unsigned)f);  // ties: lowest id wins
}

// PASS 0: draw the triangles of level `level` (assigned / 6 == level) into their chart's z-buffer.
// PASS 1: a triangle that is not the winner of one of its samples moves one level up (c -> c + 6 -> 12).
template <typename IdxT, int PASS>
__global__ __launch_bounds__(UVB) void uv_zbuffer_kernel(const float *__restrict__ rpos, const IdxT *__restrict__ faces, long nf,
                                                         const float *__restrict__ face_uv, int *__restrict__ assigned, int level, int res,
                                                         unsigned long long *__restrict__ zbuf) {
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) {
        const int idx = assigned[f];
        if (idx / 6 != level) continue;
        const int c = idx % 6;
        int vi[3];
        load_face(faces, f, vi[0], vi[1], vi[2]);
        const int ax = c >> 1;
        const float cen = (rpos[3 * (long)vi[0] + ax] + rpos[3 * (long)vi[1] + ax] + rpos[3 * (long)vi[2] + ax]) * (1.0f / 3.0f);
        const unsigned long long key = zkey((c & 1) ? -cen : cen, f);   // in front = further out along the chart's direction
        const TriRaster r = make_raster(face_uv + 6 * f, res);
        unsigned long long *zb = zbuf + (size_t)c * res * res;
        bool lost = false;
        int nsamples = 0;
        for (int py = r.y0; py <= r.y1 && !(PASS == 1 && lost); ++py)
            for (int px = r.x0; px <= r.x1; ++px) {
                if (!inside_strict(r, px + 0.5f, py + 0.5f)) continue;
                ++nsamples;
                if (PASS == 0) atomicMax(&zb[(size_t)py * res + px], key);
                else if (zb[(size_t)py * res + px] != key) { lost = true; break; }
            }
        if (PASS == 1 && nsamples == 0) {
            // smaller than a pixel: it drew nothing; it is hidden if its centroid lies inside the triangle that owns the
            // pixel and that triangle is in front of it
            const float cx = (r.x[0] + r.x[1] + r.x[2]) * (1.0f / 3.0f), cy = (r.y[0] + r.y[1] + r.y[2]) * (1.0f / 3.0f);
            const int px = min(res - 1, max(0, (int)cx)), py = min(res - 1, max(0, (int)cy));
            const unsigned long long w = zb[(size_t)py * res + px];
            if (w > key) {
                const long wf = (long)(0xffffffffu - (unsigned)(w & 0xffffffffull));
                const TriRaster rw = make_raster(face_uv + 6 * wf, res);
                lost = inside_strict(rw, cx, cy);
            }
        }
        if (PASS == 1 && lost) assigned[f] = level == 0 ? c + 6 : 12;
    }
}

__global__ __launch_bounds__(UVB) void uv_copy_chart_kernel(const int *__restrict__ chart, int *__restrict__ assigned, long nf) {
    for (long f = (long)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (long)gridDim.x * blockDim.x) assigned[f] = chart[f];
}

// ---------------------------------------------------------------------------------------------- placement
__global__ __launch_bounds__(UVB) void uv_slice_stats_kernel(const float *__restrict__ face_uv, const int *__restrict__ assigned, long nf,
                                                             unsigned *__restrict__ st, int *__restrict__ block_cnt) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool rem = false;
    int slice = -1;
    float umin = FLT_MAX, umax = -FLT_MAX, vmin = FLT_MAX, vmax = -FLT_MAX;
    if (f < nf) {
        const int a = assigned[f];
        rem = a >= 12;
        if (a >= 6 && a < 12) {
            const float *t = face_uv + 6 * f;
            slice = a - 6;
            umin = fminf(t[0], fminf(t[2], t[4])); umax = fmaxf(t[0], fmaxf(t[2], t[4]));
            vmin = fminf(t[1], fminf(t[3], t[5])); vmax = fmaxf(t[1], fmaxf(t[3], t[5]));
        }
    }
    // one atomic per wave, slice and bound instead of four per face: every face of a slice hits the same four words
    // (9.6 M faces: 20.7 ms of same-address atomics -> wave reductions first, like the chart bounds above)
    if (__ballot(slice >= 0) != 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const bool mine = slice == c;
            if (__ballot(mine) == 0) continue;  // wave-uniform
            float a0 = mine ? umin : FLT_MAX, a1 = mine ? umax : -FLT_MAX, b0 = mine ? vmin : FLT_MAX, b1 = mine ? vmax : -FLT_MAX;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                a0 = fminf(a0, __shfl_xor(a0, d, 64));
                a1 = fmaxf(a1, __shfl_xor(a1, d, 64));
                b0 = fminf(b0, __shfl_xor(b0, d, 64));
                b1 = fmaxf(b1, __shfl_xor(b1, d, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                // ... and only when the wave's bound improves on what is already there (a plain load: the bounds settle
                // after the first few thousand faces, the remaining same-address atomics would still serialise in L2)
                const unsigned o0 = uv_f2ord(a0), o1 = uv_f2ord(a1), p0 = uv_f2ord(b0), p1 = uv_f2ord(b1);
                const volatile unsigned *sv = st;
                if (o0 < sv[ST_SL_UMIN + c]) atomicMin(&st[ST_SL_UMIN + c], o0);
                if (o1 > sv[ST_SL_UMAX + c]) atomicMax(&st[ST_SL_UMAX + c], o1);
                if (p0 < sv[ST_SL_VMIN + c]) atomicMin(&st[ST_SL_VMIN + c], p0);
                if (p1 > sv[ST_SL_VMAX + c]) atomicMax(&st[ST_SL_VMAX + c], p1);
            }
        }
    }
    __shared__ int wc[UVB / 64];
    const unsigned long long bal = __ballot(rem);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < UVB / 64; ++w) t += wc[w];
        block_cnt[blockIdx.x] = t;
    }
}

// exclusive scan of block_cnt in place (one workgroup), total -> st[ST_REMAINING]
__global__ __launch_bounds__(1024) void uv_scan_blocks_kernel(int *__restrict__ block_cnt, int nb, unsigned *__restrict__ st) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_cnt[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if ((threadIdx.x & 63) >= d) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int carry = carry_s;
        if (i < nb) block_cnt[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) st[ST_REMAINING] = (unsigned)carry_s;
}

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

__global__ __launch_bounds__(UVB) void uv_place_kernel(const float *__restrict__ face_uv, const int *__restrict__ assigned, long nf, double pad,
                                                       const unsigned *__restrict__ st, const int *__restrict__ block_off,
                                                       float *__restrict__ out) {
    const long f = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = f < nf ? assigned[f] : 0;
    const bool rem = f < nf && a >= 12;
    // rank of this face among the remaining ones, in face order
    __shared__ int wc[UVB / 64];
    const unsigned long long bal = __ballot(rem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    int rank = block_off[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += wc[w];
    if (f >= nf) return;
    float uc[3], vc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { uc[k] = face_uv[6 * f + 2 * k]; vc[k] = face_uv[6 * f + 2 * k + 1]; }
    if (a >= 6 && a < 12) {  // _handle_slice_uvs: fill the patch, at most 2x magnified
        const float ulo = uv_ord2f(st[ST_SL_UMIN + a - 6]), uhi = uv_ord2f(st[ST_SL_UMAX + a - 6]);
        const float vlo = uv_ord2f(st[ST_SL_VMIN + a - 6]), vhi = uv_ord2f(st[ST_SL_VMAX + a - 6]);
        const float us = fmaxf(__fsub_rn(uhi, ulo), 0.5f), vs = fmaxf(__fsub_rn(vhi, vlo), 0.5f);
#pragma unroll
        for (int k = 0; k < 3; ++k) { uc[k] = __fdiv_rn(__fsub_rn(uc[k], ulo), us); vc[k] = __fdiv_rn(__fsub_rn(vc[k], vlo), vs); }
    }
    const float m1 = (float)(1.0 - 2.0 * pad), a1 = (float)pad;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uc[k] = clamp01(__fadd_rn(__fmul_rn(uc[k], m1), a1));
        vc[k] = clamp01(__fadd_rn(__fmul_rn(vc[k], m1), a1));
    }
    if (rem) {  // _handle_remaining_uvs: every triangle in its own cell of an nw x nh grid
        const double left = (double)st[ST_REMAINING];
        const int nw = (int)ceil(0.5 * sqrt(left / (0.5 * (1.0 / 3.0))));
        const int nh = (int)ceil(left / (double)nw);
        const double w = 1.0 / nw, h = 1.0 / nh;
        const float lim = (float)(fmin(w, h) * 1.5);
        const float ulo = fminf(uc[0], fminf(uc[1], uc[2])), uhi = fmaxf(uc[0], fmaxf(uc[1], uc[2]));
        const float vlo = fminf(vc[0], fminf(vc[1], vc[2])), vhi = fmaxf(vc[0], fmaxf(vc[1], vc[2]));
        const float us = fmaxf(__fsub_rn(uhi, ulo), lim), vs = fmaxf(__fsub_rn(vhi, vlo), lim);
        const float mu = (float)(1.0 - pad * nw * 0.5), au = (float)(pad * nw * 0.25);
        const float mv = (float)(1.0 - pad * nh * 0.5), av = (float)(pad * nh * 0.25);
        const float wf = (float)w, hf = (float)h;
        const float xo = __fmul_rn((float)(rank % nw), wf), yo = __fmul_rn((float)(rank / nw), hf);
        const float m2 = (float)(1.0 - 2.0 * pad * 0.5), a2 = (float)(pad * 0.5);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float u = clamp01(__fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(uc[k], ulo), us), mu), au));
            float v = clamp01(__fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(vc[k], vlo), vs), mv), av));
            u = __fadd_rn(__fmul_rn(u, wf), xo);
            v = __fadd_rn(__fmul_rn(v, hf), yo);
            uc[k] = clamp01(__fadd_rn(__fmul_rn(u, m2), a2));
            vc[k] = clamp01(__fadd_rn(__fmul_rn(v, m2), a2));
        }
    }
    // _find_slice_offset_and_scale
    const int lvl = a / 6, six = a % 6;
    const int gx = six % 3, gy = six / 3;
    float ox, oy, dx, dy;
    if (lvl == 0) {
        ox = (float)((1.0 / 3.0) * gx);
        oy = (float)((1.0 / 3.0) * gy);
        dx = dy = 3.f;
    } else {
        ox = (float)((1.0 / 6.0) * gx + (double)min(lvl - 1, 1) * 0.5);
        oy = (float)((1.0 / 6.0) * gy + (1.0 / 3.0) * 2.0);
        dx = lvl >= 2 ? 2.f : 6.f;
        dy = lvl >= 2 ? 3.f : 6.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[6 * f + 2 * k] = __fadd_rn(__fdiv_rn(uc[k], dx), ox);
        out[6 * f + 2 * k + 1] = __fadd_rn(__fdiv_rn(vc[k], dy), oy);
    }
}

inline int uv_grid(long n) { return (int)std::max<long>(1, std::min<long>((n + UVB - 1) / UVB, (long)num_cus() * 32)); }

}  // namespace

}  // namespace sculpt

using namespace sculpt;

extern "C" {

size_t sculpt_uv_stats_words(void) { return (size_t)ST_WORDS; }

int sculpt_uv_moments(const float *v_pos, size_t nv, double *sums9, sculpt_stream_t stream) {
    SC_REQUIRE(v_pos && sums9 && nv >= 1, "uv_moments: bad argument");
    hipStream_t st = as_stream(stream);
    SC_HIP(hipMemsetAsync(sums9, 0, 9 * sizeof(double), st));
    hipLaunchKernelGGL(uv_moments_kernel, dim3(uv_grid((long)nv)), dim3(UVB), 0, st, v_pos, (long)nv, sums9);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_uv_box_project(const float *v_pos, const float *v_nrm, size_t nv, const void *faces, int faces_i64, size_t nf,
                          const float *rot9_host, float *rot_pos, float *rot_nrm, float *face_uv, int *chart, unsigned *stats,
                          sculpt_stream_t stream) {
    SC_REQUIRE(v_pos && v_nrm && faces && rot9_host && rot_pos && rot_nrm && face_uv && chart && stats, "uv_box_project: null argument");
    SC_REQUIRE(nv >= 1 && nf >= 1, "uv_box_project: empty mesh");
    hipStream_t st = as_stream(stream);
    Rot3 R;
    for (int i = 0; i < 9; ++i) R.m[i] = rot9_host[i];
    hipLaunchKernelGGL(uv_stats_init_kernel, dim3(1), dim3(64), 0, st, stats);
    hipLaunchKernelGGL(uv_rotate_mesh_kernel, dim3(uv_grid((long)nv)), dim3(UVB), 0, st, v_pos, v_nrm, (long)nv, R, rot_pos, rot_nrm, stats);
    if (faces_i64)
        hipLaunchKernelGGL(uv_box_project_kernel<long long>, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, rot_pos, rot_nrm,
                           reinterpret_cast<const long long *>(faces), (long)nf, face_uv, chart, stats);
    else
        hipLaunchKernelGGL(uv_box_project_kernel<int>, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, rot_pos, rot_nrm,
                           reinterpret_cast<const int *>(faces), (long)nf, face_uv, chart, stats);
    hipLaunchKernelGGL(uv_box_finish_kernel, dim3(uv_grid((long)nf * 6)), dim3(UVB), 0, st, face_uv, (long)nf, stats);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_uv_chart_tangents(const float *rot_pos, const float *rot_nrm, size_t nv, const void *faces, int faces_i64, size_t nf,
                             const float *face_uv, const int *chart, float *vertex_tangents4, double *sums42, sculpt_stream_t stream) {
    SC_REQUIRE(rot_pos && rot_nrm && faces && face_uv && chart && vertex_tangents4 && sums42, "uv_chart_tangents: null argument");
    hipStream_t st = as_stream(stream);
    SC_HIP(hipMemsetAsync(vertex_tangents4, 0, nv * 4 * sizeof(float), st));
    SC_HIP(hipMemsetAsync(sums42, 0, 42 * sizeof(double), st));
    if (faces_i64) {
        const long long *F = reinterpret_cast<const long long *>(faces);
        hipLaunchKernelGGL(uv_face_tangent_kernel<long long>, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, rot_pos, F, (long)nf, face_uv, vertex_tangents4);
        hipLaunchKernelGGL(uv_vertex_tangent_kernel, dim3(uv_grid((long)nv)), dim3(UVB), 0, st, vertex_tangents4, rot_nrm, (long)nv);
        hipLaunchKernelGGL(uv_chart_sums_kernel<long long>, dim3(std::min(uv_grid((long)nf), 1024)), dim3(UVB), 0, st, rot_pos, rot_nrm, F, (long)nf,
                           chart, vertex_tangents4, sums42);
    } else {
        const int *F = reinterpret_cast<const int *>(faces);
        hipLaunchKernelGGL(uv_face_tangent_kernel<int>, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, rot_pos, F, (long)nf, face_uv, vertex_tangents4);
        hipLaunchKernelGGL(uv_vertex_tangent_kernel, dim3(uv_grid((long)nv)), dim3(UVB), 0, st, vertex_tangents4, rot_nrm, (long)nv);
        hipLaunchKernelGGL(uv_chart_sums_kernel<int>, dim3(std::min(uv_grid((long)nf), 1024)), dim3(UVB), 0, st, rot_pos, rot_nrm, F, (long)nf, chart,
                           vertex_tangents4, sums42);
    }
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_uv_rotate_charts(float *face_uv, const int *chart, size_t nf, const float *cos6_host, const float *sin6_host, unsigned *stats,
                            sculpt_stream_t stream) {
    SC_REQUIRE(face_uv && chart && cos6_host && sin6_host && stats && nf >= 1, "uv_rotate_charts: bad argument");
    hipStream_t st = as_stream(stream);
    Rot2x6 R;
    for (int c = 0; c < 6; ++c) { R.c[c] = cos6_host[c]; R.s[c] = sin6_host[c]; }
    hipLaunchKernelGGL(uv_rotate_chart_kernel, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, face_uv, chart, (long)nf, R, stats);
    hipLaunchKernelGGL(uv_rescale_chart_kernel, dim3(uv_grid((long)nf)), dim3(UVB), 0, st, face_uv, chart, (long)nf, stats);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_uv_assign_atlas(const float *rot_pos, const void *faces, int faces_i64, size_t nf, const float *face_uv, const int *chart,
                           int res, unsigned long long *zbuf, int *assigned, sculpt_stream_t stream) {
    SC_REQUIRE(rot_pos && faces && face_uv && chart && zbuf && assigned && nf >= 1, "uv_assign_atlas: bad argument");
    SC_REQUIRE(res >= 16 && res <= 8192, "uv_assign_atlas: z-buffer resolution %d out of range [16, 8192]", res);
    hipStream_t st = as_stream(stream);
    const dim3 grid(uv_grid((long)nf)), block(UVB);
    hipLaunchKernelGGL(uv_copy_chart_kernel, grid, block, 0, st, chart, assigned, (long)nf);
    const size_t zbytes = (size_t)6 * res * res * sizeof(unsigned long long);
    for (int level = 0; level < 2; ++level) {
        SC_HIP(hipMemsetAsync(zbuf, 0, zbytes, st));
        if (faces_i64) {
            const long long *F = reinterpret_cast<const long long *>(faces);
            hipLaunchKernelGGL((uv_zbuffer_kernel<long long, 0>), grid, block, 0, st, rot_pos, F, (long)nf, face_uv, assigned, level, res, zbuf);
            hipLaunchKernelGGL((uv_zbuffer_kernel<long long, 1>), grid, block, 0, st, rot_pos, F, (long)nf, face_uv, assigned, level, res, zbuf);
        } else {
            const int *F = reinterpret_cast<const int *>(faces);
            hipLaunchKernelGGL((uv_zbuffer_kernel<int, 0>), grid, block, 0, st, rot_pos, F, (long)nf, face_uv, assigned, level, res, zbuf);
            hipLaunchKernelGGL((uv_zbuffer_kernel<int, 1>), grid, block, 0, st, rot_pos, F, (long)nf, face_uv, assigned, level, res, zbuf);
        }
    }
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_uv_place(const float *face_uv, const int *assigned, size_t nf, double island_padding, unsigned *stats, int *block_scratch,
                    float *out_uv, sculpt_stream_t stream) {
    SC_REQUIRE(face_uv && assigned && stats && block_scratch && out_uv && nf >= 1, "uv_place: bad argument");
    hipStream_t st = as_stream(stream);
    const int nb = (int)((nf + UVB - 1) / UVB);
    hipLaunchKernelGGL(uv_slice_stats_kernel, dim3(nb), dim3(UVB), 0, st, face_uv, assigned, (long)nf, stats, block_scratch);
    hipLaunchKernelGGL(uv_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, block_scratch, nb, stats);
    hipLaunchKernelGGL(uv_place_kernel, dim3(nb), dim3(UVB), 0, st, face_uv, assigned, (long)nf, island_padding, stats, block_scratch, out_uv);
    SC_LAUNCH_CHECK();
    return 0;
}

// The DLL's own entry point (HOST pointers; name and signature as unwrap.py:147-154 declares them for uv_unwrapper.dll), so
// that the reference's Unwrapper class runs unchanged on Linux with ctypes.CDLL pointed at this library.
void assign_faces_uv_to_atlas_index(const float *vertices, size_t nv, const long long *indices, size_t nf, const float *face_uv,
                                    const long long *face_index, long long *out) {
    if (nf == 0) return;
    const int res = 1024;
    float *d_v = nullptr, *d_uv = nullptr;
    long long *d_f = nullptr;
    int *d_chart = nullptr, *d_assigned = nullptr;
    unsigned long long *d_z = nullptr;
    std::vector<int> chart(nf), assigned(nf);
    for (size_t i = 0; i < nf; ++i) chart[i] = (int)face_index[i];
    bool ok = hipMalloc(&d_v, sizeof(float) * 3 * (nv ? nv : 1)) == hipSuccess && hipMalloc(&d_f, sizeof(long long) * 3 * nf) == hipSuccess &&
              hipMalloc(&d_uv, sizeof(float) * 6 * nf) == hipSuccess && hipMalloc(&d_chart, sizeof(int) * nf) == hipSuccess &&
              hipMalloc(&d_assigned, sizeof(int) * nf) == hipSuccess &&
              hipMalloc(&d_z, sizeof(unsigned long long) * 6 * (size_t)res * res) == hipSuccess;
    if (ok && nv) ok = hipMemcpy(d_v, vertices, sizeof(float) * 3 * nv, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = hipMemcpy(d_f, indices, sizeof(long long) * 3 * nf, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(d_uv, face_uv, sizeof(float) * 6 * nf, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(d_chart, chart.data(), sizeof(int) * nf, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = sculpt_uv_assign_atlas(d_v, d_f, 1, nf, d_uv, d_chart, res, d_z, d_assigned, nullptr) == 0;
    if (ok) ok = hipMemcpy(assigned.data(), d_assigned, sizeof(int) * nf, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        set_error("assign_faces_uv_to_atlas_index: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
        for (size_t i = 0; i < nf; ++i) out[i] = face_index[i];   // the function returns void: leave every face in its chart
    } else {
        for (size_t i = 0; i < nf; ++i) out[i] = assigned[i];
    }
    (void)hipFree(d_v); (void)hipFree(d_f); (void)hipFree(d_uv); (void)hipFree(d_chart); (void)hipFree(d_assigned); (void)hipFree(d_z);
}

}  // extern "C"
