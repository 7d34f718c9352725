// UV-space texture baker for gfx950: rasterise a UV-unwrapped mesh into a (barycentrics, triangle id)
// map and interpolate per-vertex attributes through it.
//
// Replaces the two exports of StableFast/sf3d/texture_baker/texture_baker.dll (Windows-only binary,
// no source in the reference), whose C ABI is declared by the caller at
//   StableFast/sf3d/texture_baker/baker.py:31-57   rasterize_cpu(uv, nv, idx, nf, res, out[res*res*4])
//   StableFast/sf3d/texture_baker/baker.py:91-118  interpolate_cpu(attr, nv, idx, nf, rast, res, out[res*res*3])
// and whose semantics are restated in StableFast/sf3d/texture_baker/common.py:104-142, 214-229:
// pixel (x, y) samples the point (x/W, 1 - y/H); the pixel gets the barycentrics (u, v, w) and the index
// of a triangle with u, v, w >= 0, or (0, 0, 0, -1) when no triangle covers it.
// "First hit wins" there depends on the BVH traversal order of a binary we cannot see; here the winner
// is the LOWEST triangle index (identical for non-overlapping charts; parity unpinned where UV triangles
// overlap -- DESIGN.md section 7).
// Kernels: (1) clear the winner map; (2) one thread per triangle scans its pixel bounding box and
// atomicMin's its index into every covered pixel (HBM traffic ~ 4 B per covered pixel);
// (3) one thread per pixel recomputes the winner's barycentrics (same fp32 expression) and writes 16 B.
#include <limits.h>

#include "common.h"

#pragma clang fp contract(off)  // the oracle / CPU code rounds every fp32 operation separately

namespace sculpt {

struct Bary { float u, v, w; };

__device__ __forceinline__ Bary barycentric(float px, float py, float ax, float ay, float bx, float by, float cx, float cy) {
    // common.py:104-121, in fp32
    const float e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay, qx = px - ax, qy = py - ay;
    const float d00 = e1x * e1x + e1y * e1y;
    const float d01 = e1x * e2x + e1y * e2y;
    const float d11 = e2x * e2x + e2y * e2y;
    const float d20 = qx * e1x + qy * e1y;
    const float d21 = qx * e2x + qy * e2y;
    const float denom = d00 * d11 - d01 * d01;
    Bary b;
    b.v = (d11 * d20 - d01 * d21) / denom;
    b.w = (d00 * d21 - d01 * d20) / denom;
    b.u = 1.0f - b.v - b.w;
    return b;
}

__global__ __launch_bounds__(256) void bake_clear_kernel(int *__restrict__ best, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) best[i] = INT_MAX;
}

__global__ __launch_bounds__(256) void bake_cover_kernel(const float *__restrict__ uv, const int *__restrict__ idx,
                                                         long nf, int res, int *__restrict__ best) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nf) return;
    const int i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
    const float ax = uv[2 * i0], ay = uv[2 * i0 + 1], bx = uv[2 * i1], by = uv[2 * i1 + 1], cx = uv[2 * i2], cy = uv[2 * i2 + 1];
    const float mnx = fminf(ax, fminf(bx, cx)), mxx = fmaxf(ax, fmaxf(bx, cx));
    const float mny = fminf(ay, fminf(by, cy)), mxy = fmaxf(ay, fmaxf(by, cy));
    // pixel x samples x/res; pixel y samples 1 - y/res  ->  conservative integer ranges (+-1 pixel)
    const float R = (float)res;
    int x0 = (int)floorf(mnx * R) - 1, x1 = (int)ceilf(mxx * R) + 1;
    int y0 = (int)floorf((1.0f - mxy) * R) - 1, y1 = (int)ceilf((1.0f - mny) * R) + 1;
    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, res - 1); y1 = min(y1, res - 1);
    for (int y = y0; y <= y1; ++y) {
        const float py = 1.0f - (float)y / R;
        for (int x = x0; x <= x1; ++x) {
            const float px = (float)x / R;
            const Bary b = barycentric(px, py, ax, ay, bx, by, cx, cy);
            if (b.u >= 0.f && b.v >= 0.f && b.w >= 0.f) atomicMin(&best[(long)y * res + x], (int)t);
        }
    }
}

__global__ __launch_bounds__(256) void bake_resolve_kernel(const float *__restrict__ uv, const int *__restrict__ idx,
                                                           int res, const int *__restrict__ best, float4 *__restrict__ out) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)res * res) return;
    const int t = best[p];
    if (t == INT_MAX) { out[p] = make_float4(0.f, 0.f, 0.f, -1.f); return; }
    const int x = (int)(p % res), y = (int)(p / res);
    const float R = (float)res;
    const int i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
    const Bary b = barycentric((float)x / R, 1.0f - (float)y / R, uv[2 * i0], uv[2 * i0 + 1], uv[2 * i1], uv[2 * i1 + 1],
                               uv[2 * i2], uv[2 * i2 + 1]);
    out[p] = make_float4(b.u, b.v, b.w, (float)t);
}

// out[p] = attr[i0]*u + attr[i1]*v + attr[i2]*w  (common.py:214-229), zeros where the pixel is empty
__global__ __launch_bounds__(256) void bake_interpolate_kernel(const float *__restrict__ attr, const int *__restrict__ idx,
                                                               const float4 *__restrict__ rast, long npix,
                                                               float *__restrict__ out) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const float4 r = rast[p];
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (!(r.w < 0.f)) {
        const int t = (int)r.w;
        const int i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
        o0 = attr[3 * i0] * r.x + attr[3 * i1] * r.y + attr[3 * i2] * r.z;
        o1 = attr[3 * i0 + 1] * r.x + attr[3 * i1 + 1] * r.y + attr[3 * i2 + 1] * r.z;
        o2 = attr[3 * i0 + 2] * r.x + attr[3 * i1 + 2] * r.y + attr[3 * i2 + 2] * r.z;
    }
    out[3 * p] = o0; out[3 * p + 1] = o1; out[3 * p + 2] = o2;
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

size_t sculpt_bake_workspace_bytes(int res) { return (size_t)res * res * sizeof(int); }

int sculpt_bake_rasterize(const float *uv, size_t nv, const int *idx, size_t nf, int res, void *workspace, float *out,
                          sculpt_stream_t stream) {
    hipStream_t st = as_stream(stream);
    SC_REQUIRE(res >= 1 && out && workspace, "bake_rasterize: bad argument");
    SC_REQUIRE(nf == 0 || (uv && idx && nv > 0), "bake_rasterize: null mesh");
    int *best = reinterpret_cast<int *>(workspace);
    const long npix = (long)res * res;
    hipLaunchKernelGGL(bake_clear_kernel, dim3((int)std::min<long>((npix + 255) / 256, 2048)), dim3(256), 0, st, best, npix);
    SC_LAUNCH_CHECK();
    if (nf > 0) {
        hipLaunchKernelGGL(bake_cover_kernel, dim3(cdiv((long)nf, 256)), dim3(256), 0, st, uv, idx, (long)nf, res, best);
        SC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bake_resolve_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, uv, idx, res, best,
                       reinterpret_cast<float4 *>(out));
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_bake_interpolate(const float *attr, size_t nv, const int *idx, size_t nf, const float *rast, int res,
                            float *out, sculpt_stream_t stream) {
    SC_REQUIRE(res >= 1 && rast && out, "bake_interpolate: bad argument");
    SC_REQUIRE(attr && idx, "bake_interpolate: null mesh");
    (void)nv; (void)nf;
    const long npix = (long)res * res;
    hipLaunchKernelGGL(bake_interpolate_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, as_stream(stream), attr, idx,
                       reinterpret_cast<const float4 *>(rast), npix, out);
    SC_LAUNCH_CHECK();
    return 0;
}

// ---- drop-in replacements of texture_baker.dll's exports (HOST pointers, same names and signatures as
// declared at baker.py:34-41 and :94-101): a Linux build of the add-on can ctypes.CDLL this library.
void rasterize_cpu(const float *uv, size_t nv, const int *idx, size_t nf, long long res, float *out) {
    float *d_uv = nullptr, *d_out = nullptr;
    int *d_idx = nullptr;
    void *d_ws = nullptr;
    const size_t npix = (size_t)res * res;
    bool ok = hipMalloc(&d_uv, sizeof(float) * 2 * (nv ? nv : 1)) == hipSuccess &&
              hipMalloc(&d_idx, sizeof(int) * 3 * (nf ? nf : 1)) == hipSuccess &&
              hipMalloc(&d_out, sizeof(float) * 4 * npix) == hipSuccess && hipMalloc(&d_ws, sizeof(int) * npix) == hipSuccess;
    if (ok && nv) ok = hipMemcpy(d_uv, uv, sizeof(float) * 2 * nv, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && nf) ok = hipMemcpy(d_idx, idx, sizeof(int) * 3 * nf, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = sculpt_bake_rasterize(d_uv, nv, d_idx, nf, (int)res, d_ws, d_out, nullptr) == 0;
    if (ok) ok = hipMemcpy(out, d_out, sizeof(float) * 4 * npix, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) set_error("rasterize_cpu: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_uv); (void)hipFree(d_idx); (void)hipFree(d_out); (void)hipFree(d_ws);
}

void interpolate_cpu(const float *attr, size_t nv, const int *idx, size_t nf, const float *rast, long long res, float *out) {
    float *d_attr = nullptr, *d_rast = nullptr, *d_out = nullptr;
    int *d_idx = nullptr;
    const size_t npix = (size_t)res * res;
    bool ok = hipMalloc(&d_attr, sizeof(float) * 3 * (nv ? nv : 1)) == hipSuccess &&
              hipMalloc(&d_idx, sizeof(int) * 3 * (nf ? nf : 1)) == hipSuccess &&
              hipMalloc(&d_rast, sizeof(float) * 4 * npix) == hipSuccess && hipMalloc(&d_out, sizeof(float) * 3 * npix) == hipSuccess;
    if (ok && nv) ok = hipMemcpy(d_attr, attr, sizeof(float) * 3 * nv, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && nf) ok = hipMemcpy(d_idx, idx, sizeof(int) * 3 * nf, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = hipMemcpy(d_rast, rast, sizeof(float) * 4 * npix, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) ok = sculpt_bake_interpolate(d_attr, nv, d_idx, nf, d_rast, (int)res, d_out, nullptr) == 0;
    if (ok) ok = hipMemcpy(out, d_out, sizeof(float) * 3 * npix, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) set_error("interpolate_cpu: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_attr); (void)hipFree(d_idx); (void)hipFree(d_rast); (void)hipFree(d_out);
}

}  // extern "C"
