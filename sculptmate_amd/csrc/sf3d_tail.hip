// StableFast geometry tail on the GPU, three small HBM-bound pieces that the
// reference runs as chains of torch ops around its bake step:
//   sculpt_dilate_fill        dilate_fill            StableFast/sf3d/models/utils.py:96-133
//   sculpt_vertex_normals     Mesh._compute_vertex_normal   StableFast/sf3d/models/mesh.py:66-92
//   sculpt_vertex_tangents    Mesh._compute_vertex_tangent  StableFast/sf3d/models/mesh.py:94-139
#include "common.h"

namespace sculpt {

// ---- dilate_fill, one iteration = two passes over the image (planar [3][H][W], mask [H][W]) -------------
// pass A: newMask = maxpool3x3(oldMask); for interior centres p (unfold has no padding):
//         mean(p) = sum_{3x3} oldImg / max(sum_{3x3} oldMask, 1)
__global__ __launch_bounds__(256) void dilate_a_kernel(const float *__restrict__ img, const float *__restrict__ mask, int H,
                                                       int W, float *__restrict__ new_mask, float *__restrict__ mean) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        float mx = -INFINITY, ms = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const long j = (long)yy * W + xx;
                const float m = mask[j];
                mx = fmaxf(mx, m);
                ms += m;
                s0 += img[j]; s1 += img[n + j]; s2 += img[2 * n + j];
            }
        new_mask[i] = mx;
        const bool interior = x >= 1 && x < W - 1 && y >= 1 && y < H - 1;
        const float d = fmaxf(ms, 1.0f);
        mean[i] = interior ? s0 / d : 0.f;
        mean[n + i] = interior ? s1 / d : 0.f;
        mean[2 * n + i] = interior ? s2 / d : 0.f;
    }
}
// pass B: newImg(q) = newMask(q) * sum_{interior p in N(q)} mean(p) / max(sum_{N(q)} newMask, 1);
//         out = lerp(oldImg, newImg, newMask - oldMask)
__global__ __launch_bounds__(256) void dilate_b_kernel(const float *__restrict__ img, const float *__restrict__ mask,
                                                       const float *__restrict__ new_mask, const float *__restrict__ mean,
                                                       int H, int W, float *__restrict__ out) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        float mc = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const long j = (long)yy * W + xx;
                mc += new_mask[j];
                s0 += mean[j]; s1 += mean[n + j]; s2 += mean[2 * n + j];  // mean is 0 at non-interior centres
            }
        const float nm = new_mask[i], w = nm - mask[i], d = fmaxf(mc, 1.0f);
        const float a0 = img[i], a1 = img[n + i], a2 = img[2 * n + i];
        out[i] = a0 + w * (s0 * nm / d - a0);
        out[n + i] = a1 + w * (s1 * nm / d - a1);
        out[2 * n + i] = a2 + w * (s2 * nm / d - a2);
    }
}

// ---- vertex normals / tangents ---------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(256) void face_normal_splat_kernel(const float *__restrict__ v, const IdxT *__restrict__ f,
                                                                long nf, float *__restrict__ acc) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nf) return;
    const long i0 = f[3 * t], i1 = f[3 * t + 1], i2 = f[3 * t + 2];
    const float ax = v[3 * i1] - v[3 * i0], ay = v[3 * i1 + 1] - v[3 * i0 + 1], az = v[3 * i1 + 2] - v[3 * i0 + 2];
    const float bx = v[3 * i2] - v[3 * i0], by = v[3 * i2 + 1] - v[3 * i0 + 1], bz = v[3 * i2 + 2] - v[3 * i0 + 2];
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const long ids[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        atomicAdd(&acc[3 * ids[k]], nx); atomicAdd(&acc[3 * ids[k] + 1], ny); atomicAdd(&acc[3 * ids[k] + 2], nz);
    }
}
__global__ __launch_bounds__(256) void normal_finish_kernel(float *__restrict__ n, long nv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    float x = n[3 * i], y = n[3 * i + 1], z = n[3 * i + 2];
    if (!(x * x + y * y + z * z > 1e-20f)) { x = 0.f; y = 0.f; z = 1.f; }
    const float l = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);  // F.normalize eps
    n[3 * i] = x / l; n[3 * i + 1] = y / l; n[3 * i + 2] = z / l;
}
template <typename IdxT>
__global__ __launch_bounds__(256) void face_tangent_splat_kernel(const float *__restrict__ v, const float *__restrict__ uv,
                                                                 const IdxT *__restrict__ f, long nf,
                                                                 float *__restrict__ tan, float *__restrict__ cnt) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nf) return;
    const long i0 = f[3 * t], i1 = f[3 * t + 1], i2 = f[3 * t + 2];
    const float du1x = uv[2 * i1] - uv[2 * i0], du1y = uv[2 * i1 + 1] - uv[2 * i0 + 1];
    const float du2x = uv[2 * i2] - uv[2 * i0], du2y = uv[2 * i2 + 1] - uv[2 * i0 + 1];
    float denom = du1x * du2y - du1y * du2x;
    denom = fmaxf(denom, 1e-6f);  // denom.clip(1e-6): clip(min) only
    float tg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float dp1 = v[3 * i1 + k] - v[3 * i0 + k], dp2 = v[3 * i2 + k] - v[3 * i0 + k];
        tg[k] = (dp1 * du2y - dp2 * du1y) / denom;
    }
    const long ids[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        atomicAdd(&tan[3 * ids[k]], tg[0]); atomicAdd(&tan[3 * ids[k] + 1], tg[1]); atomicAdd(&tan[3 * ids[k] + 2], tg[2]);
        atomicAdd(&cnt[ids[k]], 1.0f);
    }
}
__global__ __launch_bounds__(256) void tangent_finish_kernel(float *__restrict__ tan, const float *__restrict__ cnt,
                                                             const float *__restrict__ nrm, long nv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    const float c = cnt[i];
    float x = tan[3 * i] / c, y = tan[3 * i + 1] / c, z = tan[3 * i + 2] / c;  // tangents / tansum
    float l = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    x /= l; y /= l; z /= l;
    const float nx = nrm[3 * i], ny = nrm[3 * i + 1], nz = nrm[3 * i + 2];
    const float d = x * nx + y * ny + z * nz;
    x -= d * nx; y -= d * ny; z -= d * nz;
    l = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    tan[3 * i] = x / l; tan[3 * i + 1] = y / l; tan[3 * i + 2] = z / l;
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

// img [3][H][W], mask [H][W] (0/1 floats); scratch = 8*H*W floats; result in out [3][H][W]
int sculpt_dilate_fill(const float *img, const float *mask, int H, int W, int iterations, float *scratch, float *out,
                       sculpt_stream_t stream) {
    SC_REQUIRE(img && mask && scratch && out && H >= 3 && W >= 3 && iterations >= 0, "dilate_fill: bad argument");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)H * W;
    float *mA = scratch, *mB = scratch + n, *mean = scratch + 2 * n, *imgB = scratch + 5 * n;  // 1+1+3+3 planes
    SC_HIP(hipMemcpyAsync(mA, mask, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    SC_HIP(hipMemcpyAsync(out, img, 3 * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    const int grid = (int)std::min<long>((n + 255) / 256, 4096);
    float *cur = out, *nxt = imgB, *mcur = mA, *mnxt = mB;
    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(dilate_a_kernel, dim3(grid), dim3(256), 0, st, cur, mcur, H, W, mnxt, mean);
        hipLaunchKernelGGL(dilate_b_kernel, dim3(grid), dim3(256), 0, st, cur, mcur, mnxt, mean, H, W, nxt);
        SC_LAUNCH_CHECK();
        float *t = cur; cur = nxt; nxt = t;
        t = mcur; mcur = mnxt; mnxt = t;
    }
    if (cur != out) SC_HIP(hipMemcpyAsync(out, cur, 3 * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

int sculpt_vertex_normals(const float *v_pos, size_t nv, const void *faces, int faces_i64, size_t nf, float *out,
                          sculpt_stream_t stream) {
    SC_REQUIRE(v_pos && faces && out, "vertex_normals: null argument");
    hipStream_t st = as_stream(stream);
    SC_HIP(hipMemsetAsync(out, 0, 3 * nv * sizeof(float), st));
    if (nf) {
        if (faces_i64)
            hipLaunchKernelGGL(face_normal_splat_kernel<long long>, dim3(cdiv((long)nf, 256)), dim3(256), 0, st, v_pos,
                               reinterpret_cast<const long long *>(faces), (long)nf, out);
        else
            hipLaunchKernelGGL(face_normal_splat_kernel<int>, dim3(cdiv((long)nf, 256)), dim3(256), 0, st, v_pos,
                               reinterpret_cast<const int *>(faces), (long)nf, out);
        SC_LAUNCH_CHECK();
    }
    if (nv) hipLaunchKernelGGL(normal_finish_kernel, dim3(cdiv((long)nv, 256)), dim3(256), 0, st, out, (long)nv);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_vertex_tangents(const float *v_pos, const float *v_tex, const float *v_nrm, size_t nv, const void *faces,
                           int faces_i64, size_t nf, float *count_scratch, float *out, sculpt_stream_t stream) {
    SC_REQUIRE(v_pos && v_tex && v_nrm && faces && out && count_scratch, "vertex_tangents: null argument");
    hipStream_t st = as_stream(stream);
    SC_HIP(hipMemsetAsync(out, 0, 3 * nv * sizeof(float), st));
    SC_HIP(hipMemsetAsync(count_scratch, 0, nv * sizeof(float), st));
    if (nf) {
        if (faces_i64)
            hipLaunchKernelGGL(face_tangent_splat_kernel<long long>, dim3(cdiv((long)nf, 256)), dim3(256), 0, st, v_pos, v_tex,
                               reinterpret_cast<const long long *>(faces), (long)nf, out, count_scratch);
        else
            hipLaunchKernelGGL(face_tangent_splat_kernel<int>, dim3(cdiv((long)nf, 256)), dim3(256), 0, st, v_pos, v_tex,
                               reinterpret_cast<const int *>(faces), (long)nf, out, count_scratch);
        SC_LAUNCH_CHECK();
    }
    if (nv) hipLaunchKernelGGL(tangent_finish_kernel, dim3(cdiv((long)nv, 256)), dim3(256), 0, st, out, count_scratch, v_nrm, (long)nv);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
