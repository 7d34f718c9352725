// Kernels for StableFast-3D's two estimators on gfx950 (BASELINE config 4; SURVEY.md 8b "SF3D boundary": run_image
// returns roughness / metallic, which come out of the image estimator).
//
// Replaces (reference file:line):
//   ClipBasedHeadEstimator.forward      StableFast/sf3d/models/image_estimator/clip_based_estimator.py:88-105
//       cond_image = rgb_cond * mask_cond (sf3d/system.py:326-329) resized to 224 x 224 with
//       F.interpolate(bilinear, align_corners=False) -- one gather kernel, the product is formed per tap.  The
//       CLIP tower itself runs on the ViT kernels (norms.hip patchify / assemble / LayerNorm, gemm.hip, attention.hip).
//   MultiHeadEstimator.forward          StableFast/sf3d/models/global_estimator/multi_head_estimator.py:38-55, 86-104
//       Conv2d(3x3, stride 2, padding 0) over the channel-concatenated triplane as im2col + GEMM (ReLU epilogue):
//       the backbone leaves the triplane as tokens [plane][pixel][F], so the im2col gathers channel c = plane*F + f
//       from group `plane` -- no [B, 3F, H, W] tensor is ever formed; then max / mean over the pixels per channel.
//
// All HBM-bound gathers and reductions: 16-byte accesses along the channel axis, no LDS needed.
#include <float.h>
#include <math.h>

#include "common.h"

namespace sculpt {

__device__ __forceinline__ void est_bilin_src(int dst, int in_size, int out_size, int &i0, int &i1, float &l1) {
    // torch area_pixel_compute_source_index, align_corners=False
    const float scale = (float)in_size / (float)out_size;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// in [Hin][Win][C] fp32 (optionally multiplied per pixel by mul [Hin][Win]) -> out [Hout][Wout][C]
__global__ __launch_bounds__(256) void resize_bilinear_hwc_kernel(const float *__restrict__ in, const float *__restrict__ mul, int Hin,
                                                                  int Win, int C, float *__restrict__ out, int Hout, int Wout) {
    const long total = (long)Hout * Wout * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int X = (int)(p % Wout), Y = (int)(p / Wout);
        int y0, y1, x0, x1;
        float ly, lx;
        est_bilin_src(Y, Hin, Hout, y0, y1, ly);
        est_bilin_src(X, Win, Wout, x0, x1, lx);
        const long p00 = (long)y0 * Win + x0, p01 = (long)y0 * Win + x1, p10 = (long)y1 * Win + x0, p11 = (long)y1 * Win + x1;
        float a = in[p00 * C + c], b = in[p01 * C + c], cc = in[p10 * C + c], d = in[p11 * C + c];
        if (mul) {
            a *= mul[p00];
            b *= mul[p01];
            cc *= mul[p10];
            d *= mul[p11];
        }
        const float hy = 1.f - ly, hx = 1.f - lx;
        out[i] = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
    }
}

// in [G][S*S][C] (16-byte chunks), out [So*So][9*G*C] with k = (ky*3+kx)*(G*C) + g*C + c, no padding, stride st
__global__ __launch_bounds__(256) void im2col3x3_strided_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, int G, int S,
                                                                int So, int st, int chunks_per_pixel) {
    const long total = (long)So * So * 9 * G * chunks_per_pixel;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks_per_pixel);
        long r = i / chunks_per_pixel;
        const int g = (int)(r % G);
        r /= G;
        const int tap = (int)(r % 9);
        r /= 9;
        const int xo = (int)(r % So), yo = (int)(r / So);
        const int yy = yo * st + tap / 3, xx = xo * st + tap % 3;
        out[i] = in[(((long)g * S + yy) * S + xx) * chunks_per_pixel + c];
    }
}

// out[c] = max (MEAN = false) or mean (MEAN = true) over rows of x [rows][ld]; one wave per 64 columns, rows strided
template <bool MEAN>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float *__restrict__ x, int ld, int rows, int cols, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float acc = MEAN ? 0.f : -FLT_MAX;
    if (c < cols)
        for (int r = wave; r < rows; r += 4) {
            const float v = x[(long)r * ld + c];
            acc = MEAN ? acc + v : fmaxf(acc, v);
        }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) {
        float a = part[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) a = MEAN ? a + part[w][lane] : fmaxf(a, part[w][lane]);
        out[c] = MEAN ? a / (float)rows : a;
    }
}

static inline int est_grid(long n) { return (int)std::min<long>((n + 255) / 256, (long)num_cus() * 32); }

}  // namespace sculpt

using namespace sculpt;

extern "C" {

int sculpt_resize_bilinear_hwc(const float *in_hwc, const float *mul_hw, int Hin, int Win, int C, float *out_hwc, int Hout, int Wout,
                               sculpt_stream_t stream) {
    SC_REQUIRE(in_hwc && out_hwc, "resize_bilinear_hwc: null argument");
    SC_REQUIRE(Hin >= 1 && Win >= 1 && Hout >= 1 && Wout >= 1 && C >= 1, "resize_bilinear_hwc: bad shape");
    hipLaunchKernelGGL(resize_bilinear_hwc_kernel, dim3(est_grid((long)Hout * Wout * C)), dim3(256), 0, as_stream(stream), in_hwc, mul_hw,
                       Hin, Win, C, out_hwc, Hout, Wout);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_im2col3x3_strided(const void *in, int n_groups, int S, int C, int elem_bytes, int stride, void *out, sculpt_stream_t stream) {
    SC_REQUIRE(in && out, "im2col3x3_strided: null argument");
    SC_REQUIRE(n_groups >= 1 && S >= 3 && stride >= 1 && (elem_bytes == 2 || elem_bytes == 4) && C >= 1 && (C * elem_bytes) % 16 == 0,
               "im2col3x3_strided: bad shape (C * elem_bytes must be a multiple of 16, S >= 3)");
    const int So = (S - 3) / stride + 1;
    const int cpp = C * elem_bytes / 16;
    const long total = (long)So * So * 9 * n_groups * cpp;
    hipLaunchKernelGGL(im2col3x3_strided_kernel, dim3(est_grid(total)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4 *>(in), reinterpret_cast<uint4 *>(out), n_groups, S, So, stride, cpp);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_col_reduce_f32(const float *x, int ld, int rows, int cols, int mean, float *out, sculpt_stream_t stream) {
    SC_REQUIRE(x && out && rows >= 1 && cols >= 1 && ld >= cols, "col_reduce_f32: bad argument");
    const dim3 grid((cols + 63) / 64);
    if (mean)
        hipLaunchKernelGGL(col_reduce_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, ld, rows, cols, out);
    else
        hipLaunchKernelGGL(col_reduce_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, ld, rows, cols, out);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
