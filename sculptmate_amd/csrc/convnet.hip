// Building blocks of a bf16 channel-last convolutional network on gfx950: used for U^2-Net, the salient-object
// network behind the reference's background removal (rembg/bg.py:149-238 -> rembg/sessions/u2net.py:16-46, an
// opaque ONNX graph there).  3x3 convolutions = im2col (this file) + the bf16 MFMA GEMM (gemm.hip, fused bias +
// ReLU, strided / column-limited output so a layer writes straight into its slice of a concatenation buffer).
// Activations are [H*W][ld] bf16 with the channels of one pixel contiguous; all kernels take (pointer to the first
// channel of the slice, row stride ld, number of real channels C).  Everything here is HBM-bound data movement.
#include "common.h"

namespace sculpt {

// rows [H*W][9*Cpad], k = (ky*3 + kx)*Cpad + c; channels C..Cpad-1 and out-of-image taps are zero.  16-byte chunks.
__global__ __launch_bounds__(256) void im2col3x3_dil_kernel(const uint16_t *__restrict__ in, int ld_in, int H, int W, int c8,
                                                            int cpad8, int dil, uint4 *__restrict__ out) {
    const long total = (long)H * W * 9 * cpad8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpad8);
        long r = i / cpad8;
        const int tap = (int)(r % 9);
        r /= 9;
        const int x = (int)(r % W), y = (int)(r / W);
        const int yy = y + (tap / 3 - 1) * dil, xx = x + (tap % 3 - 1) * dil;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < c8 && yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const uint4 *>(in + ((long)yy * W + xx) * ld_in + 8 * c);
        out[i] = v;
    }
}

// nn.MaxPool2d(2, stride=2, ceil_mode=True): out [ceil(H/2)*ceil(W/2)][ld_out], 8 channels per thread
__global__ __launch_bounds__(256) void maxpool2_kernel(const uint16_t *__restrict__ in, int ld_in, int H, int W, int c8,
                                                       uint16_t *__restrict__ out, int ld_out) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)Ho * Wo * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8);
        const long r = i / c8;
        const int x = (int)(r % Wo), y = (int)(r / Wo);
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = 2 * y + dy, xx = 2 * x + dx;
                if (yy < H && xx < W) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(in + ((long)yy * W + xx) * ld_in + 8 * c);
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        m[2 * k] = fmaxf(m[2 * k], bf16_to_f32((uint16_t)(w4[k] & 0xffff)));
                        m[2 * k + 1] = fmaxf(m[2 * k + 1], bf16_to_f32((uint16_t)(w4[k] >> 16)));
                    }
                }
            }
        uint4 o;
        o.x = f32_to_bf16(m[0]) | ((uint32_t)f32_to_bf16(m[1]) << 16);
        o.y = f32_to_bf16(m[2]) | ((uint32_t)f32_to_bf16(m[3]) << 16);
        o.z = f32_to_bf16(m[4]) | ((uint32_t)f32_to_bf16(m[5]) << 16);
        o.w = f32_to_bf16(m[6]) | ((uint32_t)f32_to_bf16(m[7]) << 16);
        *reinterpret_cast<uint4 *>(out + ((long)y * Wo + x) * ld_out + 8 * c) = o;
    }
}

// F.interpolate(mode="bilinear", align_corners=False) source coordinate (torch area_pixel_compute_source_index)
__device__ __forceinline__ void bilin_src(int dst, int in_size, int out_size, int &i0, int &i1, float &l1) {
    const float scale = (float)in_size / (float)out_size;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const uint16_t *__restrict__ in, int ld_in, int h, int w, int c8,
                                                                uint16_t *__restrict__ out, int ld_out, int H, int W) {
    const long total = (long)H * W * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8);
        const long r = i / c8;
        const int X = (int)(r % W), Y = (int)(r / W);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(Y, h, H, y0, y1, ly);
        bilin_src(X, w, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const uint4 a = *reinterpret_cast<const uint4 *>(in + ((long)y0 * w + x0) * ld_in + 8 * c);
        const uint4 b = *reinterpret_cast<const uint4 *>(in + ((long)y0 * w + x1) * ld_in + 8 * c);
        const uint4 cc = *reinterpret_cast<const uint4 *>(in + ((long)y1 * w + x0) * ld_in + 8 * c);
        const uint4 d = *reinterpret_cast<const uint4 *>(in + ((long)y1 * w + x1) * ld_in + 8 * c);
        const uint32_t A[4] = {a.x, a.y, a.z, a.w}, B[4] = {b.x, b.y, b.z, b.w}, C[4] = {cc.x, cc.y, cc.z, cc.w},
                       D[4] = {d.x, d.y, d.z, d.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sh = 16 * q;
                const float va = bf16_to_f32((uint16_t)(A[k] >> sh)), vb = bf16_to_f32((uint16_t)(B[k] >> sh));
                const float vc = bf16_to_f32((uint16_t)(C[k] >> sh)), vd = bf16_to_f32((uint16_t)(D[k] >> sh));
                // torch: hy * (hx * a + lx * b) + ly * (hx * c + lx * d)
                r2[q] = hy * (hx * va + lx * vb) + ly * (hx * vc + lx * vd);
            }
            o[k] = f32_to_bf16(r2[0]) | ((uint32_t)f32_to_bf16(r2[1]) << 16);
        }
        *reinterpret_cast<uint4 *>(out + ((long)Y * W + X) * ld_out + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// one-channel fp32 maps: out[Y][X] = bilinear(in [h][w] with element stride ld_in)
__global__ __launch_bounds__(256) void upsample_bilinear_f32_kernel(const float *__restrict__ in, int ld_in, int h, int w,
                                                                    float *__restrict__ out, int H, int W) {
    const long total = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)(i / W);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(Y, h, H, y0, y1, ly);
        bilin_src(X, w, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float a = in[((long)y0 * w + x0) * ld_in], b = in[((long)y0 * w + x1) * ld_in];
        const float c = in[((long)y1 * w + x0) * ld_in], d = in[((long)y1 * w + x1) * ld_in];
        out[i] = hy * (hx * a + lx * b) + ly * (hx * c + lx * d);
    }
}

// out = a + b on bf16 slices (the RSU residual hx1d + hxin), 8 channels per thread
__global__ __launch_bounds__(256) void add_bf16_kernel(const uint16_t *__restrict__ a, int lda, const uint16_t *__restrict__ b, int ldb,
                                                       uint16_t *__restrict__ out, int ldo, long rows, int c8) {
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8);
        const long r = i / c8;
        const uint4 x = *reinterpret_cast<const uint4 *>(a + r * lda + 8 * c);
        const uint4 y = *reinterpret_cast<const uint4 *>(b + r * ldb + 8 * c);
        const uint32_t X[4] = {x.x, x.y, x.z, x.w}, Y[4] = {y.x, y.y, y.z, y.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = bf16_to_f32((uint16_t)(X[k] & 0xffff)) + bf16_to_f32((uint16_t)(Y[k] & 0xffff));
            const float hi = bf16_to_f32((uint16_t)(X[k] >> 16)) + bf16_to_f32((uint16_t)(Y[k] >> 16));
            o[k] = f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
        }
        *reinterpret_cast<uint4 *>(out + r * ldo + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// d0 = sigmoid(sum_k w[k] * d[k] + b) over n_maps one-channel maps [n_maps][n]
__global__ __launch_bounds__(256) void fuse_sigmoid_kernel(const float *__restrict__ maps, int n_maps, long n, const float *__restrict__ w,
                                                           float bias, float *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = bias;
        for (int k = 0; k < n_maps; ++k) s += w[k] * maps[(long)k * n + i];
        out[i] = 1.0f / (1.0f + expf(-s));
    }
}

static inline int grid_for_n(long n) { return (int)std::min<long>((n + 255) / 256, (long)num_cus() * 32); }

}  // namespace sculpt

using namespace sculpt;

extern "C" {

int sculpt_im2col3x3_dilated(const uint16_t *in, int ld_in, int H, int W, int C, int C_pad, int dilation, uint16_t *out,
                             sculpt_stream_t stream) {
    SC_REQUIRE(in && out, "im2col3x3_dilated: null argument");
    SC_REQUIRE(H >= 1 && W >= 1 && C >= 8 && C % 8 == 0 && C_pad % 8 == 0 && C_pad >= C && ld_in % 8 == 0 && dilation >= 1,
               "im2col3x3_dilated: bad shape H=%d W=%d C=%d C_pad=%d ld=%d", H, W, C, C_pad, ld_in);
    const long total = (long)H * W * 9 * (C_pad / 8);
    hipLaunchKernelGGL(im2col3x3_dil_kernel, dim3(grid_for_n(total)), dim3(256), 0, as_stream(stream), in, ld_in, H, W, C / 8,
                       C_pad / 8, dilation, reinterpret_cast<uint4 *>(out));
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_maxpool2x2_ceil(const uint16_t *in, int ld_in, int H, int W, int C, uint16_t *out, int ld_out, sculpt_stream_t stream) {
    SC_REQUIRE(in && out && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0, "maxpool2x2_ceil: bad argument");
    const long total = (long)((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for_n(total)), dim3(256), 0, as_stream(stream), in, ld_in, H, W, C / 8, out, ld_out);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_upsample_bilinear_bf16(const uint16_t *in, int ld_in, int h, int w, int C, uint16_t *out, int ld_out, int H, int W,
                                  sculpt_stream_t stream) {
    SC_REQUIRE(in && out && h >= 1 && w >= 1 && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0,
               "upsample_bilinear_bf16: bad argument");
    const long total = (long)H * W * (C / 8);
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(grid_for_n(total)), dim3(256), 0, as_stream(stream), in, ld_in, h, w, C / 8,
                       out, ld_out, H, W);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_upsample_bilinear_f32(const float *in, int ld_in, int h, int w, float *out, int H, int W, sculpt_stream_t stream) {
    SC_REQUIRE(in && out && h >= 1 && w >= 1 && H >= 1 && W >= 1 && ld_in >= 1, "upsample_bilinear_f32: bad argument");
    hipLaunchKernelGGL(upsample_bilinear_f32_kernel, dim3(grid_for_n((long)H * W)), dim3(256), 0, as_stream(stream), in, ld_in, h, w,
                       out, H, W);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_add_bf16(const uint16_t *a, int lda, const uint16_t *b, int ldb, uint16_t *out, int ldo, int64_t rows, int C,
                    sculpt_stream_t stream) {
    SC_REQUIRE(a && b && out && rows >= 1 && C >= 8 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "add_bf16: bad argument");
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for_n(rows * (C / 8))), dim3(256), 0, as_stream(stream), a, lda, b, ldb, out, ldo,
                       (long)rows, C / 8);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_fuse_sigmoid(const float *maps, int n_maps, int64_t n, const float *w, float bias, float *out, sculpt_stream_t stream) {
    SC_REQUIRE(maps && w && out && n_maps >= 1 && n >= 1, "fuse_sigmoid: bad argument");
    hipLaunchKernelGGL(fuse_sigmoid_kernel, dim3(grid_for_n(n)), dim3(256), 0, as_stream(stream), maps, n_maps, (long)n, w, bias, out);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
