// Normalisation / layout kernels of the TripoSR transformer stack (all HBM-bound, vectorised).
//
//   sculpt_layernorm          nn.LayerNorm  basic_transformer_block.py:98,114,132 (eps 1e-5);
//                             HF ViT layernorm_before/after/final (eps 1e-12)
//   sculpt_groupnorm_tokens   GroupNorm(32, C, eps 1e-6) + permute(0,2,1)  transformer_1d.py:183-187
//   sculpt_transpose_add      permute(0,2,1) + residual                    transformer_1d.py:211-217
//   sculpt_vit_patchify       (x-mean)/std (tokenizers/image.py:48) + conv16/16 im2col
//   sculpt_vit_assemble       [CLS] + patches + position embeddings (HF ViTEmbeddings)
//   sculpt_upsample_scatter   ConvTranspose2d(k2,s2) pixel interleave + bias  network_utils.py:20-32
//   sculpt_cast_bf16
#include "common.h"
#include "limbs.h"

namespace sculpt {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// one wave per row; cols % 256 == 0 handled with float4 per lane per step, generic tail otherwise
template <bool IN_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ xf, const uint16_t *__restrict__ xb,
                                                        int ldx, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps,
                                                        uint16_t *__restrict__ y, int ldy, float *__restrict__ yf,
                                                        int rows, int cols, unsigned char *__restrict__ y_lt = nullptr, int lt_fmt = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int MAXV = 8;  // up to 8 * 64 * 4 = 2048 columns
    float v[MAXV][4];
    const int nv = cols / 256;  // full float4 steps
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            if (IN_BF16) {
                const uint2 p = *reinterpret_cast<const uint2 *>(xb + (long)row * ldx + c);
                v[i][0] = __uint_as_float(p.x << 16); v[i][1] = __uint_as_float(p.x & 0xffff0000u);
                v[i][2] = __uint_as_float(p.y << 16); v[i][3] = __uint_as_float(p.y & 0xffff0000u);
            } else {
                const float4 p = *reinterpret_cast<const float4 *>(xf + (long)row * ldx + c);
                v[i][0] = p.x; v[i][1] = p.y; v[i][2] = p.z; v[i][3] = p.w;
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            const float4 gm = *reinterpret_cast<const float4 *>(gamma + c);
            const float4 bt = *reinterpret_cast<const float4 *>(beta + c);
            float o[4];
            o[0] = (v[i][0] - mean) * rstd * gm.x + bt.x;
            o[1] = (v[i][1] - mean) * rstd * gm.y + bt.y;
            o[2] = (v[i][2] - mean) * rstd * gm.z + bt.z;
            o[3] = (v[i][3] - mean) * rstd * gm.w + bt.w;
            if (y) {
                uint2 pk;
                pk.x = pack_bf16x2(o[0], o[1]);
                pk.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2 *>(y + (long)row * ldy + c) = pk;
            }
            if (yf) *reinterpret_cast<float4 *>(yf + (long)row * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (y_lt) lt_store4(y_lt, cols >> 3, row, c, o, lt_fmt);   // the normalised row as limbs (the operand of gemm_l3p)
        }
    }
}

// GroupNorm statistics: one workgroup per group, x [C][T] so a group is contiguous (C/G * T floats)
__global__ __launch_bounds__(1024) void groupnorm_stats_kernel(const float *__restrict__ x, long group_elems,
                                                               float eps, float *__restrict__ stats) {
    __shared__ float red[16];
    __shared__ float mean_s;
    const float *p = x + (long)blockIdx.x * group_elems;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (long i = threadIdx.x * 4L; i < group_elems; i += 4096L) {
        const float4 v = *reinterpret_cast<const float4 *>(p + i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        mean_s = t / (float)group_elems;
    }
    __syncthreads();
    const float mean = mean_s;
    float q = 0.f;
    for (long i = threadIdx.x * 4L; i < group_elems; i += 4096L) {
        const float4 v = *reinterpret_cast<const float4 *>(p + i);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wave] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        stats[2 * blockIdx.x] = mean;
        stats[2 * blockIdx.x + 1] = rsqrtf(t / (float)group_elems + eps);
    }
}

// normalise + transpose: x [C][T] -> y [T][C] bf16, 64x64 tiles through LDS
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float *__restrict__ x, int C, int T, int cpg,
                                                              const float *__restrict__ stats,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, uint16_t *__restrict__ y,
                                                              float *__restrict__ yf) {
    __shared__ float tile[64][65];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, t = t0 + tx;
        float v = 0.f;
        if (c < C && t < T) {
            const int gi = c / cpg;
            v = (x[(long)c * T + t] - stats[2 * gi]) * stats[2 * gi + 1] * gamma[c] + beta[c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, c = c0 + tx;
        if (t < T && c < C) {
            if (y) y[(long)t * C + c] = f32_to_bf16(tile[tx][r]);
            if (yf) yf[(long)t * C + c] = tile[tx][r];
        }
    }
}

// out[c][t] = x[t][c] + res[c][t]
__global__ __launch_bounds__(256) void transpose_add_kernel(const float *__restrict__ x_tc, const float *__restrict__ res_ct,
                                                            float *__restrict__ out_ct, int T, int C) {
    __shared__ float tile[64][65];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? x_tc[(long)t * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < T) out_ct[(long)c * T + t] = tile[tx][r] + res_ct[(long)c * T + t];
    }
}

// image [S][S][3] fp32 (HWC, 0..1) -> patches [ (S/P)^2 ][ld >= 3*P*P], column = c*P*P + py*P + px, columns
// 3*P*P..ld-1 zero (K padding for the GEMM); pixels beyond (S/P)*P are ignored like a stride-P convolution does
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ img, int S, int P, int ld, float m0, float m1,
                                                       float m2, float s0, float s1, float s2,
                                                       uint16_t *__restrict__ patches, float *__restrict__ patches_f32) {
    const int np = S / P, cols = 3 * P * P;
    const long total = (long)np * np * ld;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % ld);
        const int patch = (int)(i / ld);
        float o = 0.f;
        if (col < cols) {
            const int c = col / (P * P), py = (col / P) % P, px = col % P;
            const int gy = (patch / np) * P + py, gx = (patch % np) * P + px;
            const float v = img[((long)gy * S + gx) * 3 + c];
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            o = (v - mean) / sd;
        }
        if (patches) patches[i] = f32_to_bf16(o);
        if (patches_f32) patches_f32[i] = o;
    }
}

__global__ __launch_bounds__(256) void vit_assemble_kernel(const float *__restrict__ patch_out, const float *__restrict__ cls,
                                                           const float *__restrict__ pos, float *__restrict__ tokens,
                                                           int n_patches, int hidden) {
    const long total = (long)(n_patches + 1) * hidden;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i / hidden), c = (int)(i % hidden);
        const float v = (t == 0) ? cls[c] : patch_out[(long)(t - 1) * hidden + c];
        tokens[i] = v + pos[i];
    }
}

// g [3*S*S][ldg] (column = co*4 + dy*2 + dx) -> planes [3][Co][2S][2S]
__global__ __launch_bounds__(256) void upsample_scatter_kernel(const float *__restrict__ g, int ldg,
                                                               const float *__restrict__ bias, float *__restrict__ planes,
                                                               int S, int Co) {
    const int S2 = 2 * S;
    const long total = 3L * Co * S2 * S2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int X = (int)(i % S2), Y = (int)((i / S2) % S2);
        const int co = (int)((i / ((long)S2 * S2)) % Co), pl = (int)(i / ((long)S2 * S2 * Co));
        const int w = X >> 1, dx = X & 1, hh = Y >> 1, dy = Y & 1;
        const long tok = (long)pl * S * S + (long)hh * S + w;
        planes[i] = g[tok * ldg + co * 4 + dy * 2 + dx] + bias[co];
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ x, uint16_t *__restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f32_to_bf16(x[i]);
}

static inline int grid_for(long n) { return (int)std::min<long>((n + 255) / 256, 2048); }

// one thread per (row, 64-column slice): 16 float4 loads (256 contiguous bytes), two-pass (mean, M2), bf16 copy
__global__ __launch_bounds__(256) void row_slice_stats_kernel(const float *__restrict__ x, int ldx, int rows, int slots,
                                                              float *__restrict__ stats, int stats_ld,
                                                              uint16_t *__restrict__ xb, int ldb) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)rows * slots) return;
    const int slot = (int)(t / rows), row = (int)(t % rows);  // row fastest: the slice-major stores are coalesced
    const float4 *src = reinterpret_cast<const float4 *>(x + (long)row * ldx + slot * 64);
    float4 v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = src[i]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    const float mean = s * (1.0f / 64.0f);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        m2 = fmaf(a, a, m2); m2 = fmaf(b, b, m2); m2 = fmaf(c, c, m2); m2 = fmaf(d, d, m2);
    }
    reinterpret_cast<float2 *>(stats)[(long)slot * stats_ld + row] = make_float2(mean, m2);
    if (xb) {
        uint2 *dst = reinterpret_cast<uint2 *>(xb + (long)row * ldb + slot * 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint2 pk;
            pk.x = pack_bf16x2(v[i].x, v[i].y);
            pk.y = pack_bf16x2(v[i].z, v[i].w);
            dst[i] = pk;
        }
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

int sculpt_row_slice_stats(const float *x, int ldx, int rows, int cols, float *stats, int stats_ld, uint16_t *x_bf16, int ldb,
                           sculpt_stream_t stream) {
    SC_REQUIRE(x && stats && stats_ld >= rows, "row_slice_stats: null argument or stats_ld < rows");
    SC_REQUIRE(cols >= 64 && cols % 64 == 0 && ldx % 4 == 0 && (!x_bf16 || ldb % 4 == 0), "row_slice_stats: cols=%d must be a multiple of 64", cols);
    if (rows <= 0) return 0;
    const int slots = cols / 64;
    hipLaunchKernelGGL(row_slice_stats_kernel, dim3(cdiv((long)rows * slots, 256)), dim3(256), 0, as_stream(stream), x, ldx, rows,
                       slots, stats, stats_ld, x_bf16, ldb);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_layernorm(const float *x_f32, const uint16_t *x_bf16, int ldx, const float *gamma, const float *beta,
                     float eps, uint16_t *y, int ldy, float *y_f32, int rows, int cols, sculpt_stream_t stream) {
    SC_REQUIRE((x_f32 != nullptr) != (x_bf16 != nullptr), "layernorm: give exactly one of x_f32 / x_bf16");
    SC_REQUIRE(gamma && beta && (y || y_f32), "layernorm: null argument");
    SC_REQUIRE(cols % 256 == 0 && cols <= 2048, "layernorm: cols=%d must be a multiple of 256 and <= 2048", cols);
    SC_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm: ldx/ldy must be multiples of 4");
    if (rows <= 0) return 0;
    hipStream_t st = as_stream(stream);
    if (x_bf16)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x_f32, x_bf16, ldx, gamma, beta,
                           eps, y, ldy, y_f32, rows, cols);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x_f32, x_bf16, ldx, gamma, beta,
                           eps, y, ldy, y_f32, rows, cols);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_layernorm_limbs(const float *x, int ldx, const float *gamma, const float *beta, float eps, void *y_lt, int format,
                           float *y_f32, int ldy, int rows, int cols, sculpt_stream_t stream) {
    SC_REQUIRE(x && gamma && beta && y_lt, "layernorm_limbs: null argument");
    SC_REQUIRE(format == LT_BF16X3 || format == LT_F16X2, "layernorm_limbs: unknown limb format %d", format);
    SC_REQUIRE(cols % 256 == 0 && cols <= 2048 && ldx % 4 == 0 && ((uintptr_t)y_lt & 15) == 0 && (!y_f32 || ldy % 4 == 0),
               "layernorm_limbs: cols=%d must be a multiple of 256 and <= 2048, ldx / ldy multiples of 4, y_lt 16-byte aligned", cols);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, (const uint16_t *)nullptr, ldx,
                       gamma, beta, eps, (uint16_t *)nullptr, ldy, y_f32, rows, cols, reinterpret_cast<unsigned char *>(y_lt), format);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_groupnorm_tokens(const float *x, int C, int T, int G, const float *gamma, const float *beta, float eps,
                            uint16_t *y, float *y_f32, float *stats_ws, sculpt_stream_t stream) {
    SC_REQUIRE(x && gamma && beta && (y || y_f32) && stats_ws, "groupnorm: null argument");
    SC_REQUIRE(G >= 1 && C % G == 0, "groupnorm: C=%d not divisible by G=%d", C, G);
    const long ge = (long)(C / G) * T;
    SC_REQUIRE(ge % 4 == 0, "groupnorm: group size must be a multiple of 4");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(G), dim3(1024), 0, st, x, ge, eps, stats_ws);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(cdiv(T, 64), cdiv(C, 64)), dim3(256), 0, st, x, C, T, C / G,
                       stats_ws, gamma, beta, y, y_f32);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_transpose_add(const float *x_tc, const float *residual_ct, float *out_ct, int T, int C,
                         sculpt_stream_t stream) {
    SC_REQUIRE(x_tc && residual_ct && out_ct, "transpose_add: null argument");
    hipLaunchKernelGGL(transpose_add_kernel, dim3(cdiv(T, 64), cdiv(C, 64)), dim3(256), 0, as_stream(stream), x_tc,
                       residual_ct, out_ct, T, C);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_vit_patchify(const float *image_hwc, int S, int P, const float *mean3_host, const float *std3_host,
                        uint16_t *patches, float *patches_f32, int ld, sculpt_stream_t stream) {
    SC_REQUIRE(image_hwc && mean3_host && std3_host && (patches || patches_f32), "vit_patchify: null argument");
    SC_REQUIRE(P > 0 && S >= P, "vit_patchify: bad S=%d P=%d", S, P);
    if (ld <= 0) ld = 3 * P * P;
    SC_REQUIRE(ld >= 3 * P * P, "vit_patchify: ld=%d < 3*P*P", ld);
    const long total = (long)(S / P) * (S / P) * ld;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), image_hwc, S, P, ld,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2], patches, patches_f32);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_vit_assemble(const float *patch_out, const float *cls, const float *pos, float *tokens, int n_patches,
                        int hidden, sculpt_stream_t stream) {
    SC_REQUIRE(patch_out && cls && pos && tokens, "vit_assemble: null argument");
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for((long)(n_patches + 1) * hidden)), dim3(256), 0,
                       as_stream(stream), patch_out, cls, pos, tokens, n_patches, hidden);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_upsample_scatter(const float *g, int ldg, const float *bias, float *planes, int S, int Co,
                            sculpt_stream_t stream) {
    SC_REQUIRE(g && bias && planes, "upsample_scatter: null argument");
    SC_REQUIRE(ldg >= 4 * Co, "upsample_scatter: ldg=%d < 4*Co", ldg);
    hipLaunchKernelGGL(upsample_scatter_kernel, dim3(grid_for(3L * Co * 4 * S * S)), dim3(256), 0, as_stream(stream), g,
                       ldg, bias, planes, S, Co);
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_cast_bf16(const float *x, uint16_t *y, int64_t n, sculpt_stream_t stream) {
    SC_REQUIRE(x && y, "cast_bf16: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, (long)n);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// ImagePreprocessor resize: F.interpolate(bilinear, align_corners=False, antialias=True) on an HWC
// float image (TripoSR/tsr/utils.py:82-88; the add-on feeds 1024^2, preprocessing.py:126).
// torch's separable triangle filter: scale = in/out, support = max(scale, 1), centre = scale*(i+0.5),
// taps [xmin, xmin+xsize), weights max(0, 1 - |(j + xmin - centre + 0.5)/max(scale,1)|) normalised;
// width pass first, then height pass (fp32 intermediate), like the CPU kernel.
// ---------------------------------------------------------------------------------------------
namespace sculpt {

__device__ __forceinline__ void aa_taps(int i, int in_size, float scale, int *xmin, int *xsize, float *support_inv,
                                        float *center) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    *center = scale * ((float)i + 0.5f);
    *xmin = max(0, (int)(*center - support + 0.5f));
    *xsize = min(in_size, (int)(*center + support + 0.5f)) - *xmin;
    *support_inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
}

// dir = 0: resize along x (in [H][Win][C] -> out [H][Wout][C]); dir = 1: along y ([Hin][W][C] -> [Hout][W][C])
__global__ __launch_bounds__(256) void resize_aa_kernel(const float *__restrict__ in, float *__restrict__ out, int H,
                                                        int W, int C, int in_size, int out_size, float scale, int dir) {
    const long total = (long)H * W * C;  // output elements
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int x = (int)((e / C) % W), y = (int)(e / ((long)C * W));
        const int i = dir == 0 ? x : y;
        int xmin, xsize;
        float inv, center;
        aa_taps(i, in_size, scale, &xmin, &xsize, &inv, &center);
        float tw = 0.f;
        for (int j = 0; j < xsize; ++j) {
            const float t = fabsf(((float)(j + xmin) - center + 0.5f) * inv);
            tw += t < 1.0f ? 1.0f - t : 0.f;
        }
        float acc = 0.f;
        for (int j = 0; j < xsize; ++j) {
            const float t = fabsf(((float)(j + xmin) - center + 0.5f) * inv);
            const float wgt = (t < 1.0f ? 1.0f - t : 0.f) / tw;
            const long src = dir == 0 ? ((long)y * in_size + (xmin + j)) * C + c : ((long)(xmin + j) * W + x) * C + c;
            acc += in[src] * wgt;
        }
        out[e] = acc;
    }
}

}  // namespace sculpt

extern "C" int sculpt_resize_aa_bilinear(const float *in_hwc, int Hin, int Win, int C, float *tmp, float *out_hwc,
                                         int Hout, int Wout, sculpt_stream_t stream) {
    using namespace sculpt;
    SC_REQUIRE(in_hwc && tmp && out_hwc && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0, "resize_aa: bad argument");
    hipStream_t st = as_stream(stream);
    // width pass: [Hin][Win][C] -> tmp [Hin][Wout][C]
    hipLaunchKernelGGL(resize_aa_kernel, dim3(grid_for((long)Hin * Wout * C)), dim3(256), 0, st, in_hwc, tmp, Hin, Wout, C,
                       Win, Wout, (float)Win / (float)Wout, 0);
    SC_LAUNCH_CHECK();
    // height pass: tmp [Hin][Wout][C] -> out [Hout][Wout][C]
    hipLaunchKernelGGL(resize_aa_kernel, dim3(grid_for((long)Hout * Wout * C)), dim3(256), 0, st, tmp, out_hwc, Hout, Wout, C,
                       Hin, Hout, (float)Hin / (float)Hout, 1);
    SC_LAUNCH_CHECK();
    return 0;
}
