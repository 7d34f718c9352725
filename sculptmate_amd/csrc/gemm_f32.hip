// fp32 GEMM on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32) + row softmax: the building blocks of
// the fp32 "parity mode" of the transformer stack (TSR(precision="fp32")).
//
// BASELINE config 2 runs the transformer in bf16 (gemm.hip / attention.hip); the reference itself is fp32
// end to end (TripoSR/generate.py:36-39, no autocast).  This mode reproduces the reference's fp32 numbers
// to fp32 rounding so that image -> mesh parity can be demonstrated end to end; it is ~5x slower than
// the bf16 path and is not what bench.py times.
//
//   out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] ) (+ residual[m][n])      A, W, out: fp32
// 128 x 128 x 16 tiles, 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32; operands staged k-major in LDS
// (register-staged, double buffered) so that a lane's MFMA operand is one conflict-free ds_read_b32.
// Weight tile = A operand -> a lane owns 4 consecutive output columns (float4 epilogue).
#include "gemm_f32.h"

namespace sculpt {

static constexpr int FBK = 16, FLD = 132;  // LDS row stride (floats), padded

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args g_in) {
    const GemmF32Args g = f32_batch_entry(g_in);
    __shared__ float Ws[2][FBK][FLD];
    __shared__ float As[2][FBK][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? FBW / 2 : FBW;
    const int n0 = blockIdx.x * NOUT, m0 = blockIdx.y * FBM;
    auto wrow = [&](int j) -> int { return f32_tile_wrow<EPI>(g, n0, j); };
    // staging: a tile is 128 rows x 16 k = 512 float4; thread t handles rows t/4 and t/4 + 64, k-quad t%4
    const int sr = tid >> 2, kq = tid & 3;
    const float *wp0 = g.W + (long)wrow(sr) * g.ldw + 4 * kq;
    const float *wp1 = g.W + (long)wrow(sr + 64) * g.ldw + 4 * kq;
    const float *ap0 = g.A + (long)min(m0 + sr, g.M - 1) * g.lda + 4 * kq;
    const float *ap1 = g.A + (long)min(m0 + sr + 64, g.M - 1) * g.lda + 4 * kq;
    float4 rw0, rw1, ra0, ra1;
    const int nk = g.K / FBK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define GLOAD(kt)                                                        \
    do {                                                                 \
        rw0 = *reinterpret_cast<const float4 *>(wp0 + (kt) * FBK);       \
        rw1 = *reinterpret_cast<const float4 *>(wp1 + (kt) * FBK);       \
        ra0 = *reinterpret_cast<const float4 *>(ap0 + (kt) * FBK);       \
        ra1 = *reinterpret_cast<const float4 *>(ap1 + (kt) * FBK);       \
    } while (0)
#define SWRITE(b)                                                                                              \
    do {                                                                                                       \
        Ws[b][4 * kq][sr] = rw0.x; Ws[b][4 * kq + 1][sr] = rw0.y; Ws[b][4 * kq + 2][sr] = rw0.z; Ws[b][4 * kq + 3][sr] = rw0.w; \
        Ws[b][4 * kq][sr + 64] = rw1.x; Ws[b][4 * kq + 1][sr + 64] = rw1.y; Ws[b][4 * kq + 2][sr + 64] = rw1.z; Ws[b][4 * kq + 3][sr + 64] = rw1.w; \
        As[b][4 * kq][sr] = ra0.x; As[b][4 * kq + 1][sr] = ra0.y; As[b][4 * kq + 2][sr] = ra0.z; As[b][4 * kq + 3][sr] = ra0.w; \
        As[b][4 * kq][sr + 64] = ra1.x; As[b][4 * kq + 1][sr + 64] = ra1.y; As[b][4 * kq + 2][sr + 64] = ra1.z; As[b][4 * kq + 3][sr + 64] = ra1.w; \
    } while (0)

    GLOAD(0);
    SWRITE(0);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int b = kt & 1;
        if (kt + 1 < nk) GLOAD(kt + 1);
#pragma unroll
        for (int s = 0; s < FBK / 2; ++s) {
            const int k = 2 * s + lh;
            const float a0 = Ws[b][k][wr * 64 + l31], a1 = Ws[b][k][wr * 64 + 32 + l31];
            const float b0 = As[b][k][wc * 64 + l31], b1 = As[b][k][wc * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) SWRITE(b ^ 1);
        __syncthreads();
    }
#undef GLOAD
#undef SWRITE

    f32_tile_epilogue<EPI>(g, acc, n0, m0, wr, wc, l31, lh);
}

// in-place row softmax of x [rows][ld] over the first `cols` columns; columns [cols, pad_cols) are zeroed
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ x, int ld, int rows, int cols, int pad_cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float *p = x + (long)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(p[c] - mx);
        p[c] = e;
        s += e;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) p[c] *= inv;
    for (int c = cols + lane; c < pad_cols; c += 64) p[c] = 0.f;
}

// The same with the row held in registers (cols <= 64 * NPL): one read and one write of the score matrix instead of three reads
// and two writes -- the exact-fp32 mode's attention is bound by exactly that traffic (600 MB of scores per backbone self-attention).
template <int NPL>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float *__restrict__ x, int ld, int rows, int cols, int pad_cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float *p = x + (long)row * ld;
    float v[NPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < cols ? p[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        v[i] = expf(v[i] - mx);   // exp(-inf) = 0 for the columns past `cols`
        s += v[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = lane + 64 * i;
        if (c < pad_cols) p[c] = c < cols ? v[i] * inv : 0.f;
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" {

int sculpt_gemm_f32_ex(const float *A, int lda, const float *W, int ldw, const float *bias, const float *residual, int ldr,
                       float *out, int ldo, float *out_t, int ldt, int n_split, int w_rows, int M, int N, int K, float alpha,
                       int epilogue, int arithmetic, int batch, int64_t a_bs, int64_t w_bs, int64_t o_bs, sculpt_stream_t stream) {
    SC_REQUIRE(A && W && (out || out_t), "gemm_f32: null argument");
    SC_REQUIRE(arithmetic == SCULPT_F32_EXACT || arithmetic == SCULPT_F32_BF16L3, "gemm_f32: arithmetic must be 0 (exact fp32 MFMA) or 1 (three-limb bf16)");
    const int kq = arithmetic == SCULPT_F32_BF16L3 ? 32 : FBK;
    SC_REQUIRE(M >= 1 && N >= 4 && K >= kq && K % kq == 0, "gemm_f32: bad shape M=%d N=%d K=%d (K %% %d == 0)", M, N, K, kq);
    SC_REQUIRE(N % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && (!residual || ldr % 4 == 0),
               "gemm_f32: N and the row strides must be multiples of 4");
    SC_REQUIRE(batch >= 1 && batch <= 65535 && (batch == 1 || (a_bs % 4 == 0 && w_bs % 4 == 0 && o_bs % 4 == 0)),
               "gemm_f32: bad batch %d / batch strides (multiples of 4 elements)", batch);
    if (n_split <= 0 || n_split > N) n_split = N;
    SC_REQUIRE(n_split % 4 == 0 && (n_split == N || out_t), "gemm_f32: bad n_split");
    if (w_rows <= 0 || w_rows > N) w_rows = N;
    if (!bias) {   // unconditional float4 bias loads in the epilogue: a zero page stands in for a missing vector
        long zn = 0;
        bias = zero_floats_page(&zn);
        SC_REQUIRE(bias && (epilogue == SCULPT_EPI_GEGLU ? 2L * N : (long)N) + 4 <= zn, "gemm_f32: N=%d too large without a bias", N);
    }
    SC_REQUIRE(((uintptr_t)bias & 15) == 0, "gemm_f32: bias must be 16-byte aligned");
    GemmF32Args g{A, lda, W, ldw, bias, residual, ldr, out, ldo, out_t, ldt, M, N, K, n_split, w_rows, alpha,
                  batch > 1 ? (long)a_bs : 0, batch > 1 ? (long)w_bs : 0, batch > 1 ? (long)o_bs : 0};
    hipStream_t st = as_stream(stream);
    const int mt = cdiv(M, FBM);
    if (epilogue == SCULPT_EPI_GEGLU) SC_REQUIRE(N % 64 == 0 && out && !residual && !out_t, "gemm_f32(GEGLU): N %% 64 == 0, plain output only");
    else SC_REQUIRE(epilogue == SCULPT_EPI_NONE || epilogue == SCULPT_EPI_GELU || epilogue == SCULPT_EPI_RELU, "gemm_f32: unknown epilogue %d", epilogue);
    if (arithmetic == SCULPT_F32_BF16L3) {
        gemm_l3_launch(g, epilogue, batch, st);
    } else if (epilogue == SCULPT_EPI_GEGLU) {
        hipLaunchKernelGGL(gemm_f32_kernel<SCULPT_EPI_GEGLU>, dim3(N / 64, mt, batch), dim3(256), 0, st, g);
    } else if (epilogue == SCULPT_EPI_GELU) {
        hipLaunchKernelGGL(gemm_f32_kernel<SCULPT_EPI_GELU>, dim3(cdiv(N, FBW), mt, batch), dim3(256), 0, st, g);
    } else if (epilogue == SCULPT_EPI_RELU) {
        hipLaunchKernelGGL(gemm_f32_kernel<SCULPT_EPI_RELU>, dim3(cdiv(N, FBW), mt, batch), dim3(256), 0, st, g);
    } else {
        hipLaunchKernelGGL(gemm_f32_kernel<SCULPT_EPI_NONE>, dim3(cdiv(N, FBW), mt, batch), dim3(256), 0, st, g);
    }
    SC_LAUNCH_CHECK();
    return 0;
}

int sculpt_gemm_f32(const float *A, int lda, const float *W, int ldw, const float *bias, const float *residual, int ldr,
                    float *out, int ldo, float *out_t, int ldt, int n_split, int w_rows, int M, int N, int K, float alpha,
                    int epilogue, sculpt_stream_t stream) {
    return sculpt_gemm_f32_ex(A, lda, W, ldw, bias, residual, ldr, out, ldo, out_t, ldt, n_split, w_rows, M, N, K, alpha, epilogue,
                              SCULPT_F32_EXACT, 1, 0, 0, 0, stream);
}

int sculpt_softmax_rows_f32(float *x, int ld, int rows, int cols, int pad_cols, sculpt_stream_t stream) {
    SC_REQUIRE(x && rows >= 1 && cols >= 1 && pad_cols >= cols && pad_cols <= ld, "softmax_rows: bad argument");
    if (pad_cols <= 64 * 17) hipLaunchKernelGGL(softmax_rows_reg_kernel<17>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, ld, rows, cols, pad_cols);
    else if (pad_cols <= 64 * 48) hipLaunchKernelGGL(softmax_rows_reg_kernel<48>, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, ld, rows, cols, pad_cols);
    else hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, as_stream(stream), x, ld, rows, cols, pad_cols);
    SC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
