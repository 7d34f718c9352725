// bf16 MFMA GEMM with fused epilogues for the TripoSR transformer stack on gfx950.
//
// Replaces every nn.Linear of the hot path (reference file:line):
//   Attention.to_q/to_k/to_v/to_out       TripoSR/tsr/models/transformer/attention.py:194-206
//   FeedForward / GEGLU                   TripoSR/tsr/models/transformer/basic_transformer_block.py:209-315
//   Transformer1D.proj_in / proj_out      TripoSR/tsr/models/transformer/transformer_1d.py:88,120
//   HF ViT query/key/value/dense/MLP, patch-embedding conv (as a GEMM over 16x16 patches)
//   TriplaneUpsampleNetwork (ConvTranspose2d k2 s2 == GEMM + pixel interleave)  network_utils.py:20-32
//
// out[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] ) (+ residual[m][n])
//
// Tiling: 128 activation rows x 128 weight rows per 256-thread workgroup (4 waves as 2x2), BK = 64,
// v_mfma_f32_16x16x32_bf16.  The WEIGHT tile is the MFMA A operand and the ACTIVATION tile the B
// operand, so a lane ends up holding 4 consecutive output columns n of one row m: bias / GEGLU /
// residual are per-lane vector ops and the store is one 16-byte (fp32) or 8-byte (bf16) write.
// Both operands are K-contiguous in HBM, which is exactly the fragment shape (8 consecutive k per
// lane), so tiles are staged row-major into LDS with 16-byte chunks XOR-swizzled by (row>>1)&7:
// every ds_read_b128 lane group then covers all 64 banks (conflict-free).
// Global->LDS staging is register-prefetched one K-tile ahead (loads issued before the MFMA
// block, ds_write after it), LDS double-buffered: one barrier per K-tile.
#include "common.h"

namespace sculpt {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

static constexpr int BM = 128;   // activation rows per block
static constexpr int BW = 128;   // weight rows per block
static constexpr int BK = 64;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// byte offset of 16-byte chunk c (0..7) of row r in a [rows][64 bf16] LDS tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

struct GemmArgs {
    const uint16_t *A; int lda;
    const uint16_t *W; int ldw;
    const float *bias;
    const float *residual; int ldr;
    float *out_f32; uint16_t *out_bf16; int ldo;
    uint16_t *out_t; int ldt;
    int M, N, K;
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BM * 128];  // [buf][W|A][tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware order is irrelevant here (tiles share little); plain 2D grid: x = n tile, y = m tile
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? 64 : 128;  // output columns per block
    const int n0 = blockIdx.x * NOUT;
    const int m0 = blockIdx.y * BM;

    // global row of tile row j of the weight operand
    auto wrow = [&](int j) -> int {
        if (EPI == SCULPT_EPI_GEGLU) {
            const int sub = j >> 4, within = j & 15;
            return ((sub & 1) ? g.N : 0) + n0 + (sub >> 1) * 16 + within;
        }
        return n0 + j;
    };

    // staging: 1024 chunks per operand tile, 4 per thread: chunk id = tid + 256*i -> row id>>3, c id&7
    const uint16_t *wsrc[4];
    const uint16_t *asrc[4];
    int sdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i, r = id >> 3, c = id & 7;
        wsrc[i] = g.W + (long)wrow(r) * g.ldw + c * 8;
        const int m = min(m0 + r, g.M - 1);
        asrc[i] = g.A + (long)m * g.lda + c * 8;
        sdst[i] = lds_off(r, c);
    }
    uint4 wreg[4], areg[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wreg[i] = *reinterpret_cast<const uint4 *>(wsrc[i] + kt * BK);
            areg[i] = *reinterpret_cast<const uint4 *>(asrc[i] + kt * BK);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4 *>(&smem[buf][0][sdst[i]]) = wreg[i];
            *reinterpret_cast<uint4 *>(&smem[buf][1][sdst[i]]) = areg[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    gload(0);
    swrite(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t *>(&smem[buf][0][lds_off(wr * 64 + i * 16 + fr, ks * 4 + fq)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8_t *>(&smem[buf][1][lds_off(wc * 64 + j * 16 + fr, ks * 4 + fq)]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][r] = out[m = m0 + wc*64 + j*16 + fr][tile row = wr*64 + i*16 + fq*4 + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wc * 64 + j * 16 + fr;
        if (m >= g.M) continue;
        if (EPI == SCULPT_EPI_GEGLU) {
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int n = n0 + (wr * 2 + ip) * 16 + fq * 4;  // output column of r = 0
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[2 * ip][j][r] + (g.bias ? g.bias[n + r] : 0.f);
                    const float gt = acc[2 * ip + 1][j][r] + (g.bias ? g.bias[g.N + n + r] : 0.f);
                    o[r] = v * gelu_erf(gt);
                }
                if (g.out_bf16) {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                    *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                }
                if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wr * 64 + i * 16 + fq * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + (g.bias ? g.bias[n + r] : 0.f);
                    if (EPI == SCULPT_EPI_GELU) v = gelu_erf(v);
                    o[r] = v;
                }
                if (g.residual) {
                    const float4 rs = *reinterpret_cast<const float4 *>(g.residual + (long)m * g.ldr + n);
                    o[0] += rs.x; o[1] += rs.y; o[2] += rs.z; o[3] += rs.w;
                }
                if (g.out_f32) *reinterpret_cast<float4 *>(g.out_f32 + (long)m * g.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
                if (g.out_bf16) {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                    *reinterpret_cast<uint2 *>(g.out_bf16 + (long)m * g.ldo + n) = pk;
                }
                if (g.out_t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) g.out_t[(long)(n + r) * g.ldt + m] = f32_to_bf16(o[r]);
                }
            }
        }
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" int sculpt_gemm_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                                const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                                uint16_t *out_bf16_t, int ldt, int M, int N, int K, int epilogue,
                                sculpt_stream_t stream) {
    SC_REQUIRE(A && W, "gemm_bf16: null operand");
    SC_REQUIRE(out_f32 || out_bf16 || out_bf16_t, "gemm_bf16: no output");
    SC_REQUIRE(M >= 1 && N >= 1 && K >= BK, "gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    SC_REQUIRE(K % BK == 0, "gemm_bf16: K=%d must be a multiple of %d", K, BK);
    SC_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, "gemm_bf16: lda/ldw must be multiples of 8 elements (16-byte rows)");
    SC_REQUIRE(ldo % 4 == 0 && (!residual || ldr % 4 == 0), "gemm_bf16: ldo/ldr must be multiples of 4");
    GemmArgs g{A, lda, W, ldw, bias, residual, ldr, out_f32, out_bf16, ldo, out_bf16_t, ldt, M, N, K};
    const int mt = cdiv(M, BM);
    hipStream_t st = as_stream(stream);
    if (epilogue == SCULPT_EPI_GEGLU) {
        SC_REQUIRE(N % 64 == 0, "gemm_bf16(GEGLU): N=%d must be a multiple of 64", N);
        SC_REQUIRE(!residual && !out_bf16_t, "gemm_bf16(GEGLU): residual/transposed output unsupported");
        hipLaunchKernelGGL(gemm_bf16_kernel<SCULPT_EPI_GEGLU>, dim3(N / 64, mt), dim3(256), 0, st, g);
    } else {
        SC_REQUIRE(N % 128 == 0, "gemm_bf16: N=%d must be a multiple of 128", N);
        if (epilogue == SCULPT_EPI_GELU)
            hipLaunchKernelGGL(gemm_bf16_kernel<SCULPT_EPI_GELU>, dim3(N / 128, mt), dim3(256), 0, st, g);
        else if (epilogue == SCULPT_EPI_NONE)
            hipLaunchKernelGGL(gemm_bf16_kernel<SCULPT_EPI_NONE>, dim3(N / 128, mt), dim3(256), 0, st, g);
        else
            SC_REQUIRE(false, "gemm_bf16: unknown epilogue %d", epilogue);
    }
    SC_LAUNCH_CHECK();
    return 0;
}
