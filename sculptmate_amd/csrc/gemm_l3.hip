// fp32-in / fp32-out GEMM with fp32 ARITHMETIC on the bf16 matrix pipe: both operands are split EXACTLY into three bf16 limbs
// while they are staged into LDS, and the six limb products of order >= 2^-16 are accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16 -- the density kernel's technique (triplane.hip) carried to the Linears and the
// per-head attention products of the transformer stack.
//
//   x = x1 + x2 + x3,  W = W1 + W2 + W3      each limb = round-to-nearest bf16 of the exact remainder; a 24-bit significand
//                                            minus two 8-bit limbs leaves <= 8 bits, so the third limb is exact; bf16 has the
//                                            fp32 exponent: no range limit
//   W.x = W1x3 + W3x1 + W2x2 + W1x2 + W2x1 + W1x1   (+ W2x3 + W3x2 + W3x3 < 2^-23 |W||x|, dropped)
// Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32: this is the reference's fp32 Linear
// (TripoSR/tsr/models/transformer/attention.py:569-653, basic_transformer_block.py:291-315; no autocast, generate.py:36-39)
// with 24-bit operands, at up to 16 / 6 = 2.7x the exact-fp32 matrix pipe's rate.
//
// Same interface, tiling (128 x 128 outputs, 2 x 2 waves of 2 x 2 accumulator tiles of 32 x 32) and epilogue as gemm_f32.hip.
// K-step 32.  Staging is through registers -- an LDS-DMA cannot convert -- : a thread loads 8 float4 (4 rows x 4 consecutive k
// of each operand) one K-step ahead, splits them (5.5 vector instructions per value: v_cvt_pk_bf16_f32, two expands, two exact
// subtractions) and writes the limbs as 8-byte pieces into a [limb][k-chunk of 8][row] image: a fragment (32 rows x 8 k) is
// 512 contiguous bytes, so every ds_read_b128 lane group covers all 16 slots of the bank row, and the chunk planes are
// 32 bytes off the bank period so the 8-byte writes of a 16-lane group do not collide either.  One 49-KiB buffer per workgroup
// (two barriers per K-step); up to three workgroups per CU overlap one's split / write phase with another's MFMAs.
#include "gemm_f32.h"

namespace sculpt {

typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lbf16x2 __attribute__((ext_vector_type(2)));
typedef float lf32x2 __attribute__((ext_vector_type(2)));

static constexpr int L3_BK = 32;
static constexpr int L3_CS = 128 * 16 + 32;   // bytes from one k-chunk plane (128 rows x 16 B) to the next
static constexpr int L3_LT = 4 * L3_CS;       // one limb of one operand tile
static constexpr int L3_OP = 3 * L3_LT;       // one operand tile, three limbs

__device__ __forceinline__ unsigned l3_cvt_pk(float lo, float hi) {
    const lf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, lbf16x2));
}

// four consecutive-k fp32 values -> the 8-byte piece of each limb plane
__device__ __forceinline__ void l3_split4(const float4 &x, uint2 &p1, uint2 &p2, uint2 &p3) {
#pragma clang fp contract(off)
    const unsigned a1 = l3_cvt_pk(x.x, x.y), b1 = l3_cvt_pk(x.z, x.w);
    const float r0 = x.x - __uint_as_float(a1 << 16), r1 = x.y - __uint_as_float(a1 & 0xffff0000u);   // exact
    const float r2 = x.z - __uint_as_float(b1 << 16), r3 = x.w - __uint_as_float(b1 & 0xffff0000u);
    const unsigned a2 = l3_cvt_pk(r0, r1), b2 = l3_cvt_pk(r2, r3);
    const float s0 = r0 - __uint_as_float(a2 << 16), s1 = r1 - __uint_as_float(a2 & 0xffff0000u);     // exact, <= 8 bits
    const float s2 = r2 - __uint_as_float(b2 << 16), s3 = r3 - __uint_as_float(b2 & 0xffff0000u);
    p1 = make_uint2(a1, b1);
    p2 = make_uint2(a2, b2);
    p3 = make_uint2(l3_cvt_pk(s0, s1), l3_cvt_pk(s2, s3));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_l3_kernel(GemmF32Args g_in) {
    const GemmF32Args g = f32_batch_entry(g_in);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * L3_OP];   // [W limbs | A limbs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int NOUT = (EPI == SCULPT_EPI_GEGLU) ? FBW / 2 : FBW;
    const int n0 = blockIdx.x * NOUT, m0 = blockIdx.y * FBM;

    // staging: a tile is 128 rows x 32 k = 1024 float4; thread t takes k-quad t % 8 of rows t / 8 + 32 i (a row's 128 bytes are
    // one cache line read by 8 neighbouring lanes)
    const int sr = tid >> 3, kq = tid & 7;
    const float *wp[4], *ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wp[i] = g.W + (long)f32_tile_wrow<EPI>(g, n0, sr + 32 * i) * g.ldw + 4 * kq;
        ap[i] = g.A + (long)min(m0 + sr + 32 * i, g.M - 1) * g.lda + 4 * kq;
    }
    // LDS byte offset of this thread's 8-byte piece of row sr in a limb plane (chunk = kq / 2, half = kq % 2)
    const int wofs = (kq >> 1) * L3_CS + sr * 16 + (kq & 1) * 8;
    float4 rw[4], ra[4];
    const int nk = g.K / L3_BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // fragment read offsets: k-step s reads chunk 2 s + lh; rows wr * 64 + i * 32 + l31 (W) / wc * 64 + j * 32 + l31 (A)
    const int wfo = lh * L3_CS + (wr * 64 + l31) * 16;
    const int afo = L3_OP + lh * L3_CS + (wc * 64 + l31) * 16;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rw[i] = *reinterpret_cast<const float4 *>(wp[i]);
        ra[i] = *reinterpret_cast<const float4 *>(ap[i]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();   // every wave has read the previous K-step's fragments
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 p1, p2, p3;
            l3_split4(rw[i], p1, p2, p3);
            unsigned char *d = smem + wofs + i * (32 * 16);
            *reinterpret_cast<uint2 *>(d) = p1;
            *reinterpret_cast<uint2 *>(d + L3_LT) = p2;
            *reinterpret_cast<uint2 *>(d + 2 * L3_LT) = p3;
            l3_split4(ra[i], p1, p2, p3);
            d += L3_OP;
            *reinterpret_cast<uint2 *>(d) = p1;
            *reinterpret_cast<uint2 *>(d + L3_LT) = p2;
            *reinterpret_cast<uint2 *>(d + 2 * L3_LT) = p3;
        }
        __syncthreads();
        if (kt + 1 < nk) {   // the next K-step's operands travel while this one is multiplied
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rw[i] = *reinterpret_cast<const float4 *>(wp[i] + (kt + 1) * L3_BK);
                ra[i] = *reinterpret_cast<const float4 *>(ap[i] + (kt + 1) * L3_BK);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            lbf16x8 wf[2][3], af[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    wf[i][l] = *reinterpret_cast<const lbf16x8 *>(smem + wfo + l * L3_LT + 2 * s * L3_CS + i * (32 * 16));
                    af[i][l] = *reinterpret_cast<const lbf16x8 *>(smem + afo + l * L3_LT + 2 * s * L3_CS + i * (32 * 16));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];   // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][2], af[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], af[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], af[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], af[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    }
    f32_tile_epilogue<EPI>(g, acc, n0, m0, wr, wc, l31, lh);
}

int gemm_l3_launch(const GemmF32Args &g, int epilogue, int batch, hipStream_t st) {
    const int mt = cdiv(g.M, FBM);
    if (epilogue == SCULPT_EPI_GEGLU) hipLaunchKernelGGL(gemm_l3_kernel<SCULPT_EPI_GEGLU>, dim3(g.N / 64, mt, batch), dim3(256), 0, st, g);
    else if (epilogue == SCULPT_EPI_GELU) hipLaunchKernelGGL(gemm_l3_kernel<SCULPT_EPI_GELU>, dim3(cdiv(g.N, FBW), mt, batch), dim3(256), 0, st, g);
    else if (epilogue == SCULPT_EPI_RELU) hipLaunchKernelGGL(gemm_l3_kernel<SCULPT_EPI_RELU>, dim3(cdiv(g.N, FBW), mt, batch), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_l3_kernel<SCULPT_EPI_NONE>, dim3(cdiv(g.N, FBW), mt, batch), dim3(256), 0, st, g);
    return 0;
}

}  // namespace sculpt
