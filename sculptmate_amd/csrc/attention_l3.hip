// Fused attention for the fast parity mode (TSR(precision="bf16l3")): fp32 Q / K / V^T in, fp32 O out, fp32 ARITHMETIC on the bf16
// matrix pipe -- both products, S = (c q) . k and O = P . v, take their operands split EXACTLY into three bf16 limbs and sum
// the six limb products of order >= 2^-16 in fp32 (gemm_l3.hip has the argument); the softmax (running maximum, exp2, row sums,
// rescale) is plain fp32 in registers.  Replaces the parity modes' three launches per attention (scores into an fp32
// [heads][Tq][Tk] scratch, row softmax, P V: 600 MB written and re-read three times per backbone self-attention) by one.
// Reference: F.scaled_dot_product_attention, TripoSR/tsr/models/transformer/attention.py:629-631 (fp32, no autocast); HF
// ViTSelfAttention's eager softmax(QK^T / 8) V.
//
// One workgroup = 4 waves = 128 queries of one head; a wave owns 32 queries (the query is the MFMA column, lane & 31, so a
// lane holds 2 x 16 scores of ONE query per 64-key tile and the softmax statistics are in-register + one cross-half exchange).
// Per 64-key tile:  K rows and V^T rows arrive as fp32 through registers (loaded one tile ahead), are split and written as
// limbs into a [limb][8-wide k chunk][row] LDS image (gemm_l3.hip's: a fragment is 512 contiguous bytes);
//   S^T = K . Q^T     48 MFMAs (2 row tiles x 4 k-steps x 6 limb products); the Q limbs stay in registers for the whole loop;
//                     key row m of a 32-row tile sits at LDS row m with bits 2 and 3 swapped, so that a lane's accumulator
//                     registers 8 g .. 8 g + 7 hold 8 CONTIGUOUS keys -- exactly one B fragment of the second product;
//   P = exp2(S - M)   fp32; split into three limbs in registers (no LDS round trip);
//   O^T += V^T . P^T  48 MFMAs.
// One 49-KiB LDS buffer, two barriers per tile; two workgroups per CU overlap one's split / softmax with the other's MFMAs.
#include "attention_tile.h"

namespace sculpt {

typedef __bf16 albf16x2 __attribute__((ext_vector_type(2)));
typedef float alf32x2 __attribute__((ext_vector_type(2)));

static constexpr int AL_CS = 64 * 16 + 16;    // bytes from one k-chunk plane (64 rows x 16 B) to the next (+16: the 8-byte writes
                                              // of a 16-lane group land on 16 different 8-byte slots of the 128-byte bank row)
static constexpr int AL_LT = 8 * AL_CS;       // one limb of one operand tile (64 rows x 64 k)
static constexpr int AL_OP = 3 * AL_LT;       // one operand tile, three limbs

__device__ __forceinline__ unsigned al_cvt_pk(float lo, float hi) {
    const alf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, albf16x2));
}

// two fp32 values -> their three packed limb pairs
__device__ __forceinline__ void al_split2(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
#pragma clang fp contract(off)
    p1 = al_cvt_pk(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);   // exact
    p2 = al_cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u); // exact, <= 8 bits
    p3 = al_cvt_pk(sa, sb);
}

typedef unsigned alu32x4 __attribute__((ext_vector_type(4)));

// eight fp32 values -> one MFMA operand fragment per limb
__device__ __forceinline__ void al_split8(const float (&x)[8], abf16x8 &f1, abf16x8 &f2, abf16x8 &f3) {
    alu32x4 v1, v2, v3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a, b, c;
        al_split2(x[2 * i], x[2 * i + 1], a, b, c);
        v1[i] = a; v2[i] = b; v3[i] = c;
    }
    f1 = __builtin_bit_cast(abf16x8, v1);
    f2 = __builtin_bit_cast(abf16x8, v2);
    f3 = __builtin_bit_cast(abf16x8, v3);
}

__global__ __launch_bounds__(256, 2) void attention_l3_kernel(const float *__restrict__ Q, int ldq, const float *__restrict__ K, int ldk,
                                                              const float *__restrict__ Vt, int ldvt, float *__restrict__ O, int ldo,
                                                              int Tq, int Tk, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AL_OP];   // [K limbs | V^T limbs]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + qc;
    const int qld = min(q, Tq - 1);

    // Q limbs (B operand of the first product: B[k = 8 h + j][column = query]), scaled by softmax_scale * log2(e) in fp32
    abf16x8 qf[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 a = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8);
        const float4 b = *reinterpret_cast<const float4 *>(Q + (long)qld * ldq + head * 64 + ks * 16 + h * 8 + 4);
        const float x[8] = {a.x * scale_log2e, a.y * scale_log2e, a.z * scale_log2e, a.w * scale_log2e,
                            b.x * scale_log2e, b.y * scale_log2e, b.z * scale_log2e, b.w * scale_log2e};
        al_split8(x, qf[ks][0], qf[ks][1], qf[ks][2]);
    }

    // staging: a tile is 64 rows x 64 fp32 = 1024 float4 per operand; thread t takes quad t % 16 of rows t / 16 + 16 i
    const int sr = tid >> 4, sq = tid & 15;
    const float *Kh = K + head * 64 + 4 * sq;                       // + key * ldk
    const float *Vh = Vt + (long)(head * 64 + sr) * ldvt + 4 * sq;  // + 16 i * ldvt + key0
    // LDS byte offsets of this thread's 8-byte pieces: chunk = sq / 2, half = sq % 2; K rows go to their permuted position
    int kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = sr + 16 * i, m = kk & 31;
        const int pos = (kk & 32) | (m & 0x13) | ((m & 4) << 1) | ((m & 8) >> 1);
        kofs[i] = (sq >> 1) * AL_CS + pos * 16 + (sq & 1) * 8;
    }
    const int vofs = AL_OP + (sq >> 1) * AL_CS + sr * 16 + (sq & 1) * 8;   // + 16 i rows = + 256 i bytes
    // fragment read offsets: k-step ks reads chunk 2 ks + h, rows 32 t + (lane & 31)
    const int fro = h * AL_CS + qc * 16;

    const int nt = (Tk + 63) / 64;
    float4 rk[4], rv[4];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(t * 64 + sr + 16 * i, Tk - 1);
            rk[i] = *reinterpret_cast<const float4 *>(Kh + (long)key * ldk);
            rv[i] = *reinterpret_cast<const float4 *>(Vh + (long)(16 * i) * ldvt + t * 64);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    gload(0);
    for (int t = 0; t < nt; ++t) {
        if (t > 0) __syncthreads();   // every wave has read the previous tile's fragments
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned a1, a2, a3, b1, b2, b3;
            al_split2(rk[i].x, rk[i].y, a1, a2, a3);
            al_split2(rk[i].z, rk[i].w, b1, b2, b3);
            unsigned char *d = smem + kofs[i];
            *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(d + AL_LT) = make_uint2(a2, b2);
            *reinterpret_cast<uint2 *>(d + 2 * AL_LT) = make_uint2(a3, b3);
            al_split2(rv[i].x, rv[i].y, a1, a2, a3);
            al_split2(rv[i].z, rv[i].w, b1, b2, b3);
            d = smem + vofs + i * 256;
            *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(d + AL_LT) = make_uint2(a2, b2);
            *reinterpret_cast<uint2 *>(d + 2 * AL_LT) = make_uint2(a3, b3);
        }
        __syncthreads();
        if (t + 1 < nt) gload(t + 1);   // the next tile travels while this one is multiplied

        // ---- S^T = K . Q^T (exponents of 2)
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            abf16x8 k0[3], k1[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                k0[l] = *reinterpret_cast<const abf16x8 *>(smem + fro + l * AL_LT + 2 * ks * AL_CS);
                k1[l] = *reinterpret_cast<const abf16x8 *>(smem + fro + l * AL_LT + 2 * ks * AL_CS + 32 * 16);
            }
#define AL_SIX(acc, a, b)                                                           \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0)
            AL_SIX(s0, k0, qf[ks]);
            AL_SIX(s1, k1, qf[ks]);
        }
        // lane (query, h): s{rt}[8 g + j] = key 64 t + 32 rt + 16 g + 8 h + j
        if (t * 64 + 64 > Tk) {   // ragged last tile (wave-uniform)
            const int kb = t * 64 + 8 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb + (r & 7) + 16 * (r >> 3);
                if (key >= Tk) s0[r] = -INFINITY;
                if (key + 32 >= Tk) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);          // finite from the first tile on: every tile has at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            ps += s0[r] + s1[r];
        }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T . P^T: k-step kstep = 2 rt + g takes P registers 8 g .. 8 g + 7 of s{rt} (contiguous keys)
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (kstep < 2) ? s0[8 * (kstep & 1) + j] : s1[8 * (kstep & 1) + j];
            abf16x8 p[3], v0[3], v1[3];
            al_split8(x, p[0], p[1], p[2]);
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                v0[l] = *reinterpret_cast<const abf16x8 *>(smem + AL_OP + fro + l * AL_LT + 2 * kstep * AL_CS);
                v1[l] = *reinterpret_cast<const abf16x8 *>(smem + AL_OP + fro + l * AL_LT + 2 * kstep * AL_CS + 32 * 16);
            }
            AL_SIX(o0, v0, p);
            AL_SIX(o1, v1, p);
        }
#undef AL_SIX
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Tq) {
        float *orow = O + (long)q * ldo + head * 64;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {   // registers 4 g4 .. 4 g4 + 3 = d 8 g4 + 4 h + {0..3} (+ 32 for o1)
            *reinterpret_cast<float4 *>(orow + 8 * g4 + 4 * h) =
                make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + 8 * g4 + 4 * h) =
                make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
        }
    }
}

}  // namespace sculpt

using namespace sculpt;

extern "C" int sculpt_attention_f32_l3(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, float *O, int ldo,
                                       int Tq, int Tk, int heads, float scale, sculpt_stream_t stream) {
    SC_REQUIRE(Q && K && Vt && O, "attention_f32_l3: null argument");
    SC_REQUIRE(Tq >= 1 && Tk >= 1 && heads >= 1 && heads <= 65535, "attention_f32_l3: bad shape Tq=%d Tk=%d heads=%d", Tq, Tk, heads);
    SC_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldvt % 4 == 0 && ldo % 4 == 0, "attention_f32_l3: row strides must keep 16-byte alignment");
    SC_REQUIRE(ldvt >= ((Tk + 63) / 64) * 64, "attention_f32_l3: ldvt=%d must be >= round_up(Tk=%d, 64) (finite padding columns)", ldvt, Tk);
    SC_REQUIRE(scale > 0.f && scale == scale, "attention_f32_l3: scale must be positive");
    hipLaunchKernelGGL(attention_l3_kernel, dim3(cdiv(Tq, 128), heads), dim3(256), 0, as_stream(stream), Q, ldq, K, ldk, Vt, ldvt, O,
                       ldo, Tq, Tk, scale * 1.44269504088896340736f);
    SC_LAUNCH_CHECK();
    return 0;
}
