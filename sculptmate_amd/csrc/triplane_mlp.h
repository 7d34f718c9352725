// Device-side core of the fused triplane sample + NeRF-MLP kernels (csrc/triplane.hip, csrc/density_filter.hip): the packed
// decoder layout (sculpt_mlp_pack), SiLU, the LDS weight image, the limb splits and the k-step pipeline of the three-limb
// hidden layers.  Reference arithmetic: TripoSR/tsr/models/network_utils.py:116-124, nerf_renderer.py:41-91.
#pragma once
#include <math.h>

#include "common.h"

namespace sculpt {

static constexpr int HID = 64;          // hidden width (n_neurons)
static constexpr uint32_t PACK_MAGIC = 0x53434d33u;  // "SCM3": activations scaled by log2(e) (silu_f), third bf16 limb of the hidden weights

struct MlpPackHeader {
    uint32_t magic;
    int32_t K0;        // in_channels (3*C)
    int32_t NH;        // number of 64x64 hidden layers
    int32_t total_floats;
    int32_t off_w0raw; // [64][K0]
    int32_t off_b0raw; // [64]
    int32_t off_a0;    // [2][K0/2][64]
    int32_t off_bacc;  // [NH+1][2 h][2 t][16 r]
    int32_t off_hid;   // [NH][2 T][8 s4][64 lane][4]
    int32_t off_wlast; // [4][2 h][2 t][16 r]
    int32_t off_blast; // [4]
    int32_t off_x3;    // [NH][hi|lo][2 T][4 s][64 lane][8] bf16 (as 4096 floats per layer): split weights, bf16x3 mode
    int32_t off_x3h;   // same with fp16 halves (fp16x3 mode)
    int32_t off_w3;    // [NH][2 T][4 s][64 lane][8] bf16 (as 2048 floats per layer): third limb W - W1 - W2, bf16 3-limb mode
    int32_t pad[2];
};
static_assert(sizeof(MlpPackHeader) == 64, "header is 16 words");

__host__ __device__ __forceinline__ int nrow(int t, int r, int h) { return 32 * t + 8 * (r >> 2) + 4 * h + (r & 3); }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// Activations are carried SCALED by log2(e): with y = x * log2(e),
//     silu(x) * log2(e) = y / (1 + 2^-y)
// so v_exp_f32 (a base-2 exponential) takes -y directly (free source modifier) and the `x * -log2(e)` multiply of the
// textbook form disappears.  The scale is folded into the weights once, at pack time (sculpt_mlp_pack): layer 0 (weights
// and bias) x log2(e), every hidden bias x log2(e), the last layer's weights x ln(2); the hidden weights are untouched
// (W . (silu(x) log2e) + b log2e = log2e (W . silu(x) + b)).  Operands stay full fp32; the result differs from the unscaled
// evaluation by fp32 rounding only (tests/test_gpu_triplane.py tolerance unchanged).
__device__ __forceinline__ float silu_f(float y) {
    // y / (1 + 2^-y); v_exp_f32 and v_rcp_f32 are 1 ulp each
    const float e = __builtin_amdgcn_exp2f(-y);
    return y * __builtin_amdgcn_rcpf(1.0f + e);
}
typedef float tf32x2 __attribute__((ext_vector_type(2)));
// 16 SiLUs with the two full-rate operations on PAIRS (v_pk_add / v_pk_mul: bit-identical to the scalar forms): every VALU
// instruction, transcendental or not, takes the fp32 matrix pipe's issue slot for ~4 cycles (measured: replacing v_rcp by 7
// plain ops costs +8 %, sharing one v_rcp per pair at +3 plain ops costs +2 %), so the lever is the instruction COUNT:
// 5 (scalar textbook form) -> 3.5 (pairs) -> 3 per value (log2e folded into the weights).
__device__ __forceinline__ f32x16 silu16(f32x16 v) {
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const tf32x2 x = {v[i], v[i + 1]};
        const tf32x2 one = {1.0f, 1.0f};
        const tf32x2 e = {__builtin_amdgcn_exp2f(-x[0]), __builtin_amdgcn_exp2f(-x[1])};
        const tf32x2 d = e + one;
        const tf32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const tf32x2 y = x * r;
        o[i] = y[0]; o[i + 1] = y[1];
    }
    return o;
}

// the same with scalar full-rate operations: beside bf16 MFMAs the packed fp32 forms cost more issue time than the two scalar
// instructions they replace (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ f32x16 silu16_scalar(f32x16 v) {
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = silu_f(v[i]);
    return o;
}

// LDS image: [hid NH*4096*4 floats][bacc (NH+1)*64][wlast 256][blast 4]
struct LdsView {
    const float *hid, *bacc, *wlast, *blast;
};

__device__ __forceinline__ LdsView lds_view(float *smem, int NH) {
    LdsView v;
    v.hid = smem;
    v.bacc = smem + NH * 4096;
    v.wlast = v.bacc + (NH + 1) * 64;
    v.blast = v.wlast + 256;
    return v;
}

__host__ __device__ __forceinline__ int lds_floats_for(int NH) { return NH * 4096 + (NH + 1) * 64 + 256 + 4; }

__device__ __forceinline__ void load_weights_to_lds(float *smem, const float *blob, const MlpPackHeader &hd) {
    const int NH = hd.NH;
    const int nh4 = NH * 4096 / 4;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(blob + hd.off_hid);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    for (int i = threadIdx.x; i < nh4; i += blockDim.x) dst[i] = src[i];
    float *bacc = smem + NH * 4096;
    for (int i = threadIdx.x; i < (NH + 1) * 64; i += blockDim.x) bacc[i] = blob[hd.off_bacc + i];
    float *wl = bacc + (NH + 1) * 64;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) wl[i] = blob[hd.off_wlast + i];
    if (threadIdx.x < 4) wl[256 + threadIdx.x] = blob[hd.off_blast + threadIdx.x];
    __syncthreads();
}

__device__ __forceinline__ f32x16 lds_bias16(const float *bacc, int l, int h, int t) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(bacc + ((l * 2 + h) * 2 + t) * 16);
    f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3];
    o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
    return o;
}

// NH hidden layers: in/out = activations (post-SiLU) in accumulator layout.
__device__ __forceinline__ void hidden_layers(const LdsView &L, int NH, int lane, int h, f32x16 &x0, f32x16 &x1) {
    for (int l = 0; l < NH; ++l) {
        f32x16 acc0 = lds_bias16(L.bacc, l + 1, h, 0);
        f32x16 acc1 = lds_bias16(L.bacc, l + 1, h, 1);
        const f32x4 *A0 = reinterpret_cast<const f32x4 *>(L.hid) + ((l * 2 + 0) * 8) * 64 + lane;
        const f32x4 *A1 = reinterpret_cast<const f32x4 *>(L.hid) + ((l * 2 + 1) * 8) * 64 + lane;
        // (reading the A operands one group ahead in the source changes nothing measurable: the scheduler places the
        // ds_read_b128 pairs itself and four waves per SIMD cover the LDS latency)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            f32x4 a0 = A0[s4 * 64];
            f32x4 a1 = A1[s4 * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = s4 * 4 + j;
                const float b = (s < 16) ? x0[s & 15] : x1[s & 15];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b, acc1, 0, 0, 0);
            }
        }
        x0 = silu16(acc0);
        x1 = silu16(acc1);
    }
}

// last layer row o: partial dot over this lane's 32 neurons + other half
__device__ __forceinline__ float last_dot(const LdsView &L, int o, int h, const f32x16 &x0, const f32x16 &x1) {
    const float *w = L.wlast + (o * 2 + h) * 32;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(w[r], x0[r], s);
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(w[16 + r], x1[r], s);
    s += __shfl_xor(s, 32, 64);
    return s + L.blast[o];
}

__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }

// bilinear taps of torch grid_sample(align_corners=False, zeros padding) along one axis
struct Tap1 {
    int i0;       // floor index (may be -1 .. size-1)
    float w1;     // weight of i0+1 (fraction), weight of i0 is 1-w1
};
template <bool AC = false>
__device__ __forceinline__ Tap1 tap_of(float g, int size) {
    // torch grid_sampler_unnormalize: align_corners ? (g+1)/2*(size-1) : ((g+1)*size-1)/2
    float f = AC ? ((g + 1.0f) / 2.0f) * (float)(size - 1) : ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
    float fl = floorf(f);
    Tap1 t;
    t.i0 = (int)fl;
    t.w1 = f - fl;
    return t;
}
// scale_tensor(p, (-r, r), (-1, 1))  (nerf_renderer.py:52-54); true fp32 division like torch CPU
__device__ __forceinline__ float to_unit(float p, float radius, float span) {
    float d = (p - (-radius)) / span;
    return d * 2.0f + (-1.0f);
}

__device__ __forceinline__ void load_row32(const float *row, f32x16 &a, f32x16 &b) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(row);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 v = p[i];
        a[4 * i] = v[0]; a[4 * i + 1] = v[1]; a[4 * i + 2] = v[2]; a[4 * i + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 v = p[4 + i];
        b[4 * i] = v[0]; b[4 * i + 1] = v[1]; b[4 * i + 2] = v[2]; b[4 * i + 3] = v[3];
    }
}

typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    // v_cvt_pk_bf16_f32 (round to nearest even; a NaN stays a NaN); the builtin form, not inline asm, so that hipcc can
    // schedule it between MFMAs like any other vector instruction
    const tf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tbf16x2));
}

// x (16 fp32 accumulator values of one 32-neuron tile) -> two B-operand vectors per part: hi[2], lo[2]
__device__ __forceinline__ void split16(const f32x16 &x, tbf16x8 hi[2], tbf16x8 lo[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 ph, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = x[8 * q + 2 * i], b = x[8 * q + 2 * i + 1];
            const unsigned h2 = cvt_pk_bf16(a, b);
            const float ah = __uint_as_float(h2 << 16), bh = __uint_as_float(h2 & 0xffff0000u);
            ph[i] = h2;
            pl[i] = cvt_pk_bf16(a - ah, b - bh);
        }
        hi[q] = __builtin_bit_cast(tbf16x8, ph);
        lo[q] = __builtin_bit_cast(tbf16x8, pl);
    }
}
// the same with IEEE half parts (x - xh is exact in fp32; xl carries the next 11 bits).  Pairwise vector
// conversions: v_cvt_pk_f16_f32 (RNE), two v_cvt_f32_f16, one v_pk_add_f32 with a negated operand, v_cvt_pk_f16_f32.
__device__ __forceinline__ void split16(const f32x16 &x, tf16x8 hi[2], tf16x8 lo[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 ph, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const tf32x2 v = {x[8 * q + 2 * i], x[8 * q + 2 * i + 1]};
            const tf16x2 h2 = __builtin_convertvector(v, tf16x2);
            const tf32x2 back = __builtin_convertvector(h2, tf32x2);
            const tf16x2 l2 = __builtin_convertvector(v - back, tf16x2);
            ph[i] = __builtin_bit_cast(unsigned, h2);
            pl[i] = __builtin_bit_cast(unsigned, l2);
        }
        hi[q] = __builtin_bit_cast(tf16x8, ph);
        lo[q] = __builtin_bit_cast(tf16x8, pl);
    }
}
#ifdef SCULPT_L3_SHAPE_EXPERIMENT
// TIMING EXPERIMENT ONLY (wrong values): every 32x32x16 bf16 MFMA replaced by two 16x16x32 ones of the same FLOPs on the same
// operand registers -- the wall-clock effect of the small shape on this kernel's real instruction stream, before rewriting it
__device__ __forceinline__ f32x16 mfma16(tbf16x8 a, tbf16x8 b, f32x16 c) {
    typedef float f32x4e __attribute__((ext_vector_type(4)));
    f32x4e lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, hi, 0, 0, 0);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3]; c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
    return c;
}
#else
__device__ __forceinline__ f32x16 mfma16(tbf16x8 a, tbf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ f32x16 mfma16(tf16x8 a, tf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// x (16 fp32 accumulator values of one 32-neuron tile) -> B-operand vectors of its two k-steps, three limbs each
__device__ __forceinline__ void split16_l3(const f32x16 &x, tbf16x8 p1[2], tbf16x8 p2[2], tbf16x8 p3[2]) {
    // no contraction: x is y * rcp(..) of the inlined SiLU, and fma(y, r, -x1) would be the remainder of the EXACT product
    // (up to 24 bits of it) instead of the exact remainder of the fp32 value x, which the three limbs then could not hold
#pragma clang fp contract(off)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 v1, v2, v3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = x[8 * q + 2 * i], b = x[8 * q + 2 * i + 1];
            const unsigned h1 = cvt_pk_bf16(a, b);
            const float ra = a - __uint_as_float(h1 << 16), rb = b - __uint_as_float(h1 & 0xffff0000u);  // exact
            const unsigned h2 = cvt_pk_bf16(ra, rb);
            const float sa = ra - __uint_as_float(h2 << 16), sb = rb - __uint_as_float(h2 & 0xffff0000u);  // exact, <= 8 bits
            v1[i] = h1;
            v2[i] = h2;
            v3[i] = cvt_pk_bf16(sa, sb);
        }
        p1[q] = __builtin_bit_cast(tbf16x8, v1);
        p2[q] = __builtin_bit_cast(tbf16x8, v2);
        p3[q] = __builtin_bit_cast(tbf16x8, v3);
    }
}

struct VState {
    float x[8], t[8];
    unsigned h1[4], h2[4], h3[4];
};

// chunk C (0..11) of the SiLU + three-limb split of the 8 values in s.x; stage-major so that neighbours are independent.
// SPLIT = false (input of the last layer): SiLU only, s.x holds the activations afterwards.
template <int C, bool SPLIT>
__device__ __forceinline__ void vchunk(VState &s) {
#pragma clang fp contract(off)
    constexpr unsigned M = 0xffff0000u;
    if constexpr (C == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.t[i] = __builtin_amdgcn_exp2f(-s.x[i]);
    } else if constexpr (C == 1) {
#pragma unroll
        for (int i = 4; i < 8; ++i) s.t[i] = __builtin_amdgcn_exp2f(-s.x[i]);
    } else if constexpr (C == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.t[i] = 1.0f + s.t[i];
    } else if constexpr (C == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.t[i] = __builtin_amdgcn_rcpf(s.t[i]);
    } else if constexpr (C == 4) {
#pragma unroll
        for (int i = 4; i < 8; ++i) s.t[i] = __builtin_amdgcn_rcpf(s.t[i]);
    } else if constexpr (C == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.x[i] = s.x[i] * s.t[i];
    } else if constexpr (!SPLIT) {
    } else if constexpr (C == 6) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s.h1[q] = cvt_pk_bf16(s.x[2 * q], s.x[2 * q + 1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) { s.t[2 * q] = __uint_as_float(s.h1[q] << 16); s.t[2 * q + 1] = __uint_as_float(s.h1[q] & M); }
    } else if constexpr (C == 7) {
#pragma unroll
        for (int q = 2; q < 4; ++q) { s.t[2 * q] = __uint_as_float(s.h1[q] << 16); s.t[2 * q + 1] = __uint_as_float(s.h1[q] & M); }
#pragma unroll
        for (int i = 0; i < 4; ++i) s.x[i] = s.x[i] - s.t[i];
    } else if constexpr (C == 8) {
#pragma unroll
        for (int i = 4; i < 8; ++i) s.x[i] = s.x[i] - s.t[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) s.h2[q] = cvt_pk_bf16(s.x[2 * q], s.x[2 * q + 1]);
    } else if constexpr (C == 9) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { s.t[2 * q] = __uint_as_float(s.h2[q] << 16); s.t[2 * q + 1] = __uint_as_float(s.h2[q] & M); }
    } else if constexpr (C == 10) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.x[i] = s.x[i] - s.t[i];
    } else if constexpr (C == 11) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s.h3[q] = cvt_pk_bf16(s.x[2 * q], s.x[2 * q + 1]);
    }
}

struct Frags {  // A operands that cross a k-step boundary: W1 tile 0 (LDS) and the two W3 tiles (L2) of the NEXT k-step
    tbf16x8 a10, c0, c1;
};

__device__ __forceinline__ void l3_table_sum(const float *FA, const float *FB, const float *FC, int R, int ixl, int iy, int izc,
                                             int h, f32x16 &x0, f32x16 &x1) {
    f32x16 y0, y1;
    load_row32(FA + ((long)ixl * R + iy) * 64 + h * 32, x0, x1);
    load_row32(FB + ((long)ixl * R + izc) * 64 + h * 32, y0, y1);
    x0 += y0; x1 += y1;
    load_row32(FC + ((long)iy * R + izc) * 64 + h * 32, y0, y1);
    x0 += y0; x1 += y1;
}

// One k-step of ONE tile (the k-step software pipeline of density_grid_l3k_kernel): twelve MFMAs with the limbs p1 / p2 / p3
// of k-step g, and behind them (VALU = true) the SiLU + split chunks of the values already placed in `s` (k-step g + 1).
// MODE: 0 = MFMAs only, 1 = SiLU + split chunks behind them
template <int MODE>
__device__ __forceinline__ void l3_kstep(f32x16 &acc0, f32x16 &acc1, const tbf16x8 &p1, const tbf16x8 &p2, const tbf16x8 &p3,
                                         VState &s, Frags &f, const tbf16x8 *Ag, const tbf16x8 *An, const tbf16x8 *A3n) {
#define L3_SLOT(CH, ACC, AOP, BOP, PREFETCH)                 \
    ACC = mfma16(AOP, BOP, ACC);                             \
    if (MODE == 1) vchunk<CH, true>(s);                      \
    PREFETCH;                                                \
    __builtin_amdgcn_sched_barrier(0);
    tbf16x8 a11, a20, a21;
    Frags n;
    L3_SLOT(0, acc0, f.a10, p3, a11 = Ag[256])
    L3_SLOT(1, acc0, f.a10, p2, n.c0 = A3n[0])
    L3_SLOT(2, acc0, f.a10, p1, n.c1 = A3n[256])
    L3_SLOT(3, acc1, a11, p3, a20 = Ag[512])
    L3_SLOT(4, acc1, a11, p2, )
    L3_SLOT(5, acc1, a11, p1, a21 = Ag[768])
    L3_SLOT(6, acc0, a20, p2, )
    L3_SLOT(7, acc0, a20, p1, n.a10 = An[0])
    L3_SLOT(8, acc1, a21, p2, )
    L3_SLOT(9, acc1, a21, p1, )
    L3_SLOT(10, acc0, f.c0, p1, )
    L3_SLOT(11, acc1, f.c1, p1, )
    f = n;
#undef L3_SLOT
}

__device__ __forceinline__ void vstate_limbs(const VState &s, tbf16x8 &p1, tbf16x8 &p2, tbf16x8 &p3) {
    const u32x4 w1 = {s.h1[0], s.h1[1], s.h1[2], s.h1[3]}, w2 = {s.h2[0], s.h2[1], s.h2[2], s.h2[3]},
                w3 = {s.h3[0], s.h3[1], s.h3[2], s.h3[3]};
    p1 = __builtin_bit_cast(tbf16x8, w1);
    p2 = __builtin_bit_cast(tbf16x8, w2);
    p3 = __builtin_bit_cast(tbf16x8, w3);
}

// [W1 | W2: NH*4096 floats][bacc][wlast][blast]: the LDS image of the three-limb kernels
__device__ __forceinline__ void l3_load_lds(float *smem, const float *blob, const MlpPackHeader &hd) {
    const int NH = hd.NH;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(blob + hd.off_x3);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    for (int i = threadIdx.x; i < NH * 1024; i += blockDim.x) dst[i] = src[i];
    float *bacc = smem + NH * 4096;
    for (int i = threadIdx.x; i < (NH + 1) * 64; i += blockDim.x) bacc[i] = blob[hd.off_bacc + i];
    float *wl = bacc + (NH + 1) * 64;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) wl[i] = blob[hd.off_wlast + i];
    if (threadIdx.x < 4) wl[256 + threadIdx.x] = blob[hd.off_blast + threadIdx.x];
    __syncthreads();
}

// The NH hidden layers + the SiLU in front of the last layer for ONE tile of 32 points, k-step pipeline (density_grid_l3k_kernel,
// density_list_l3k_kernel): in x0 / x1 the layer-0 pre-activations (accumulator layout), out the activations the last layer reads.
// A: the W1 | W2 image in LDS (+ lane), A3: the third limbs in L2 (+ lane).  The value of a point depends on its own column only:
// the same point gives the same bits wherever it sits in whatever tile.
__device__ __forceinline__ void l3k_hidden(const LdsView &L, int NH, const tbf16x8 *A, const tbf16x8 *A3, int h, f32x16 &x0,
                                           f32x16 &x1) {
    Frags f;
    f.a10 = A[0];
    f.c0 = A3[0]; f.c1 = A3[256];
    for (int l = 0; l < NH; ++l) {
        const tbf16x8 *Al = A + (long)l * 1024, *A3l = A3 + (long)l * 512;
        const int ln = min(l + 1, NH - 1);  // the fetch behind the last k-step is never used; keep it inside the arrays
        f32x16 acc0 = lds_bias16(L.bacc, l + 1, h, 0);
        f32x16 acc1 = lds_bias16(L.bacc, l + 1, h, 1);
        tbf16x8 p1, p2, p3;
        {   // k-step 0's SiLU + split has no MFMAs of this tile to run behind
            VState s;
#pragma unroll
            for (int i = 0; i < 8; ++i) s.x[i] = x0[i];
            vchunk<0, true>(s); vchunk<1, true>(s); vchunk<2, true>(s); vchunk<3, true>(s); vchunk<4, true>(s); vchunk<5, true>(s);
            vchunk<6, true>(s); vchunk<7, true>(s); vchunk<8, true>(s); vchunk<9, true>(s); vchunk<10, true>(s); vchunk<11, true>(s);
            vstate_limbs(s, p1, p2, p3);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            VState s;
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s.x[i] = (g + 1 < 2 ? x0 : x1)[8 * ((g + 1) & 1) + i];
            }
            const tbf16x8 *Ag = Al + g * 64;
            const tbf16x8 *An = g < 3 ? Al + (g + 1) * 64 : A + (long)ln * 1024;
            const tbf16x8 *A3n = g < 3 ? A3l + (g + 1) * 64 : A3 + (long)ln * 512;
            if (g < 3) {
                l3_kstep<1>(acc0, acc1, p1, p2, p3, s, f, Ag, An, A3n);
                vstate_limbs(s, p1, p2, p3);
            } else {
                l3_kstep<0>(acc0, acc1, p1, p2, p3, s, f, Ag, An, A3n);
            }
        }
        x0 = acc0;
        x1 = acc1;
    }
    x0 = silu16_scalar(x0);
    x1 = silu16_scalar(x1);
}

}  // namespace sculpt
