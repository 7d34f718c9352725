// microbenchmark: which bf16 MFMA shape is faster BY WALL CLOCK in a loop shaped like the density kernel's layer -- the chip holds
// its clock down under matrix load, and the clock it holds can depend on the shape (MI355X_MICROARCH.md, DVFS give-back item 7).
// Every wave owns a 64 (neurons) x 32 (points) fp32 output tile and runs, per iteration, one K = 32 slice of one limb product:
//   shape A  v_mfma_f32_32x32x16_bf16:  2 row blocks x 2 k-steps            = 4 MFMAs of 32 cycles, 4 A fragments from LDS
//   shape B  v_mfma_f32_16x16x32_bf16:  4 row blocks x 2 point blocks x 1   = 8 MFMAs of 16 cycles, 4 A fragments from LDS
// (each B-shape fragment feeds two MFMAs: the LDS bytes per FLOP are the same), B operands in registers, random operand bits,
// plus FILL vector instructions per iteration spread behind the MFMAs (v_exp_f32 : plain = 1 : 2 like the SiLU + limb split).
// Reports wall TFLOP/s over >= 1 s of back-to-back launches, shader cycles per iteration (s_memtime) and the in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz), for 1 / 2 / 4 waves per SIMD on all 256 CUs.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ITER 2048
#define FENCE __builtin_amdgcn_sched_barrier(0)

// one "plain" and one transcendental filler on independent registers
#define PLAIN(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(j) & 7]) : "v"(z), "v"(z2))  /* stays in [-1.5, 0) */
#define TRANS(j) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(j) & 7]))

// FILL = number of filler GROUPS per iteration (a group = 1 v_exp + 2 v_fma = 16 issue cycles); slots = MFMAs per iteration
template <int FILL, int SLOTS, int S>
__device__ __forceinline__ void fillers(float (&r)[8], float z, float z2) {
    // group g goes behind MFMA slot (g * SLOTS / FILL)
#pragma unroll
    for (int g = 0; g < FILL; ++g)
        if (g * SLOTS / (FILL > 0 ? FILL : 1) == S) { TRANS(3 * g); PLAIN(3 * g + 1); PLAIN(3 * g + 2); }
}

template <int SHAPE, int FILL>
__global__ __launch_bounds__(1024) void k(const uint4 *__restrict__ wsrc, unsigned long long *stamps, float *sink) {
    __shared__ uint4 lds[64 * 64];  // 64 fragments of 1 KiB
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = wsrc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 b0, b1;
    {
        const uint4 u0 = wsrc[(threadIdx.x * 7 + 1) & 4095], u1 = wsrc[(threadIdx.x * 13 + 5) & 4095];
        b0 = __builtin_bit_cast(bf16x8, u0);
        b1 = __builtin_bit_cast(bf16x8, u1);
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = -1.0f - 0.001f * (threadIdx.x + j);
    const float z = -0.5f, z2 = -1.0f;
    f32x16 ca = {}, cb = {};
    f32x4 t[8] = {};
    const uint4 *frag = lds + lane;
    uint4 f0 = frag[0], f1 = frag[64], f2 = frag[128], f3 = frag[192];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; ++i) {
        const uint4 *nx = frag + (((i + 1) & 15) * 4) * 64;
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, f0), a1 = __builtin_bit_cast(bf16x8, f1);
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, f2), a3 = __builtin_bit_cast(bf16x8, f3);
        FENCE;
        if (SHAPE == 0) {
            ca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, ca, 0, 0, 0);
            f0 = nx[0];
            fillers<FILL, 4, 0>(r, z, z2);
            FENCE;
            cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, cb, 0, 0, 0);
            f1 = nx[64];
            fillers<FILL, 4, 1>(r, z, z2);
            FENCE;
            ca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, ca, 0, 0, 0);
            f2 = nx[128];
            fillers<FILL, 4, 2>(r, z, z2);
            FENCE;
            cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, cb, 0, 0, 0);
            f3 = nx[192];
            fillers<FILL, 4, 3>(r, z, z2);
            FENCE;
        } else {
            t[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, t[0], 0, 0, 0);
            fillers<FILL, 8, 0>(r, z, z2);
            FENCE;
            t[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, t[1], 0, 0, 0);
            f0 = nx[0];
            fillers<FILL, 8, 1>(r, z, z2);
            FENCE;
            t[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, t[2], 0, 0, 0);
            fillers<FILL, 8, 2>(r, z, z2);
            FENCE;
            t[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, t[3], 0, 0, 0);
            f1 = nx[64];
            fillers<FILL, 8, 3>(r, z, z2);
            FENCE;
            t[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, t[4], 0, 0, 0);
            fillers<FILL, 8, 4>(r, z, z2);
            FENCE;
            t[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, t[5], 0, 0, 0);
            f2 = nx[128];
            fillers<FILL, 8, 5>(r, z, z2);
            FENCE;
            t[6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b0, t[6], 0, 0, 0);
            fillers<FILL, 8, 6>(r, z, z2);
            FENCE;
            t[7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, t[7], 0, 0, 0);
            f3 = nx[192];
            fillers<FILL, 8, 7>(r, z, z2);
            FENCE;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    float acc = ca[0] + cb[1];
    for (int j = 0; j < 8; ++j) acc += r[j] + t[j][j & 3];
    if (acc == 12345.678f) sink[0] = acc;
    if (lane == 0) {  // stamps go to a buffer of their own
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

template <int SHAPE, int FILL>
static void run(int waves_per_simd, const uint4 *w, unsigned long long *st, float *sink) {
    const int threads = 256 * waves_per_simd, nw = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // warm the clock governor: ~1 s of back-to-back launches, then time the last 50
    float ms = 0;
    int warm = 0;
    for (;;) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<SHAPE, FILL>), dim3(256), dim3(threads), 0, 0, w, st, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1);
        ms += m; ++warm;
        if (ms > 1000.f && warm >= 2) { ms = m; break; }
    }
    std::vector<unsigned long long> h(2 * nw);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 100e6; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 64 * 32 * 32 * (double)ITER * nw;  // per launch
    // (cycles: one wave's own stamps; with several waves per SIMD the oldest finishes early, so only w = 1 reads as cycles per SIMD)
    printf("  shape %s fill %2d w=%d: %7.1f TFLOP/s by wall | %6.1f cycles/iteration of one wave (128 = matrix pipe at w=1) | clock %.2f GHz\n",
           SHAPE ? "16x16x32" : "32x32x16", FILL, waves_per_simd, flop * 50 / (ms * 1e-3) / 1e12, cyc[nw / 2] / ITER, clk[nw / 2] / 1e9);
    fflush(stdout);
}

int main() {
    uint4 *w; unsigned long long *st; float *sink;
    hipMalloc(&w, 4096 * sizeof(uint4)); hipMalloc(&st, sizeof(unsigned long long) * 2 * 256 * 16); hipMalloc(&sink, 4);
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto &x : h) {  // random bf16 in +-[0.25, 2): random mantissa and sign, a few exponents
        const unsigned s = rand() & 1, e = 125 + rand() % 3, m = rand() & 127;
        x = (unsigned short)((s << 15) | (e << 7) | m);
    }
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 4}) {
        printf("%d wave(s) per SIMD\n", wps);
        run<0, 0>(wps, w, st, sink); run<1, 0>(wps, w, st, sink);
        run<0, 4>(wps, w, st, sink); run<1, 4>(wps, w, st, sink);
        run<0, 8>(wps, w, st, sink); run<1, 8>(wps, w, st, sink);
    }
    return 0;
}
