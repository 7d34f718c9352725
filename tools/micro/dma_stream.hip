// L2 -> LDS streaming rate of a GEMM-shaped workgroup, with and without the tile's LDS reads + MFMAs (gfx950).
// Each workgroup walks K = 1024 in 64-wide steps for one (A rows, W rows) tile of an FF1-sized problem (3072 x 8192 x 1024 bf16),
// staging BM + BN rows of 128 B per step with global_load_lds_dwordx4 into an NSTAGE ring (counted vmcnt, one raw barrier per
// step) -- the staging skeleton of csrc/gemm.hip.  COMPUTE adds, per step and wave, NREAD ds_read_b128 + NMFMA
// v_mfma_f32_16x16x32_bf16.  Prints GB/s per CU.   hipcc -O3 --offload-arch=gfx950 -o dma_stream dma_stream.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_tile(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// ORDER: how consecutive tiles of an XCD's chunk map to the tile grid -- 0: n fastest (one A panel, every W panel per 64
// tiles), 1: m fastest (every A panel, few W panels), 2: 8 x 8 blocks (8 + 8 panels per 64 tiles)
template <int BM, int BN, int NW, int NSTAGE, int NREAD, int NMFMA, int ORDER = 0, int BLK_M = 8, int BLK_N = 8>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const uint16_t *A, const uint16_t *W, int K, int nt_n, float *sink) {
    constexpr int ROWS = BM + BN, STAGE = ROWS * 128, PIECES = ROWS / 8 / NW, DIST = NSTAGE - 1;
    static_assert(PIECES * 8 * NW == ROWS, "rows must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int nt_m = gridDim.x / nt_n;
    int mt, nt;
    if (ORDER == 0) { mt = tile / nt_n; nt = tile % nt_n; }
    else if (ORDER == 1) { nt = tile / nt_m; mt = tile % nt_m; }
    else { const int blk = tile / (BLK_M * BLK_N), in = tile % (BLK_M * BLK_N), bpr = nt_n / BLK_N; mt = (blk / bpr) * BLK_M + in / BLK_N; nt = (blk % bpr) * BLK_N + in % BLK_N; }
    const int m0 = mt * BM, n0 = nt * BN;
    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t *src[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int r = 8 * (wave * PIECES + q) + srow;  // tile row: first BM rows from A, then BN rows from W
        src[q] = (r < BM ? A + (long)(m0 + r) * K : W + (long)(n0 + r - BM) * K) + (sslot << 3);
    }
    const int nk = K / 64;
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[q] + kt * 64), (lds_ptr_t)(smem + buf * STAGE + (wave * PIECES + q) * 1024), 16, 0, 0);
    };
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int s = 0; s < DIST && s < nk; ++s) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        if (DIST > 1 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DIST - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + DIST < nk) stage((kt + DIST) % NSTAGE, kt + DIST);
        if (NREAD > 0) {
            const unsigned char *b = smem + (kt % NSTAGE) * STAGE;
            constexpr int NR = NREAD > 0 ? NREAD : 1;
            bf16x8_t fr[NR];
#pragma unroll
            for (int i = 0; i < NREAD; ++i) fr[i] = *reinterpret_cast<const bf16x8_t *>(b + ((wave * 8 + i) * 1024 + lane * 16) % STAGE);
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i % NR], fr[(i + 1) % NR], acc[i & 3], 0, 0, 0);
        }
    }
    if (sink && acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[threadIdx.x] = acc[0][0];
}

template <int BM, int BN, int NW, int NSTAGE, int NREAD, int NMFMA, int ORDER = 0, int BLK_M = 8, int BLK_N = 8>
void run(const char *name, const uint16_t *A, const uint16_t *W, float *sink, int wg_per_cu_hint) {
    const int M = 3072, N = 8192, K = 1024;
    const int tiles = (M / BM) * (N / BN);
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    if (ORDER == 2 && ((M / BM) % BLK_M || (N / BN) % BLK_N)) { printf("%s: block does not divide the grid\n", name); return; }
    auto k = stream_kernel<BM, BN, NW, NSTAGE, NREAD, NMFMA, ORDER, BLK_M, BLK_N>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NW * 64), lds, 0, A, W, K, N / BN, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NW * 64), lds, 0, A, W, K, N / BN, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, bytes = (double)tiles * (K / 64) * (BM + BN) * 128;
    printf("%-44s tiles %5d  lds %6zu B (%d wg/cu)  %7.1f us  %6.1f MB  %5.1f GB/s per CU  %5.2f TB/s\n", name, tiles, lds, wg_per_cu_hint, us,
           bytes / 1e6, bytes / us / 1e3 / 256, bytes / us / 1e6);
}

// The same walk with the ring cut into HALF-tile slots (A rows of a step, then W rows of a step, alternating): NSLOT slots of
// BM * 128 B (BM == BN).  Step kt reads loads 2kt and 2kt+1; after its barrier the slots of step kt-1 are free and loads
// 2kt+NSLOT-2 and 2kt+NSLOT-1 are issued, so NSLOT-2 half-tiles stay in flight while two are being read.
template <int BM, int NW, int NSLOT, int NREAD, int NMFMA>
__global__ __launch_bounds__(NW * 64) void stream_half_kernel(const uint16_t *A, const uint16_t *W, int K, int nt_n, float *sink) {
    constexpr int SLOT = BM * 128, PIECES = BM / 8 / NW;
    static_assert(PIECES * 8 * NW == BM, "rows must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int nt_m = gridDim.x / nt_n;
    const int m0 = (tile % nt_m) * BM, n0 = (tile / nt_m) * BM;  // m fastest
    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t *srcA[PIECES], *srcW[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int r = 8 * (wave * PIECES + q) + srow;
        srcA[q] = A + (long)(m0 + r) * K + (sslot << 3);
        srcW[q] = W + (long)(n0 + r) * K + (sslot << 3);
    }
    const int nk = K / 64, nl = 2 * nk;
    auto load = [&](int j) {  // half-tile j: A of step j/2 (even) or W of step j/2 (odd), into slot j % NSLOT
        const int kt = j >> 1, slot = j % NSLOT;
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(((j & 1) ? srcW[q] : srcA[q]) + kt * 64),
                                             (lds_ptr_t)(smem + slot * SLOT + (wave * PIECES + q) * 1024), 16, 0, 0);
    };
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int j = 0; j < NSLOT - 2 && j < nl; ++j) load(j);  // loads 0 .. NSLOT-3 (the loop issues two more per step)
    for (int kt = 0; kt < nk; ++kt) {
        // need loads <= 2kt+1; loads 2kt+2 .. 2kt+NSLOT-3 were issued after them and may stay in flight
        const int later = min(nl, 2 * kt + NSLOT - 2) - (2 * kt + 2);
        if (later >= NSLOT - 4 && NSLOT > 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NSLOT - 4)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (2 * kt + NSLOT - 2 < nl) load(2 * kt + NSLOT - 2);
        if (2 * kt + NSLOT - 1 < nl) load(2 * kt + NSLOT - 1);
        if (NREAD > 0) {
            const unsigned char *b = smem + ((2 * kt) % NSLOT) * SLOT, *c = smem + ((2 * kt + 1) % NSLOT) * SLOT;
            constexpr int NR = NREAD > 0 ? NREAD : 1;
            bf16x8_t fr[NR];
#pragma unroll
            for (int i = 0; i < NREAD; ++i) fr[i] = *reinterpret_cast<const bf16x8_t *>(((i & 1) ? c : b) + ((wave * 8 + i) * 1024 + lane * 16) % SLOT);
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i % NR], fr[(i + 1) % NR], acc[i & 3], 0, 0, 0);
        }
    }
    if (sink && acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[threadIdx.x] = acc[0][0];
}

template <int BM, int NW, int NSLOT, int NREAD, int NMFMA>
void run_half(const char *name, const uint16_t *A, const uint16_t *W, float *sink, int wg_per_cu_hint) {
    const int M = 3072, N = 8192, K = 1024;
    const int tiles = (M / BM) * (N / BM);
    const size_t lds = (size_t)NSLOT * BM * 128;
    auto k = stream_half_kernel<BM, NW, NSLOT, NREAD, NMFMA>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NW * 64), lds, 0, A, W, K, N / BM, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(NW * 64), lds, 0, A, W, K, N / BM, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, bytes = (double)tiles * (K / 64) * (2 * BM) * 128;
    printf("%-44s tiles %5d  lds %6zu B (%d wg/cu)  %7.1f us  %6.1f MB  %5.1f GB/s per CU  %5.2f TB/s\n", name, tiles, lds, wg_per_cu_hint, us,
           bytes / 1e6, bytes / us / 1e3 / 256, bytes / us / 1e6);
}

int main() {
    uint16_t *A, *W;
    float *sink;
    hipMalloc(&A, 3072 * 1024 * 2);
    hipMalloc(&W, 8192 * 1024 * 2);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(A, 0, 3072 * 1024 * 2);
    hipMemset(W, 0, 8192 * 1024 * 2);
    printf("-- staging only\n");
    run<128, 128, 8, 2, 0, 0>("128+128 rows, 8 waves, 2 stages", A, W, sink, 2);
    run<128, 128, 8, 3, 0, 0>("128+128 rows, 8 waves, 3 stages (1 wg/cu)", A, W, sink, 1);
    run<256, 128, 8, 3, 0, 0>("256+128 rows, 8 waves, 3 stages", A, W, sink, 1);
    run<256, 128, 16, 3, 0, 0>("256+128 rows, 16 waves, 3 stages", A, W, sink, 1);
    run<256, 256, 8, 2, 0, 0>("256+256 rows, 8 waves, 2 stages", A, W, sink, 1);
    run<256, 256, 16, 2, 0, 0>("256+256 rows, 16 waves, 2 stages", A, W, sink, 1);
    run<128, 64, 8, 3, 0, 0>("128+64 rows, 8 waves, 3 stages", A, W, sink, 2);
    run<128, 128, 8, 2, 0, 0, 1>("128+128 rows, 8 waves, 2 stages, m fastest", A, W, sink, 2);
    run<128, 128, 8, 2, 0, 0, 2>("128+128 rows, 8 waves, 2 stages, 8x8 blocks", A, W, sink, 2);
    run<128, 64, 8, 3, 0, 0, 1>("128+64 rows, 8 waves, 3 stages, m fastest", A, W, sink, 2);
    run<128, 64, 8, 3, 0, 0, 2>("128+64 rows, 8 waves, 3 stages, 8x8 blocks", A, W, sink, 2);
    run<256, 128, 8, 3, 0, 0, 1>("256+128 rows, 8 waves, 3 stages, m fastest", A, W, sink, 1);
    run<256, 128, 8, 3, 0, 0, 2, 4, 8>("256+128 rows, 8 waves, 3 stages, 4x8 blocks", A, W, sink, 1);
    run_half<128, 8, 4, 0, 0>("128+128 rows, 8 waves, 4 half-slots (m fastest from here)", A, W, sink, 2);
    run_half<128, 8, 5, 0, 0>("128+128 rows, 8 waves, 5 half-slots", A, W, sink, 2);
    run_half<128, 8, 8, 0, 0>("128+128 rows, 8 waves, 8 half-slots (1 wg/cu)", A, W, sink, 1);
    run_half<128, 16, 10, 0, 0>("128+128 rows, 16 waves, 10 half-slots (1 wg/cu)", A, W, sink, 1);
    printf("-- with the tile's LDS reads + MFMAs\n");
    run_half<128, 8, 4, 12, 16>("128+128 rows, 8 waves, 4 half-slots, 12r/16m", A, W, sink, 2);
    run_half<128, 8, 5, 12, 16>("128+128 rows, 8 waves, 5 half-slots, 12r/16m", A, W, sink, 2);
    run_half<128, 8, 8, 12, 16>("128+128 rows, 8 waves, 8 half-slots, 12r/16m", A, W, sink, 1);
    run_half<128, 16, 10, 6, 8>("128+128 rows, 16 waves, 10 half-slots, 6r/8m", A, W, sink, 1);
    run<128, 128, 8, 2, 12, 16>("128+128 rows, 8 waves, 2 stages, 12r/16m", A, W, sink, 2);
    run<128, 128, 8, 2, 12, 16, 1>("128+128 rows, 2 stages, 12r/16m, m fastest", A, W, sink, 2);
    run<128, 128, 8, 2, 12, 16, 2>("128+128 rows, 2 stages, 12r/16m, 8x8 blocks", A, W, sink, 2);
    run<256, 128, 8, 3, 16, 32, 2, 4, 8>("256+128 rows, 3 stages, 16r/32m, 4x8 blocks", A, W, sink, 1);
    run<256, 128, 8, 3, 16, 32>("256+128 rows, 8 waves, 3 stages, 16r/32m", A, W, sink, 1);
    run<256, 128, 16, 3, 12, 16>("256+128 rows, 16 waves, 3 stages, 12r/16m", A, W, sink, 1);
    run<256, 256, 8, 2, 16, 64>("256+256 rows, 8 waves, 2 stages, 16r/64m", A, W, sink, 1);
    run<256, 256, 16, 2, 16, 32>("256+256 rows, 16 waves, 2 stages, 16r/32m", A, W, sink, 1);
    return 0;
}
