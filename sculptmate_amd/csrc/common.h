// Shared helpers for the gfx950 kernels and their C-ABI launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "../../include/sculpt_hip.h"

namespace sculpt {

void set_error(const char *fmt, ...);

#define SC_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            sculpt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                              __FILE__, __LINE__);                                       \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

#define SC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            sculpt::set_error(__VA_ARGS__);   \
            return 2;                         \
        }                                     \
    } while (0)

#define SC_LAUNCH_CHECK() SC_HIP(hipGetLastError())

static inline hipStream_t as_stream(sculpt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// number of CUs of the current device (cached)
int num_cus();

// Non-default kernel forms, for the tests that pin each adopted form against another and for A/B timing: ONE environment variable
// per kernel family (SCULPT_GEMM_TILE, SCULPT_L3_TILE, SCULPT_ATTN_FORM, SCULPT_DENSITY_FORM, SCULPT_MC_FORM), a comma-separated
// list of tokens read per call.  form_has(var, "nopipe"): the token is there; form_int(var, "gm", -1): the value of "gm=4".
static inline bool form_token(const char *list, const char *token, const char **value) {
    if (!list) return false;
    const size_t n = strlen(token);
    for (const char *p = list; *p;) {
        const char *q = strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len >= n && strncmp(p, token, n) == 0 && (len == n || p[n] == '=')) {
            if (value) *value = len == n ? nullptr : p + n + 1;
            return true;
        }
        if (!q) break;
        p = q + 1;
    }
    return false;
}
static inline bool form_has(const char *var, const char *token) { return form_token(getenv(var), token, nullptr); }
static inline int form_int(const char *var, const char *key, int dflt) {
    const char *v = nullptr;
    return (form_token(getenv(var), key, &v) && v) ? atoi(v) : dflt;
}

// XCD-aware tile order (guide T1, bijective form): workgroups are handed to the 8 XCDs round-robin by linear id, and
// every XCD has a private L2; this returns a tile index such that the workgroups sharing an XCD (same id % 8) get a
// CONTIGUOUS chunk of the natural tile order, so tiles that share an operand panel hit the same L2.
__device__ __forceinline__ int xcd_tile(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round to nearest even, a NaN stays a NaN: v_cvt_pk_bf16_f32.  (Until round 6 this was the integer form u + 0x7fff + ((u >> 16) & 1)
// with a NaN test in front: ~8 instructions and, the test being a branch, a pair of exec-mask edits per value in an unrolled
// epilogue -- gemm256_kernel's 96 conversions per lane were 105 s_and_saveexec / 91 s_nop, tools/gemm_timeline.py.  Same bits for
// every number; a NaN keeps its sign and turns quiet either way.)
typedef float sc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sc_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const sc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sc_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

// In-kernel timeline stamps of -DSCULPT_EXPERIMENTS builds (tools/gemm_timeline.py): `g` is a kernel's argument block with a member
// `unsigned long long *stamps` (nullptr in every product launch); per workgroup 16 words -- s_memrealtime (100 MHz) at the stamped
// points 0..4, HW_ID, XCC_ID, and at [8 + k] s_memtime (shader clock) of the same points.
#ifdef SCULPT_EXPERIMENTS
#define GEMM_STAMP(g, k)                                                                                                        \
    do {                                                                                                                        \
        if ((g).stamps && threadIdx.x == 0) {                                                                                   \
            unsigned long long *sp_ = (g).stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;                          \
            sp_[k] = __builtin_amdgcn_s_memrealtime();                                                                          \
            sp_[8 + (k)] = __builtin_amdgcn_s_memtime();                                                                        \
        }                                                                                                                       \
    } while (0)
#define GEMM_STAMP_IDS(g)                                                                                                       \
    do {                                                                                                                        \
        if ((g).stamps && threadIdx.x == 0) {                                                                                   \
            unsigned long long *sp_ = (g).stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;                          \
            sp_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                                                 \
            sp_[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                                                \
        }                                                                                                                       \
    } while (0)
extern unsigned long long *g_gemm_stamps;   // gemm.hip; set by sculpt_experiment_gemm_stamps
#else
#define GEMM_STAMP(g, k) do { } while (0)
#define GEMM_STAMP_IDS(g) do { } while (0)
#endif

}  // namespace sculpt
