// Shared helpers for libpdhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/pdhip.h"

namespace pdhip {

void set_error(const char* fmt, ...);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define PD_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            pdhip::set_error(__VA_ARGS__);            \
            return PDHIP_E_ARG;                       \
        }                                             \
    } while (0)

#define PD_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            pdhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PDHIP_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define PD_LAUNCH_CHECK()                                                                  \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) {                                                            \
            pdhip::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return PDHIP_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

// ---- lab builds.  Several PD_LAB_* switches compile deliberately WRONG results into a kernel (a skipped GroupNorm pass, a K loop
// without one operand stream, no barrier ...) to time what is left.  tools/lab_*.sh write such builds to csrc/build/lab_*.so, never to
// libpdhip.so, and pass -DPD_LAB_BUILD; should one of the switches reach a translation unit of the shipped library anyway, that unit
// registers itself here and pdhip_lab_build() (capi.hip) reports it -- tests/test_abi.py and every GPU test session refuse such a library.
#if defined(PD_LAB_BUILD) || defined(PD_LAB_SKIP_GN) || defined(PD_LAB_SK_NOA) || defined(PD_LAB_SK_NOB) || defined(PD_LAB_SK_NOCOMPUTE) || \
    defined(PD_LAB_SK_NOGLDS) || defined(PD_LAB_SK_NOBARRIER) || defined(PD_LAB_NOA) || defined(PD_LAB_NOB) || defined(PD_LAB_NOBARRIER) ||     \
    defined(PD_LAB_NODMA) || defined(PD_LAB_NOEPI) || defined(PD_LAB_NOLDS) || defined(PD_LAB_NOSTORE) || defined(PD_LAB_NOWAIT) ||             \
    defined(PD_LAB_AP_NOCOMPUTE) || defined(PD_LAB_AP_NOFETCH) || defined(PD_LAB_AP_NOSTORE) || defined(PD_LAB_AP_NOTRANS) ||                   \
    defined(PD_LAB_AP_LOOSEVM) || defined(PD_LAB_RASTER_LOADONLY) || defined(PD_LAB_RASTER_NORESOLVE) || defined(PD_LAB_RR_NOW) ||             \
    defined(PD_LAB_RR_NOA) || defined(PD_LAB_RR_NOMFMA) || defined(PD_LAB_HT_FILL) || defined(PD_LAB_NOPROWAIT) || defined(PD_LAB_HALO_NOSTORE)
#define PD_LAB_WRONG_RESULTS 1
#endif
extern int g_lab_units;                                    // translation units built with a wrong-result lab switch (capi.hip)
struct LabMark { LabMark() { ++g_lab_units; } };
#ifdef PD_LAB_WRONG_RESULTS
namespace { LabMark pd_lab_mark_; }
#endif

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- arithmetic contract: float32, one rounding per op, this order, no FMA.
// All geometry translation units are compiled with -ffp-contract=off.
struct Cam {
    float r[9], t[3], fx, fy, A, B;
};

__device__ __forceinline__ Cam load_cam(const float* p) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.r[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = p[9 + i];
    c.fx = p[12]; c.fy = p[13]; c.A = p[14]; c.B = p[15];
    return c;
}

__device__ __forceinline__ void cam_transform(const Cam& c, float x, float y, float z, float& xn, float& yn, float& zn) {
    float xc = ((c.r[0] * x + c.r[1] * y) + c.r[2] * z) + c.t[0];
    float yc = ((c.r[3] * x + c.r[4] * y) + c.r[5] * z) + c.t[1];
    float zc = ((c.r[6] * x + c.r[7] * y) + c.r[8] * z) + c.t[2];
    float w = -zc;
    xn = (c.fx * xc) / w;
    yn = (c.fy * yc) / w;
    zn = (c.A * zc + c.B) / w;
}

// order-preserving float <-> uint32 map (for atomic min/max on floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// long(clip(v, 0, hi)) with torch semantics for the values that occur (NaN -> treated as 0)
__device__ __forceinline__ int clip_to_int(float v, int hi) {
    float c = fminf(fmaxf(v, 0.0f), (float)hi);
    if (!(c == c)) c = 0.0f;
    return (int)c;
}

}  // namespace pdhip
