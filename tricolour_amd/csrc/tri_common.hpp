// tri_common.hpp / tricolour_amd.hip -- MI355X (gfx950 / CDNA4) SumThreshold RFI flagger.
//
// Hand-written HIP implementation of the hot path of ratt-ru/tricolour
// (reference tricolour/flagging.py:175-976, 1076-1196; packing.py:243-278,
// 369-415) behind the C ABI of include/tricolour_amd.h.  Not a translation:
// the reference is a serial per-baseline numba loop nest; here every step is a
// batched kernel over (window, line) with the window held in HBM in BOTH
// orientations -- "TF" (time rows, channel columns) and "FT" (channel rows,
// time columns) -- so that
//   * every sequential float64 recurrence of the reference (box-filter running
//     sums, SumThreshold prefix sums, NaN interpolation) runs one thread per
//     line with the line index on the coalesced axis ("column kernels"), in
//     exactly the reference's order of operations => bit-exact by
//     construction, and
//   * every exact median runs over lines that are contiguous in memory
//     ("row select": multi-pass radix select on the |x| bit patterns).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math: the
// results must follow IEEE evaluation order).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/tricolour_amd.h"

#define TRI_MAD_NORMAL 1.4826  // flagging.py:22

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                          \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess)                                                \
            return set_err(TRI_EHIP, "%s failed: %s (%s:%d)", #expr,          \
                           hipGetErrorString(e__), __FILE__, __LINE__);       \
    } while (0)

#define LAUNCHCHK()                                                           \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess)                                                \
            return set_err(TRI_EHIP, "kernel launch failed: %s (%s:%d)",      \
                           hipGetErrorString(e__), __FILE__, __LINE__);       \
    } while (0)

// ---------------------------------------------------------------------------
// Kernel log (measurement hook, off by default): every launch of this translation unit goes through the macro
// below, which counts launches per kernel symbol while the calling thread has the log switched on
// (tri_kernel_log()).  bench.py uses it to name the device kernels a roofline leg timed and to count the launches
// of each kernel family in one step -- the dominant kernel is derived from that, not hard-coded.
// ---------------------------------------------------------------------------
#include <unordered_map>
static thread_local bool g_klog_on = false;
static thread_local std::unordered_map<const void*, long long>* g_klog = nullptr;
static inline void tri_klog(const void* fn) {
    if (!g_klog_on) return;
    if (!g_klog) g_klog = new std::unordered_map<const void*, long long>();
    ++(*g_klog)[fn];
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                  \
    do {                                                                                                   \
        tri_klog(reinterpret_cast<const void*>(kernelName));                                               \
        (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);               \
    } while (0)

extern "C" const char* tri_last_error(void) { return g_err; }
extern "C" int tri_version(void) { return 100; }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// |complex64| with libm hypotf semantics (numba lowers abs(complex64) to
// hypotf, flagging.py:856): glibc evaluates (float)sqrt((double)x*x +
// (double)y*y) with the C99 infinity rule.  Products are exact in float64, one
// rounding in the sum, IEEE sqrt, one narrowing cast.
__device__ __forceinline__ float tri_hypotf(float re, float im) {
    if (isinf(re) || isinf(im)) return INFINITY;
    double s = (double)re * (double)re + (double)im * (double)im;
    return (float)sqrt(s);
}

template <int VD>
__device__ __forceinline__ float load_amp(const void* vis, size_t i) {
    if (VD == TRI_VIS_C64) {
        float2 z = reinterpret_cast<const float2*>(vis)[i];
        return tri_hypotf(z.x, z.y);
    } else if (VD == TRI_VIS_F64) {
        return (float)fabs(reinterpret_cast<const double*>(vis)[i]);   // (k_prepare accumulates the float64 value itself)
    } else {
        return fabsf(reinterpret_cast<const float*>(vis)[i]);
    }
}

template <int VD>
__device__ __forceinline__ bool load_isnan(const void* vis, size_t i) {
    if (VD == TRI_VIS_C64) {
        float2 z = reinterpret_cast<const float2*>(vis)[i];
        return isnan(z.x) || isnan(z.y);
    } else if (VD == TRI_VIS_F64) {
        return isnan(reinterpret_cast<const double*>(vis)[i]);
    } else {
        return isnan(reinterpret_cast<const float*>(vis)[i]);
    }
}

// First statement of a rarely taken branch that must STAY a branch: the compiler otherwise if-converts short arms
// (it did so for the IEEE-division fallback of box_divide(): both divisions were computed for every output and
// selected afterwards, 22 extra vector instructions per sample).  A volatile asm cannot be executed speculatively.
#define TRI_KEEP_BRANCH() asm volatile("; cold path")
