// tricolour_amd.hip -- MI355X (gfx950 / CDNA4) SumThreshold RFI flagger.
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
    } else {
        return fabsf(reinterpret_cast<const float*>(vis)[i]);
    }
}

template <int VD>
__device__ __forceinline__ bool load_isnan(const void* vis, size_t i) {
    if (VD == TRI_VIS_C64) {
        float2 z = reinterpret_cast<const float2*>(vis)[i];
        return isnan(z.x) || isnan(z.y);
    } else {
        return isnan(reinterpret_cast<const float*>(vis)[i]);
    }
}

// ---------------------------------------------------------------------------
// K1  _average_freq (flagging.py:819-875): |vis| -> f32, NaN -> flagged,
// flagged -> 0, channel averaging by `factor` (f32 accumulation in ascending
// channel order, f32 / count).  One thread per averaged sample.
// grid (ceil(T*Fa/256), W)
// ---------------------------------------------------------------------------
template <int VD>
__global__ void k_prepare(const void* __restrict__ vis, const uint8_t* __restrict__ iflags,
                          float* __restrict__ data, uint8_t* __restrict__ flags,
                          int T, int F, int Fa, int factor) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t NA = (size_t)T * Fa;
    if (idx >= NA) return;
    int t = (int)(idx / Fa), fo = (int)(idx % Fa);
    size_t w = blockIdx.y;
    size_t base = w * (size_t)T * F + (size_t)t * F;
    int f0 = fo * factor;
    int f1 = min(F, f0 + factor);
    float sum = 0.0f;
    int cnt = 0;
    for (int f = f0; f < f1; f++) {
        float a = load_amp<VD>(vis, base + f);
        if (!iflags[base + f] && !isnan(a)) { sum += a; cnt++; }
    }
    size_t o = w * NA + idx;
    if (cnt == 0) { data[o] = 0.0f; flags[o] = 1; }
    else { data[o] = sum / (float)cnt; flags[o] = 0; }
}

__global__ void k_abs_c64(const float2* __restrict__ z, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tri_hypotf(z[i].x, z[i].y);
}

// ---------------------------------------------------------------------------
// K2  batched tiled transpose  src[W][R][C] -> dst[W][C][R]  (64x64 LDS tile)
// grid (ceil(C/64), ceil(R/64), W), block (64,4)
// ---------------------------------------------------------------------------
// `denom` != 0 (float images only): the stored value is x / denom -- the
// final division of _box_gaussian_filter1d (flagging.py:419), deferred from
// the latency-bound sequential filter kernel to this bandwidth-bound copy.
template <typename T>
__global__ void k_transpose(const T* __restrict__ src, T* __restrict__ dst, int R, int C,
                            size_t src_ws, size_t dst_ws, float denom) {
    __shared__ T tile[64][65];
    const T* s = src + (size_t)blockIdx.z * src_ws;
    T* d = dst + (size_t)blockIdx.z * dst_ws;
    int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    int tx = threadIdx.x, ty = threadIdx.y;
    for (int j = ty; j < 64; j += 4) {
        int r = r0 + j, c = c0 + tx;
        if (r < R && c < C) tile[j][tx] = s[(size_t)r * C + c];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int c = c0 + j, r = r0 + tx;
        if (r < R && c < C) {
            T v = tile[tx][j];
            if (sizeof(T) == 4 && denom != 0.0f) v = (T)((float)v / denom);
            d[(size_t)c * R + r] = v;
        }
    }
}

// uint8 transpose with 4-byte accesses on both sides (R % 4 == 0, C % 4 == 0):
// 64x64 byte tile; thread (tx, ty) of a (16,16) block moves uchar4 groups.
__global__ void k_transpose_u8x4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int R,
                                 int C, size_t src_ws, size_t dst_ws) {
    __shared__ uint8_t tile[64][68];
    const uint8_t* s = src + (size_t)blockIdx.z * src_ws;
    uint8_t* d = dst + (size_t)blockIdx.z * dst_ws;
    int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    int tx = threadIdx.x, ty = threadIdx.y;   // 16 x 16
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int r = r0 + ty + 16 * j, c = c0 + 4 * tx;
        uchar4 v = make_uchar4(0, 0, 0, 0);
        if (r < R && c < C) v = *reinterpret_cast<const uchar4*>(s + (size_t)r * C + c);
        *reinterpret_cast<uchar4*>(&tile[ty + 16 * j][4 * tx]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int c = c0 + ty + 16 * j, r = r0 + 4 * tx;   // output row = source column
        if (c < C && r < R) {
            int cc = ty + 16 * j;
            uchar4 v = make_uchar4(tile[4 * tx][cc], tile[4 * tx + 1][cc], tile[4 * tx + 2][cc], tile[4 * tx + 3][cc]);
            *reinterpret_cast<uchar4*>(d + (size_t)c * R + r) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// K3  segmented exact median of |x| over unflagged samples ("row select").
// np.median under numba (np/arraymath.py:1365-1399): odd n -> middle element,
// even n -> f32(a + b) / 2 in float64.  NaN when nothing is unflagged
// (flagging.py:276-277, 300-301).  |x| of a float32 is a sign-bit clear, so
// ordering |x| = ordering the low 31 bits as unsigned integers: a 4-digit
// (7+8+8+8 bit) radix select is exact.
// One workgroup per segment.  Segment (win, row, g) covers elements
//   data[win*WSd + row*RS + (seg_start[g] + i)*ES],  i < seg_len[g]
// (flags likewise with window stride WSf).
// Output med[(win*R + row)*G + g] (float64).
// grid (R*G, W), block 256
// ---------------------------------------------------------------------------
#define SEL_CACHE 8
#define SEL_BINS 2048
// Three radix passes over the 31-bit key: digits of 11, 10 and 10 bits.  The
// even-count partner (rank n/2 - 1) needs no extra pass: it equals the median
// key when that key is duplicated below rank n/2, else the largest occupied
// bin below it in the last histogram, else the largest key with a smaller
// 21-bit prefix (tracked during the last pass).
// VEC: segments are contiguous, 16-byte aligned and a multiple of 4 long ->
// float4 / uchar4 loads.
template <bool VEC>
__global__ void __launch_bounds__(256)
k_median(const float* __restrict__ data, const uint8_t* __restrict__ flags,
         double* __restrict__ med, size_t WSd, size_t WSf, size_t RS, size_t ES,
         const int64_t* __restrict__ seg_start, const int64_t* __restrict__ seg_len,
         int R, int G) {
    __shared__ unsigned hist[SEL_BINS];
    __shared__ unsigned sh_wsum[4];
    __shared__ unsigned sh_prefix, sh_k, sh_sel, sh_maxbelow, sh_lobin1;
    const unsigned SENT = 0xFFFFFFFFu;
    int seg = blockIdx.x;
    int row = seg / G, g = seg % G;
    size_t win = blockIdx.y;
    int64_t len = seg_len[g];
    size_t rel = (size_t)row * RS + (size_t)seg_start[g] * ES;
    data += win * WSd + rel;
    flags += win * WSf + rel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool cached = !VEC && len <= (int64_t)SEL_CACHE * 256;
    unsigned keys[SEL_CACHE];
    if (cached) {
#pragma unroll
        for (int u = 0; u < SEL_CACHE; u++) {
            int64_t i = (int64_t)u * 256 + tid;
            unsigned k = SENT;
            if (i < len) {
                size_t a = (size_t)i * ES;
                if (!flags[a]) k = __float_as_uint(data[a]) & 0x7FFFFFFFu;
            }
            keys[u] = k;
        }
    }
    if (tid == 0) { sh_maxbelow = 0; sh_lobin1 = 0; }
    unsigned prefix = 0, pmask = 0, kk = 0, n = 0;
    for (int p = 0; p < 3; p++) {
        const int shift = p == 0 ? 20 : (p == 1 ? 10 : 0);
        const unsigned dm = p == 0 ? 0x7FFu : 0x3FFu;
#pragma unroll
        for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = 0;
        __syncthreads();
        unsigned mb = 0;
        auto visit = [&](unsigned k) {
            if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & dm], 1u);
            else if (p == 2 && k < prefix) mb = max(mb, k);
        };
        if (cached) {
#pragma unroll
            for (int u = 0; u < SEL_CACHE; u++)
                if (keys[u] != SENT) visit(keys[u]);
        } else if (VEC) {
            const float4* d4 = reinterpret_cast<const float4*>(data);
            const uchar4* f4 = reinterpret_cast<const uchar4*>(flags);
            for (int64_t i = tid; i < len / 4; i += 256) {
                float4 dv = d4[i];
                uchar4 fv = f4[i];
                if (!fv.x) visit(__float_as_uint(dv.x) & 0x7FFFFFFFu);
                if (!fv.y) visit(__float_as_uint(dv.y) & 0x7FFFFFFFu);
                if (!fv.z) visit(__float_as_uint(dv.z) & 0x7FFFFFFFu);
                if (!fv.w) visit(__float_as_uint(dv.w) & 0x7FFFFFFFu);
            }
        } else {
            for (int64_t i = tid; i < len; i += 256) {
                size_t a = (size_t)i * ES;
                if (!flags[a]) visit(__float_as_uint(data[a]) & 0x7FFFFFFFu);
            }
        }
        if (p == 2 && mb) atomicMax(&sh_maxbelow, mb);
        __syncthreads();
        // bucket search: thread t owns bins [8t, 8t+8)
        unsigned v[8];
        {
            uint4 q0 = reinterpret_cast<const uint4*>(hist)[2 * tid];
            uint4 q1 = reinterpret_cast<const uint4*>(hist)[2 * tid + 1];
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
            v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        }
        unsigned sacc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) sacc += v[j];
        unsigned inc = sacc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned t2 = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t2;
        }
        if (lane == 63) sh_wsum[wave] = inc;
        __syncthreads();
        unsigned woff = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++) {
            unsigned t2 = sh_wsum[w2];
            if (w2 < wave) woff += t2;
            total += t2;
        }
        if (p == 0) { n = total; kk = total >> 1; }
        unsigned exc = woff + inc - sacc;
        if (total > 0 && kk >= exc && kk < exc + sacc) {
            unsigned c = exc;
            int j = 0;
#pragma unroll
            for (int q = 0; q < 7; q++)
                if (j == q && kk >= c + v[q]) { c += v[q]; j = q + 1; }
            sh_sel = 8u * tid + j;
            sh_prefix = prefix | ((8u * tid + j) << shift);
            sh_k = kk - c;
        }
        __syncthreads();
        if (n == 0) break;
        if (p == 2) {
            unsigned sel = sh_sel, cand = 0;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (v[j] && 8u * tid + j < sel) cand = 8u * tid + j + 1;
            if (cand) atomicMax(&sh_lobin1, cand);
        }
        prefix = sh_prefix;
        kk = sh_k;
        pmask |= dm << shift;
        __syncthreads();
    }
    if (tid == 0) {
        size_t oidx = (win * (size_t)R + row) * G + g;
        double m;
        if (n == 0) m = __longlong_as_double(0x7FF8000000000000LL);
        else if (n & 1u) m = (double)__uint_as_float(prefix);
        else {
            unsigned hi = prefix, lo;
            if (kk > 0) lo = hi;
            else if (sh_lobin1) lo = (hi & ~0x3FFu) | (sh_lobin1 - 1);
            else lo = sh_maxbelow;
            float sm = __uint_as_float(lo) + __uint_as_float(hi);
            m = (double)sm / 2.0;
        }
        med[oidx] = m;
    }
}

// ---------------------------------------------------------------------------
// K3b  Wave-per-segment form of k_median for segments of at most 1024 samples
// (time lines of a window, per-chunk channel runs): the segment's keys stay in
// registers (16 per lane), each wave owns a 256-bin LDS histogram, and the
// bucket search is a wave scan -- no workgroup-wide work per segment.  Four
// segments per 256-thread workgroup; barriers are executed uniformly.
// grid (ceil(R*G/4), W), block 256
// ---------------------------------------------------------------------------
#define MW_K 16   // register slots per lane of the largest instantiation (segments <= 1024)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor(v, o, 64));
    return v;
}

template <int KS, bool VEC4>   // KS register slots per lane: segments of at most 64 * KS samples
__global__ void __launch_bounds__(256)
k_median_wave(const float* __restrict__ data, const uint8_t* __restrict__ flags,
              double* __restrict__ med, size_t WSd, size_t WSf, size_t RS, size_t ES,
              const int64_t* __restrict__ seg_start, const int64_t* __restrict__ seg_len,
              int R, int G) {
    __shared__ unsigned hist[4][256];
    const unsigned SENT = 0xFFFFFFFFu;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = blockIdx.x * 4 + wave;
    const bool live = seg < R * G;
    const int row = live ? seg / G : 0, g = live ? seg % G : 0;
    const size_t win = blockIdx.y;
    const int len = live ? (int)seg_len[g] : 0;
    const size_t rel = (size_t)row * RS + (size_t)seg_start[g] * ES;
    const float* d = data + win * WSd + rel;
    const uint8_t* f = flags + win * WSf + rel;
    unsigned keys[KS];
    unsigned nloc = 0;
    if (VEC4) {
        // rows are 16-byte aligned and a multiple of 4 long: float4 / uchar4 loads
        // of the aligned groups covering the segment, samples outside it masked
        // (a selection does not care which lane holds which sample)
        const int mis = live ? (int)(seg_start[g] & 3) : 0;
        const float* d4 = d - mis;
        const uint8_t* f4 = f - mis;
#pragma unroll
        for (int u4 = 0; u4 < KS / 4; u4++) {
            const int i = (u4 * 64 + lane) * 4;          // offset of the aligned group
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
            uchar4 fv = make_uchar4(1, 1, 1, 1);
            if (i < len + mis) {
                dv = *reinterpret_cast<const float4*>(d4 + i);
                fv = *reinterpret_cast<const uchar4*>(f4 + i);
            }
            const int j = i - mis;                         // logical index of the group's first sample
            const bool v0 = j >= 0 && j < len && !fv.x;
            const bool v1 = j + 1 >= 0 && j + 1 < len && !fv.y;
            const bool v2 = j + 2 >= 0 && j + 2 < len && !fv.z;
            const bool v3 = j + 3 >= 0 && j + 3 < len && !fv.w;
            keys[4 * u4 + 0] = v0 ? (__float_as_uint(dv.x) & 0x7FFFFFFFu) : SENT;
            keys[4 * u4 + 1] = v1 ? (__float_as_uint(dv.y) & 0x7FFFFFFFu) : SENT;
            keys[4 * u4 + 2] = v2 ? (__float_as_uint(dv.z) & 0x7FFFFFFFu) : SENT;
            keys[4 * u4 + 3] = v3 ? (__float_as_uint(dv.w) & 0x7FFFFFFFu) : SENT;
            nloc += (v0 ? 1 : 0) + (v1 ? 1 : 0) + (v2 ? 1 : 0) + (v3 ? 1 : 0);
        }
    } else {
#pragma unroll
        for (int u = 0; u < KS; u++) {
            int i = u * 64 + lane;
            unsigned k = SENT;
            if (i < len) {
                size_t a = (size_t)i * ES;
                if (!f[a]) { k = __float_as_uint(d[a]) & 0x7FFFFFFFu; nloc++; }
            }
            keys[u] = k;
        }
    }
    const unsigned n = wave_sum_u32(nloc);
    // Normalise the keys to their minimum and radix-select only the B
    // significant bits of the spread: the leading digit then follows the
    // sample distribution (a plain top byte of a float is its exponent, which
    // puts almost every sample of a line into one or two bins and serialises
    // the LDS atomics).
    unsigned kmin = SENT, kmax = 0;
#pragma unroll
    for (int u = 0; u < KS; u++)
        if (keys[u] != SENT) { kmin = min(kmin, keys[u]); kmax = max(kmax, keys[u]); }
    kmin = ~wave_max_u32(~kmin);
    kmax = wave_max_u32(kmax);
    const int B = (n == 0 || kmax == kmin) ? 0 : 32 - __clz((int)(kmax - kmin));
    const int P = (B + 7) >> 3;
#pragma unroll
    for (int u = 0; u < KS; u++)
        if (keys[u] != SENT) keys[u] -= kmin;
    unsigned prefix = 0, pmask = 0, kk = n >> 1;
    unsigned* h = hist[wave];
    for (int p = 0; p < 4; p++) {
        const int shift = max(B - 8 * (p + 1), 0);
        const bool act = p < P;
        reinterpret_cast<uint4*>(h)[lane] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if (act) {
#pragma unroll
            for (int u = 0; u < KS; u++) {
                unsigned k = keys[u];
                if (k != SENT && (k & pmask) == prefix) atomicAdd(&h[(k >> shift) & 0xFFu], 1u);
            }
        }
        __syncthreads();
        if (act) {
            uint4 hv = reinterpret_cast<uint4*>(h)[lane];
            unsigned sacc = hv.x + hv.y + hv.z + hv.w;
            unsigned inc = sacc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                unsigned v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            unsigned exc = inc - sacc;
            bool mine = n > 0 && kk >= exc && kk < inc;
            unsigned dsel = 0, cbase = exc;
            if (mine) {
                if (kk < exc + hv.x) dsel = 0;
                else if (kk < exc + hv.x + hv.y) { dsel = 1; cbase = exc + hv.x; }
                else if (kk < exc + hv.x + hv.y + hv.z) { dsel = 2; cbase = exc + hv.x + hv.y; }
                else { dsel = 3; cbase = exc + hv.x + hv.y + hv.z; }
            }
            unsigned long long bm = __ballot(mine);
            if (bm) {
                int src = __ffsll((long long)bm) - 1;
                unsigned digit = __shfl(4u * lane + dsel, src, 64);
                unsigned base = __shfl(cbase, src, 64);
                prefix |= digit << shift;   // overlapping bits of a short last digit are already equal
                kk -= base;
            }
            pmask |= 0xFFu << shift;
        }
        __syncthreads();
    }
    unsigned cnt = 0, mx = 0;
#pragma unroll
    for (int u = 0; u < KS; u++) {
        unsigned k = keys[u];
        if (k != SENT && k < prefix) { cnt++; mx = max(mx, k); }
    }
    cnt = wave_sum_u32(cnt);
    mx = wave_max_u32(mx) + kmin;
    const unsigned hi = prefix + kmin;
    if (live && lane == 0) {
        size_t oidx = (win * (size_t)R + row) * G + g;
        double m;
        if (n == 0) m = __longlong_as_double(0x7FF8000000000000LL);
        else if (n & 1u) m = (double)__uint_as_float(hi);
        else {
            unsigned lo = (cnt == (n >> 1)) ? mx : hi;
            float sm = __uint_as_float(lo) + __uint_as_float(hi);
            m = (double)sm / 2.0;
        }
        med[oidx] = m;
    }
}

// spec_data[f][w] / spec_flags from the per-channel time medians
// (flagging.py:258-263): none unflagged -> 0 and flagged.
// med layout [w][f]; outputs in spectrum layout [Fa][Wn].
__global__ void k_spec_from_med(const double* __restrict__ med, float* __restrict__ sdata,
                                uint8_t* __restrict__ sflags, int Fa, int Wn) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Fa * Wn) return;
    int f = (int)(idx / Wn), w = (int)(idx % Wn);
    double m = med[(size_t)w * Fa + f];
    bool none = isnan(m);
    sdata[idx] = none ? 0.0f : (float)m;
    sflags[idx] = none ? 1 : 0;
}

// ---------------------------------------------------------------------------
// K4  _box_gaussian_filter1d (flagging.py:362-419) along the line axis of a
// [n][C] array, one thread per (column, image): four running box sums of
// width 2r+1 over a left-zero-padded line, float64 accumulator, every pass
// stored as float32 -- in exactly the reference's order (add the leading
// sample, store, subtract the trailing sample).  The padded line lives in a
// global scratch buffer buf[P = n + 4r][C] and the passes run in place, as in
// the reference; the never-written zero padding is synthesised instead of
// stored (rows below lo_p read as 0).
//   SRCMODE 0: pass 1 builds weight = !flag / data = flag ? 0 : x on the fly
//              from (srcData, srcFlags) (masked_gaussian_filter,
//              flagging.py:500-503).
//   SRCMODE 1: the unfiltered images were already written into rows
//              [4r, 4r+n) of bufW / bufO (by the transposing copy).
// Pass 4 divides by float32(d)**4 (host-computed, square-and-multiply as
// numba does) and writes rows [0,n) of dstW / dstO.
// grid (ceil(C/BLK), W, 2 images), block BLK
// ---------------------------------------------------------------------------
#define CF_U 8
template <int SRCMODE>
__global__ void __launch_bounds__(256)
k_colfilter(float* __restrict__ bufW, float* __restrict__ bufO,
            const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
            float* __restrict__ dstW, float* __restrict__ dstO,
            int n, int C, int r, float denom, size_t bws, size_t sws, size_t dws) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    size_t win = blockIdx.y;
    const int img = blockIdx.z;  // 0 = weight image, 1 = data image
    float* buf = (img == 0 ? bufW : bufO) + win * bws + c;
    float* dst = (img == 0 ? dstW : dstO) + win * dws + c;
    const float* sd = SRCMODE == 0 ? srcData + win * sws + c : nullptr;
    const uint8_t* sf = SRCMODE == 0 ? srcFlags + win * sws + c : nullptr;
    const int R2 = 2 * r, R4 = 4 * r;
    const int P = n + R4;
    const size_t Cs = (size_t)C;

    auto rd = [&](int j, int p) -> float {
        // value of padded[j] as seen by pass p (j in [0, P))
        if (p == 1) {
            int jj = j - R4;
            if (jj < 0) return 0.0f;
            if (SRCMODE == 0) {
                bool fl = sf[(size_t)jj * Cs] != 0;
                if (img == 0) return fl ? 0.0f : 1.0f;
                return fl ? 0.0f : sd[(size_t)jj * Cs];
            }
            return buf[(size_t)j * Cs];
        }
        if (p == 2 && j < R2) return 0.0f;
        return buf[(size_t)j * Cs];
    };

    for (int p = 1; p <= 4; p++) {
        double s = 0.0;
        if (p >= 3) {
            // flagging.py:404-405: pre-add padded[prev_start .. start + 2r)
            int i = 0;
            for (; i + CF_U <= R2; i += CF_U) {
                float v[CF_U];
#pragma unroll
                for (int u = 0; u < CF_U; u++) v[u] = rd(i + u, p);
#pragma unroll
                for (int u = 0; u < CF_U; u++) s += (double)v[u];
            }
            for (; i < R2; i++) s += (double)rd(i, p);
        }
        const int start = (p == 1) ? R2 : 0;
        const int stop = (p == 4) ? n : (p == 3 ? n + R2 : P);
        const int tail = n + R2;
        const int mainEnd = min(tail, stop);
        int i = start;
        for (; i + CF_U <= mainEnd; i += CF_U) {
            float lead[CF_U], prev[CF_U], o[CF_U];
#pragma unroll
            for (int u = 0; u < CF_U; u++) lead[u] = rd(i + u + R2, p);
#pragma unroll
            for (int u = 0; u < CF_U; u++) prev[u] = rd(i + u, p);
#pragma unroll
            for (int u = 0; u < CF_U; u++) {
                s += (double)lead[u];
                o[u] = (float)s;
                s -= (double)prev[u];
            }
            if (p < 4) {
#pragma unroll
                for (int u = 0; u < CF_U; u++) buf[(size_t)(i + u) * Cs] = o[u];
            } else {
#pragma unroll
                for (int u = 0; u < CF_U; u++) dst[(size_t)(i + u) * Cs] = o[u] / denom;
            }
        }
        for (; i < mainEnd; i++) {
            float lead = rd(i + R2, p);
            float prev = rd(i, p);
            s += (double)lead;
            float o = (float)s;
            s -= (double)prev;
            if (p < 4) buf[(size_t)i * Cs] = o;
            else dst[(size_t)i * Cs] = o / denom;
        }
        // flagging.py:412-416 (no leading sample left)
        for (i = mainEnd; i < stop; i++) {
            float prev = rd(i, p);
            float o = (float)s;
            s -= (double)prev;
            if (p < 4) buf[(size_t)i * Cs] = o;
            else dst[(size_t)i * Cs] = o / denom;
        }
    }
}

// ---------------------------------------------------------------------------
// K4b  Single-sweep variant of the box filter for moderate radii: the four
// passes run as a cascade of causal running sums in ONE pass over the line
// (HBM traffic: one read and one write per image sample), with each stage's
// 2r-deep delay line in LDS ([stage][slot][thread], conflict-free).
//
// Equivalence with the in-place passes of flagging.py:394-417 (t = causal
// index, in_p = input stream of pass p, zero where the reference's padded
// array holds padding or was never written):
//     s_p += in_p[t];  out_p[t] = f32(s_p);  s_p -= in_p[t - 2r]
// with in_1 = data (t in [0,n)), in_2 = out_1 (t in [0,n+2r)), in_3 = out_2
// (t in [0,n+4r)), in_4 = out_3 restricted to t >= 2r (the reference never
// forms padded_3 below index 0), result y[i] = out_4[i + 4r] / f32(d)**4.
// Adding / subtracting the synthesised zeros is exact, so every float64
// value equals the reference's.  Stage p+1 runs one step behind stage p so
// the four float64 chains of a step are independent.
// grid (ceil(C/BT), W, 2 images), block BT, dynamic LDS 4 * 2r * BT floats
// ---------------------------------------------------------------------------
template <int SRCMODE, bool DIV, bool TOUT>
__global__ void __launch_bounds__(256)
k_colfilter_lds(const float* __restrict__ srcW, const float* __restrict__ srcO,
                const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
                float* __restrict__ dstW, float* __restrict__ dstO,
                int n, int C, int r, float denom, size_t sws_img, size_t sws, size_t dws) {
    extern __shared__ float cf_ring[];
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const size_t win = blockIdx.y;
    const int img = blockIdx.z;
    const int BT = blockDim.x;
    const int R2 = 2 * r;
    const size_t Cs = (size_t)C;
    const float* src = SRCMODE == 1 ? ((img == 0 ? srcW : srcO) + win * sws_img + c) : nullptr;
    const float* sd = SRCMODE != 1 ? srcData + win * sws + c : nullptr;
    const uint8_t* sf = SRCMODE == 0 ? srcFlags + win * sws + c : nullptr;
    // SRCMODE 2: flags packed four line positions per 32-bit word, [n/4][C]
    // words (byte k of word q = flag of position 4q + k): one coalesced dword
    // load per four steps instead of a byte load per step
    const unsigned* sf4 = SRCMODE == 2 ? reinterpret_cast<const unsigned*>(srcFlags) + win * (sws / 4) + c : nullptr;
    // TOUT: the output is written TRANSPOSED -- line c becomes row c of an
    // [C][n] image (n % 4 == 0) -- four consecutive outputs per 16-byte store,
    // so the frequency-axis stage can consume it without a transpose pass.
    float* dst = (img == 0 ? dstW : dstO) + win * dws + (TOUT ? (size_t)c * n : (size_t)c);
    float tacc0 = 0.0f, tacc1 = 0.0f, tacc2 = 0.0f;
    float* ring = cf_ring + threadIdx.x;            // element (p, slot) at ((p*R2)+slot)*BT
    for (int k = 0; k < 4 * R2; k++) ring[(size_t)k * BT] = 0.0f;

    auto load = [&](int t) -> float {
        if (SRCMODE == 0) {
            bool fl = sf[(size_t)t * Cs] != 0;
            if (img == 0) return fl ? 0.0f : 1.0f;
            return fl ? 0.0f : sd[(size_t)t * Cs];
        }
        if (SRCMODE == 2) {
            unsigned wq = sf4[(size_t)(t >> 2) * Cs];
            bool fl = ((wq >> (8 * (t & 3))) & 0xFFu) != 0;
            if (img == 0) return fl ? 0.0f : 1.0f;
            return fl ? 0.0f : sd[(size_t)t * Cs];
        }
        return src[(size_t)t * Cs];
    };

    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    float o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;       // stage outputs of the previous step
    int slot1 = 0, slot2 = 0, slot3 = 0, slot4 = 0;
    const int total = n + 4 * r + 3;
    // Deep prefetch: with the LDS rings capping occupancy at 2 waves / SIMD,
    // bytes in flight (Little's law against ~2 us of HBM latency) come from
    // per-thread loads, not from thread count.
    constexpr int PF = 32;
    float pre[PF], cur[PF];
    unsigned prew[PF / 4];   // SRCMODE 2: raw packed-flag words in flight with pre[]
    // issue the loads of samples [t0, t0 + PF); the values are only consumed one
    // block later, so the loads stay in flight across a block of arithmetic
    auto issue = [&](int t0) {
        if (SRCMODE == 2) {
#pragma unroll
            for (int q = 0; q < PF / 4; q++) {
                int t = t0 + 4 * q;
                prew[q] = (t < n) ? sf4[(size_t)(t >> 2) * Cs] : 0x01010101u;
            }
            if (img == 1) {
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    int t = t0 + u;
                    pre[u] = (t < n) ? sd[(size_t)t * Cs] : 0.0f;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                int t = t0 + u;
                pre[u] = (t < n) ? load(t) : 0.0f;
            }
        }
    };
    issue(0);

    // One cascade step.  FAST = every stage is inside its steady range
    // (4r + 3 <= m, m < n): no bounds tests, so the four float64 chains of a
    // step are straight-line code the scheduler can interleave.
    // Ring reads are issued one step ahead (R2 >= 2, so the slot read for step
    // m + 1 differs from the slot written at step m): their LDS latency hides
    // behind the arithmetic of the current step instead of stalling each stage.
    float* rp1 = ring;
    float* rp2 = ring + (size_t)(1 * R2) * BT;
    float* rp3 = ring + (size_t)(2 * R2) * BT;
    float* rp4 = ring + (size_t)(3 * R2) * BT;
    float old1 = 0.0f, old2 = 0.0f, old3 = 0.0f, old4 = 0.0f;   // rings start zeroed
    auto nextslot = [&](int sl) { return (sl + 1 == R2) ? 0 : sl + 1; };

    auto step = [&](auto fastc, const int m, const float xin) {
        constexpr bool FAST = decltype(fastc)::value;
        const bool a4 = FAST || (m - 3 >= 0 && m - 3 < n + 4 * r);
        const bool a3 = FAST || (m - 2 >= 0 && m - 2 < n + 4 * r);
        const bool a2 = FAST || (m - 1 >= 0 && m - 1 < n + 4 * r);
        const bool a1 = FAST || (m < n + R2);
        const int ns1 = a1 ? nextslot(slot1) : slot1, ns2 = a2 ? nextslot(slot2) : slot2;
        const int ns3 = a3 ? nextslot(slot3) : slot3, ns4 = a4 ? nextslot(slot4) : slot4;
        // prefetch next step's trailing samples
        const float nold1 = rp1[(size_t)ns1 * BT], nold2 = rp2[(size_t)ns2 * BT];
        const float nold3 = rp3[(size_t)ns3 * BT], nold4 = rp4[(size_t)ns4 * BT];
        // stage 4 (time t4 = m - 3): input out_3[t4], present for 2r <= t4 < n + 4r
        if (a4) {
            const int t = m - 3;
            float in = (FAST || t >= R2) ? o3 : 0.0f;
            rp4[(size_t)slot4 * BT] = in;
            s4 += (double)in;
            float out = (float)s4;
            s4 -= (double)old4;
            int i = t - 4 * r;
            if (FAST || i >= 0) {
                float y = DIV ? out / denom : out;
                if (TOUT) {
                    const int ph = i & 3;
                    if (ph == 0) tacc0 = y;
                    else if (ph == 1) tacc1 = y;
                    else if (ph == 2) tacc2 = y;
                    else *reinterpret_cast<float4*>(dst + (i - 3)) = make_float4(tacc0, tacc1, tacc2, y);
                } else {
                    dst[(size_t)i * Cs] = y;
                }
            }
        }
        // stage 3 (t3 = m - 2): input out_2[t3], t3 in [0, n + 4r)
        if (a3) {
            float in = o2;
            rp3[(size_t)slot3 * BT] = in;
            s3 += (double)in;
            o3 = (float)s3;
            s3 -= (double)old3;
        }
        // stage 2 (t2 = m - 1): input out_1[t2] for t2 < n + 2r, then drains to n + 4r
        if (a2) {
            float in = (FAST || m - 1 < n + R2) ? o1 : 0.0f;
            rp2[(size_t)slot2 * BT] = in;
            s2 += (double)in;
            o2 = (float)s2;
            s2 -= (double)old2;
        }
        // stage 1 (t1 = m): input data[t1] for t1 < n, drains to n + 2r
        if (a1) {
            float in = (FAST || m < n) ? xin : 0.0f;
            rp1[(size_t)slot1 * BT] = in;
            s1 += (double)in;
            o1 = (float)s1;
            s1 -= (double)old1;
        }
        // a stage that did not run keeps its pending trailing sample
        if (a1) { old1 = nold1; slot1 = ns1; }
        if (a2) { old2 = nold2; slot2 = ns2; }
        if (a3) { old3 = nold3; slot3 = ns3; }
        if (a4) { old4 = nold4; slot4 = ns4; }
    };

    for (int m0 = 0; m0 < total; m0 += PF) {
        if (SRCMODE == 2) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                bool fl = ((prew[u >> 2] >> (8 * (u & 3))) & 0xFFu) != 0;
                cur[u] = (img == 0) ? (fl ? 0.0f : 1.0f) : (fl ? 0.0f : pre[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < PF; u++) cur[u] = pre[u];
        }
        issue(m0 + PF);
        if (m0 >= 4 * r + 3 && m0 + PF <= n) {
#pragma unroll
            for (int u = 0; u < PF; u++) step(std::true_type{}, m0 + u, cur[u]);
        } else {
#pragma unroll
            for (int u = 0; u < PF; u++) step(std::false_type{}, m0 + u, cur[u]);
        }
    }
}

// ---------------------------------------------------------------------------
// K4b'  Single-sweep box filter (K4b) whose INPUT images are stored
// transposed: line c is row c of an [C][ld] array (so the time-axis stage's TF
// output feeds the frequency-axis stage without a transpose pass).  Each
// workgroup (128 lines) stages 32 line positions at a time through an LDS
// tile: global loads are 128-byte row segments (coalesced along the line),
// the tile is read back column-wise, one value per thread and step.  The
// arithmetic is that of K4b.  grid (ceil(C/128), W, 2 images), block 128,
// dynamic LDS: 4 * 2r * 128 floats (rings) + 32 * 129 floats (tile)
// ---------------------------------------------------------------------------
#define CFT_BT 128
#define CFT_PF 32
template <bool DIV>
__global__ void __launch_bounds__(CFT_BT)
k_colfilter_lds_t(const float* __restrict__ srcW, const float* __restrict__ srcO,
                  float* __restrict__ dstW, float* __restrict__ dstO,
                  int n, int C, int ld, int r, float denom, size_t sws_img, size_t dws) {
    extern __shared__ float cf_ring[];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * CFT_BT;
    const int c = c0 + tid;
    const bool colok = c < C;
    const size_t win = blockIdx.y;
    const int img = blockIdx.z;
    const int R2 = 2 * r;
    const size_t Cs = (size_t)C;
    const float* src = (img == 0 ? srcW : srcO) + win * sws_img;
    float* dst = (img == 0 ? dstW : dstO) + win * dws + (colok ? c : 0);
    float* ring = cf_ring + tid;                                 // element (p, slot) at ((p*R2)+slot)*BT
    float* tile = cf_ring + (size_t)4 * R2 * CFT_BT;             // [CFT_PF][CFT_BT + 1]
    for (int k = 0; k < 4 * R2; k++) ring[(size_t)k * CFT_BT] = 0.0f;

    // staging: element e = j * 128 + tid of a [128 lines][32 positions] patch:
    // line = e / 32, position = e % 32  ->  lanes 0..31 read 128 contiguous bytes
    const int s_pos = tid & 31;
    const int s_line0 = tid >> 5;                                // + 4 j
    float pre[CFT_PF], cur[CFT_PF];
    auto issue = [&](int t0) {
#pragma unroll
        for (int j = 0; j < CFT_PF; j++) {
            int line = c0 + 4 * j + s_line0;
            int t = t0 + s_pos;
            pre[j] = (line < C && t < n) ? src[(size_t)line * ld + t] : 0.0f;
        }
    };
    // registers -> LDS tile (transposed) -> this thread's 32 samples
    auto exchange = [&]() {
        __syncthreads();                                         // previous tile fully consumed
#pragma unroll
        for (int j = 0; j < CFT_PF; j++) tile[s_pos * (CFT_BT + 1) + 4 * j + s_line0] = pre[j];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CFT_PF; u++) cur[u] = tile[u * (CFT_BT + 1) + tid];
    };

    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    float o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
    int slot1 = 0, slot2 = 0, slot3 = 0, slot4 = 0;
    float* rp1 = ring;
    float* rp2 = ring + (size_t)(1 * R2) * CFT_BT;
    float* rp3 = ring + (size_t)(2 * R2) * CFT_BT;
    float* rp4 = ring + (size_t)(3 * R2) * CFT_BT;
    float old1 = 0.0f, old2 = 0.0f, old3 = 0.0f, old4 = 0.0f;
    auto nextslot = [&](int sl) { return (sl + 1 == R2) ? 0 : sl + 1; };
    const int total = n + 4 * r + 3;

    auto step = [&](auto fastc, const int m, const float xin) {
        constexpr bool FAST = decltype(fastc)::value;
        const bool a4 = FAST || (m - 3 >= 0 && m - 3 < n + 4 * r);
        const bool a3 = FAST || (m - 2 >= 0 && m - 2 < n + 4 * r);
        const bool a2 = FAST || (m - 1 >= 0 && m - 1 < n + 4 * r);
        const bool a1 = FAST || (m < n + R2);
        const int ns1 = a1 ? nextslot(slot1) : slot1, ns2 = a2 ? nextslot(slot2) : slot2;
        const int ns3 = a3 ? nextslot(slot3) : slot3, ns4 = a4 ? nextslot(slot4) : slot4;
        const float nold1 = rp1[(size_t)ns1 * CFT_BT], nold2 = rp2[(size_t)ns2 * CFT_BT];
        const float nold3 = rp3[(size_t)ns3 * CFT_BT], nold4 = rp4[(size_t)ns4 * CFT_BT];
        if (a4) {
            const int t = m - 3;
            float in = (FAST || t >= R2) ? o3 : 0.0f;
            rp4[(size_t)slot4 * CFT_BT] = in;
            s4 += (double)in;
            float out = (float)s4;
            s4 -= (double)old4;
            int i = t - 4 * r;
            if ((FAST || i >= 0) && colok) dst[(size_t)i * Cs] = DIV ? out / denom : out;
        }
        if (a3) {
            float in = o2;
            rp3[(size_t)slot3 * CFT_BT] = in;
            s3 += (double)in;
            o3 = (float)s3;
            s3 -= (double)old3;
        }
        if (a2) {
            float in = (FAST || m - 1 < n + R2) ? o1 : 0.0f;
            rp2[(size_t)slot2 * CFT_BT] = in;
            s2 += (double)in;
            o2 = (float)s2;
            s2 -= (double)old2;
        }
        if (a1) {
            float in = (FAST || m < n) ? xin : 0.0f;
            rp1[(size_t)slot1 * CFT_BT] = in;
            s1 += (double)in;
            o1 = (float)s1;
            s1 -= (double)old1;
        }
        if (a1) { old1 = nold1; slot1 = ns1; }
        if (a2) { old2 = nold2; slot2 = ns2; }
        if (a3) { old3 = nold3; slot3 = ns3; }
        if (a4) { old4 = nold4; slot4 = ns4; }
    };

    issue(0);
    for (int m0 = 0; m0 < total; m0 += CFT_PF) {
        exchange();                 // tile of positions [m0, m0 + 32) -> cur[]
        issue(m0 + CFT_PF);         // next tile's loads stay in flight during the arithmetic
        if (m0 >= 4 * r + 3 && m0 + CFT_PF <= n) {
#pragma unroll
            for (int u = 0; u < CFT_PF; u++) step(std::true_type{}, m0 + u, cur[u]);
        } else {
#pragma unroll
            for (int u = 0; u < CFT_PF; u++) step(std::false_type{}, m0 + u, cur[u]);
        }
    }
}

// ---------------------------------------------------------------------------
// K4c  "Lane-per-stage" form of the single-sweep box filter for medium radii
// (the four 2r-deep delay lines of K4b no longer fit LDS at useful occupancy):
// the four cascade stages of one line run in the four lanes of a quad, every
// thread owning ONE stage and ONE LDS ring (2r floats), so the same LDS holds
// four times the threads.  Stage p takes its input from lane p-1's output of
// the previous step (DPP quad shuffle); the quad's four lanes prefetch four
// consecutive line positions per load instruction and the stage-0 lane picks
// them up by DPP broadcast.  Arithmetic per stage is identical to K4b (same
// causal running sums, same order) -- only the thread that executes a stage
// differs.  One wave (16 lines) per workgroup; no barriers.
// grid (ceil(C/16), W, 2 images), block 64, dynamic LDS 2r * 64 floats
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// quad_perm(a,b,c,d): lane i of each quad reads lane {a,b,c,d}[i]
#define QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int SRCMODE, bool DIV>
__global__ void __launch_bounds__(64)
k_colfilter_lane4(const float* __restrict__ srcW, const float* __restrict__ srcO,
                  const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
                  float* __restrict__ dstW, float* __restrict__ dstO,
                  int n, int C, int r, float denom, size_t sws_img, size_t sws, size_t dws) {
    extern __shared__ float cf_ring[];
    const int lane = threadIdx.x;
    const int p = lane & 3;                         // cascade stage of this lane
    const int c = blockIdx.x * 16 + (lane >> 2);
    const bool colok = c < C;
    const int cc = colok ? c : C - 1;               // out-of-range quads compute on a valid column, store nothing
    const size_t win = blockIdx.y;
    const int img = blockIdx.z;
    const int R2 = 2 * r;
    const size_t Cs = (size_t)C;
    const float* src = SRCMODE == 1 ? ((img == 0 ? srcW : srcO) + win * sws_img + cc) : nullptr;
    const float* sd = SRCMODE != 1 ? srcData + win * sws + cc : nullptr;
    const unsigned* sf4 = SRCMODE == 2 ? reinterpret_cast<const unsigned*>(srcFlags) + win * (sws / 4) + cc : nullptr;
    float* dst = (img == 0 ? dstW : dstO) + win * dws + cc;
    float* ring = cf_ring + lane;                   // slot k at ring[k * 64]
    for (int k = 0; k < R2; k++) ring[k * 64] = 0.0f;

    // per-stage ranges (see K4b): stage p runs for t in [0, tend); its input is
    // the upstream value for t in [ilo, ihi), zero otherwise
    const int tend = (p == 0) ? n + R2 : n + 4 * r;
    const int ilo = (p == 3) ? R2 : 0;
    const int ihi = (p == 0) ? n : ((p == 1) ? n + R2 : n + 4 * r);

    constexpr int PF = 32;                          // steps per block; each lane prefetches PF/4 positions
    float pre[PF / 4], cur[PF / 4];
    unsigned prew[PF / 4];
    // lane (line, p) loads positions t0 + 4 q + p
    auto issue = [&](int t0) {
#pragma unroll
        for (int q = 0; q < PF / 4; q++) {
            int t = t0 + 4 * q + p;
            if (SRCMODE == 2) {
                prew[q] = (t < n) ? sf4[(size_t)(t >> 2) * Cs] : 0x01010101u;
                pre[q] = (t < n && img == 1) ? sd[(size_t)t * Cs] : 0.0f;
            } else {
                pre[q] = (t < n) ? src[(size_t)t * Cs] : 0.0f;
            }
        }
    };
    issue(0);

    double s = 0.0;
    float o_last = 0.0f;                            // this stage's output of the previous step
    int slot = 0;
    float old = 0.0f;
    const int total = n + 4 * r + 3;
    for (int m0 = 0; m0 < total; m0 += PF) {
#pragma unroll
        for (int q = 0; q < PF / 4; q++) {
            if (SRCMODE == 2) {
                bool fl = ((prew[q] >> (8 * p)) & 0xFFu) != 0;   // byte p of the word = position 4q + p
                cur[q] = (img == 0) ? (fl ? 0.0f : 1.0f) : (fl ? 0.0f : pre[q]);
            } else {
                cur[q] = pre[q];
            }
        }
        issue(m0 + PF);
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int m = m0 + u;
            const int t = m - p;                    // this stage's time index
            // sample for stage 0: position m was loaded by lane (u & 3) of the quad
            float xs;
            if ((u & 3) == 0) xs = dpp_quad<QUAD_PERM(0, 0, 0, 0)>(cur[u >> 2]);
            else if ((u & 3) == 1) xs = dpp_quad<QUAD_PERM(1, 1, 1, 1)>(cur[u >> 2]);
            else if ((u & 3) == 2) xs = dpp_quad<QUAD_PERM(2, 2, 2, 2)>(cur[u >> 2]);
            else xs = dpp_quad<QUAD_PERM(3, 3, 3, 3)>(cur[u >> 2]);
            // upstream stage's previous output
            const float up = dpp_quad<QUAD_PERM(0, 0, 1, 2)>(o_last);
            const bool act = t >= 0 && t < tend;
            float in = (p == 0) ? xs : up;
            in = (t >= ilo && t < ihi) ? in : 0.0f;
            const int ns = (slot + 1 == R2) ? 0 : slot + 1;
            const float nold = ring[ns * 64];       // next step's trailing sample (R2 >= 2)
            if (act) {
                ring[slot * 64] = in;
                s += (double)in;
                o_last = (float)s;
                s -= (double)old;
                old = nold;
                slot = ns;
                if (p == 3) {
                    const int i = t - 4 * r;
                    if (i >= 0 && colok) dst[(size_t)i * Cs] = DIV ? o_last / denom : o_last;
                }
            }
        }
    }
}

// r == 0 on both axes: weight = !flag, data = flag ? 0 : x (flagging.py:500-503
// followed by the plain copy of flagging.py:465-466).
__global__ void k_build_wo(const float* __restrict__ data, const uint8_t* __restrict__ flags,
                           float* __restrict__ w, float* __restrict__ o, size_t nper,
                           size_t sws, size_t dws) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nper) return;
    size_t win = blockIdx.y;
    bool fl = flags[win * sws + i] != 0;
    w[win * dws + i] = fl ? 0.0f : 1.0f;
    o[win * dws + i] = fl ? 0.0f : data[win * sws + i];
}

__global__ void k_build_wo4(const float* __restrict__ data, const uint8_t* __restrict__ flags,
                            float* __restrict__ w, float* __restrict__ o, size_t n4per,
                            size_t sws, size_t dws) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4per) return;
    size_t win = blockIdx.y;
    uchar4 f = reinterpret_cast<const uchar4*>(flags + win * sws)[i];
    float4 d = reinterpret_cast<const float4*>(data + win * sws)[i];
    reinterpret_cast<float4*>(w + win * dws)[i] =
        make_float4(f.x ? 0.0f : 1.0f, f.y ? 0.0f : 1.0f, f.z ? 0.0f : 1.0f, f.w ? 0.0f : 1.0f);
    reinterpret_cast<float4*>(o + win * dws)[i] =
        make_float4(f.x ? 0.0f : d.x, f.y ? 0.0f : d.y, f.z ? 0.0f : d.z, f.w ? 0.0f : d.w);
}

// ---------------------------------------------------------------------------
// K5  masked_gaussian_filter tail (flagging.py:506-513) and the background
// residual (flagging.py:563-566):  bg = w == 0 ? NaN : o / w;
//   MODE 0: o <- bg          MODE 1: o <- |data - bg|
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void k_masked_div(const float* __restrict__ w, float* __restrict__ o,
                             const float* __restrict__ data, size_t nper, size_t ws_wo,
                             size_t ws_data, float denom, uint8_t* __restrict__ nanflag, int C) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nper) return;
    size_t win = blockIdx.y;
    float wv = w[win * ws_wo + i];
    float ov = o[win * ws_wo + i];
    if (denom != 0.0f) { wv = wv / denom; ov = ov / denom; }   // deferred flagging.py:419
    float bg = (wv == 0.0f) ? NAN : ov / wv;
    // remember which lines (columns) hold a NaN: only those need the
    // sequential interpolation pass
    if (MODE == 0 && nanflag && isnan(bg)) nanflag[win * (size_t)C + (i % C)] = 1;
    if (MODE == 1) bg = fabsf(data[win * ws_data + i] - bg);
    o[win * ws_wo + i] = bg;
}

// flags |= resid > median * (MAD_NORMAL * reject)   (flagging.py:567-574);
// float32 residual compared in float64; NaN compares false.
// Array layout [L][C] per window; chunk_of[l] gives the chunk of line index l.
// thr index: TWOD ? (win*G + g) : (c*G + g)   [spectrum layout: column = window]
template <bool TWOD>
__global__ void k_reject(const float* __restrict__ resid, uint8_t* __restrict__ flags,
                         const double* __restrict__ med, const int* __restrict__ chunk_of,
                         double scale, int L, int C, int G, size_t ws_resid, size_t ws_flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * C) return;
    size_t win = blockIdx.y;
    int l = (int)(i / C), c = (int)(i % C);
    int g = chunk_of[l];
    double m = TWOD ? med[win * G + g] : med[(size_t)c * G + g];
    double thr = m * scale;
    if ((double)resid[win * ws_resid + i] > thr) flags[win * ws_flags + i] = 1;
}

// ---------------------------------------------------------------------------
// K6  _linearly_interpolate_nans1d (flagging.py:307-344) along the line axis
// of [L][C], one thread per column.  numba typing: grad = (f32 - f32) / int64
// -> float64; value = f32(f32 + int64 * f64) evaluated in float64.
// grid (ceil(C/256), W)
// ---------------------------------------------------------------------------
__global__ void k_colinterp(float* __restrict__ a, int L, int C, size_t ws,
                            const uint8_t* __restrict__ nanflag) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (nanflag && !nanflag[(size_t)blockIdx.y * C + c]) return;   // no NaN in this line
    float* x = a + (size_t)blockIdx.y * ws + c;
    const size_t Cs = (size_t)C;
    int last = -1;       // index of the last valid sample
    float lastv = 0.0f;
    int run = 0;         // start of the current NaN run
    for (int i = 0; i < L; i++) {
        float v = x[(size_t)i * Cs];
        if (isnan(v)) continue;
        if (run < i) {
            if (last < 0) {
                for (int j = run; j < i; j++) x[(size_t)j * Cs] = v;  // extrapolate backwards
            } else {
                float diff = v - lastv;
                double grad = (double)diff / (double)(i - last);
                for (int j = run; j < i; j++)
                    x[(size_t)j * Cs] = (float)((double)lastv + (double)(j - last) * grad);
            }
        }
        last = i;
        lastv = v;
        run = i + 1;
    }
    if (run < L) {
        float fill = last < 0 ? 0.0f : lastv;  // all NaN -> zeros; else extrapolate forwards
        for (int j = run; j < L; j++) x[(size_t)j * Cs] = fill;
    }
}

// out = a - b  (flagging.py:950, 962)
__global__ void k_sub(const float* __restrict__ a, const float* __restrict__ b,
                      float* __restrict__ out, size_t nper, size_t ws_a, size_t ws_b,
                      size_t ws_o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nper) return;
    size_t win = blockIdx.y;
    out[win * ws_o + i] = a[win * ws_a + i] - b[win * ws_b + i];
}

__global__ void k_or(uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t nper,
                     size_t ws_a, size_t ws_b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nper) return;
    size_t win = blockIdx.y;
    if (b[win * ws_b + i]) a[win * ws_a + i] = 1;
}

__global__ void k_copy_u8(const uint8_t* __restrict__ a, uint8_t* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

// flags[w][t][f] |= spec[f][w]   (flagging.py:954); spec in spectrum layout
__global__ void k_or_spec(uint8_t* __restrict__ flags, const uint8_t* __restrict__ spec, int T,
                          int Fa, int Wn) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Fa) return;
    size_t win = blockIdx.y;
    int f = (int)(i % Fa);
    if (spec[(size_t)f * Wn + win]) flags[win * (size_t)T * Fa + i] = 1;
}

// ---------------------------------------------------------------------------
// K7  _sum_threshold1d + _convolve_flags (flagging.py:582-681) along the line
// axis of [L][C]: one thread per (column, chunk) streams down its padded line
// ONCE, running all windows as a cascade: stage j ingests position i (clamp
// with the flags of stages < j, float64 sequential prefix sum), forms the
// rolling sum S_k = cum[k+w] - cum[k] for k = i + 1 - w from a ring of the
// last w prefix values, thresholds +S and -S against thr0 / rho^log2(w), and
// dilates hits over w samples; stage j+1 runs w_j - 1 positions behind so
// that its clamp sees exactly the flags the reference's window loop would
// (flagging.py:638-674, window order as given).  Every float64 value is
// produced by the same operations in the same order as the reference.
// Input flags are used only in the MAD (flagging.py:622), never OR-ed in.
//
// Dynamic variant: arbitrary windows; prefix rings and the position ring of
// accumulated (pos,neg) bits live in a global scratch laid out
// [slot][thread] (coalesced).
// grid (ceil(C/BLK), G, W), block BLK
// ---------------------------------------------------------------------------
struct StWin {
    int nw;
    int w[TRI_MAX_WINDOWS];
    double tf[TRI_MAX_WINDOWS];     // rho ** log2(w)           (host libm, flagging.py:641)
    double scale[TRI_MAX_WINDOWS];  // (double)(float)(1.0 / w) (flagging.py:664)
    int ringoff[TRI_MAX_WINDOWS];   // slot offset of stage j's prefix ring
    int ringtot;                    // sum of w
    int delay[TRI_MAX_WINDOWS + 1]; // D_j = sum_{j'<j} (w_j' - 1)
    int acccap;                     // >= D_nw + 1
    int maxw;
};

__global__ void __launch_bounds__(256)
k_colst_dyn(const float* __restrict__ data, const double* __restrict__ med,
            uint8_t* __restrict__ out, double* __restrict__ ringbuf,
            uint8_t* __restrict__ accbuf, const int64_t* __restrict__ chunk_ends, StWin sw,
            double thr_scale, int L, int C, int G, size_t ws_data, size_t ws_out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int g = blockIdx.y;
    size_t win = blockIdx.z;
    int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    if (c1 <= c0) return;
    // thread-private scratch, [slot][thread]
    size_t nthreads = (size_t)gridDim.z * G * C;
    size_t tidg = (win * G + g) * (size_t)C + c;
    double* ring = ringbuf + tidg;
    uint8_t* acc = accbuf + tidg;
    const float* x = data + win * ws_data + c;
    uint8_t* o = out + win * ws_out + c;
    const size_t Cs = (size_t)C;

    // flagging.py:622-628
    float mad = (float)med[(win * (size_t)C + c) * G + g];
    float thr0 = isnan(mad) ? INFINITY : (float)((double)mad * thr_scale);
    // flagging.py:630-633 (slicing clamps to the axis length)
    int p0 = max(c0 - sw.maxw + 1, 0);
    int p1 = min(c1 + sw.maxw - 1, L);
    int Lp = p1 - p0;

    const int nw = sw.nw;
    double thr[TRI_MAX_WINDOWS], cumlast[TRI_MAX_WINDOWS];
    int sincep[TRI_MAX_WINDOWS], sincen[TRI_MAX_WINDOWS];
    for (int j = 0; j < nw; j++) {
        thr[j] = (double)thr0 / sw.tf[j];
        cumlast[j] = 0.0;
        sincep[j] = sincen[j] = 1 << 30;
        ring[(size_t)sw.ringoff[j] * nthreads] = 0.0;  // cum[0] = 0 in slot 0
    }
    for (int s = 0; s < sw.acccap; s++) acc[(size_t)s * nthreads] = 0;

    const int total = Lp + sw.delay[nw];
    for (int n = 0; n < total; n++) {
        for (int j = 0; j < nw; j++) {
            const int w = sw.w[j];
            int i = n - sw.delay[j];      // ingest position
            if (i < 0) continue;
            int e = i + 1 - w;            // emit position
            if (e >= Lp) continue;
            bool hp = false, hn = false;
            if (i < Lp) {
                uint8_t a = acc[(size_t)(i % sw.acccap) * nthreads];
                float xf = x[(size_t)(p0 + i) * Cs];
                double clamped = (double)xf;
                double limit = thr[j];
                if ((a & 1) && clamped > limit) clamped = limit;
                else if ((a & 2) && clamped < -limit) clamped = -limit;
                double cumnew = cumlast[j] + clamped;
                cumlast[j] = cumnew;
                size_t slot = (size_t)(sw.ringoff[j] + ((i + 1) % w)) * nthreads;
                if (e >= 0) {
                    double S = cumnew - ring[slot];
                    hp = S * sw.scale[j] > limit;
                    hn = S * (-sw.scale[j]) > limit;
                }
                ring[slot] = cumnew;
            }
            if (e >= 0) {
                sincep[j] = hp ? 0 : min(sincep[j] + 1, 1 << 30);
                sincen[j] = hn ? 0 : min(sincen[j] + 1, 1 << 30);
                uint8_t add = (sincep[j] < w ? 1 : 0) | (sincen[j] < w ? 2 : 0);
                if (add) acc[(size_t)(e % sw.acccap) * nthreads] |= add;
            }
        }
        int ef = n - sw.delay[nw];        // position final after the last stage
        if (ef >= 0 && ef < Lp) {
            size_t aslot = (size_t)(ef % sw.acccap) * nthreads;
            uint8_t a = acc[aslot];
            acc[aslot] = 0;               // recycle the slot
            int pos = p0 + ef;
            if (pos >= c0 && pos < c1) o[(size_t)pos * Cs] = a ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------
// K7b  Register-resident SumThreshold cascade for power-of-two windows
// {W0,W1,W2,W3} with W3 <= 8 (the library default and every shipped strategy's
// time axis: 1,2,4,8).  Same arithmetic as k_colst_dyn, but
//   * the prefix rings (w doubles per stage), the 8-deep sample ring and the
//     hit / input-flag histories (bit shift registers) live in VGPRs: the tick
//     loop is unrolled by 8 so that every ring index is a compile-time
//     constant;
//   * stage j+1 runs w_j positions behind stage j (one more than necessary),
//     so the flag hand-off crosses a tick boundary and the four stages of one
//     tick are independent instruction streams;
//   * for a power-of-two window, S * f32(1/w) > thr  <=>  S > thr * w exactly
//     (both sides scale by 2^k; S is a multiple of 2^-203 or larger, far above
//     the underflow range), so the threshold tests need no multiply.
// One thread per (column, chunk); HBM traffic = 4 B in + 1 B out per sample.
// grid (ceil(C/BLK), G, W), block BLK
// ---------------------------------------------------------------------------
struct StFusedArgs {
    double tf[4];   // rho ** log2(w)
};

#ifndef ST_WAVES
#define ST_WAVES 2
#endif
template <int W0, int W1, int W2, int W3>
__global__ void __launch_bounds__(256, ST_WAVES)
k_colst_fused(const float* __restrict__ data, const double* __restrict__ med,
              uint8_t* __restrict__ out, const int64_t* __restrict__ chunk_ends,
              StFusedArgs fa, double thr_scale, int L, int C, int G, size_t ws_data,
              size_t ws_out) {
    constexpr int W[4] = {W0, W1, W2, W3};
    constexpr int D[4] = {0, W0, W0 + W1, W0 + W1 + W2};   // ingest delay of stage j
    constexpr int DOUT = W0 + W1 + W2 + W3 - 1;             // final flags lag the head by this
    constexpr int MAXW = W3;
    constexpr int UN = 16;                                   // ticks per unrolled block
    static_assert(W0 <= W1 && W1 <= W2 && W2 <= W3 && W3 <= 8, "windows must be sorted, <= 8");
    static_assert((W0 & (W0 - 1)) == 0 && (W1 & (W1 - 1)) == 0 && (W2 & (W2 - 1)) == 0 &&
                  (W3 & (W3 - 1)) == 0, "power-of-two windows");
    static_assert(W0 + W1 + W2 <= 7, "sample ring is 8 deep");
    static_assert(DOUT < UN, "flag ring is 16 deep");

    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int g = blockIdx.y;
    const size_t win = blockIdx.z;
    const int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    if (c1 <= c0) return;
    const float* x = data + win * ws_data + c;
    uint8_t* o = out + win * ws_out + c;
    const size_t Cs = (size_t)C;

    float mad = (float)med[(win * (size_t)C + c) * G + g];
    float thr0 = isnan(mad) ? INFINITY : (float)((double)mad * thr_scale);
    const int p0 = max(c0 - MAXW + 1, 0);
    const int p1 = min(c1 + MAXW - 1, L);
    const int Lp = p1 - p0;
    const int o0 = c0 - p0, o1 = c1 - p0;   // output interior in padded coordinates
    x += (size_t)p0 * Cs;
    o += (size_t)p0 * Cs;

    // thr = f64(thr0) / rho^log2(w) (flagging.py:643); T = thr * w (exact);
    // lim = largest float32 <= thr, so that for a float32 sample xf
    //   (double)xf > thr  <=>  xf > lim     and    (double)xf < -thr  <=>  xf < -lim
    double thr[4], T[4];
    float lim[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        thr[j] = (double)thr0 / fa.tf[j];
        T[j] = thr[j] * (double)W[j];
        float l = (float)thr[j];
        if ((double)l > thr[j]) {   // rounded up: step to the next float32 below
            unsigned b = __float_as_uint(l);
            b = (l > 0.0f) ? b - 1u : ((l < 0.0f) ? b + 1u : 0x80000001u);
            l = __uint_as_float(b);
        }
        lim[j] = l;
    }
    double cumlast[4] = {0.0, 0.0, 0.0, 0.0};
    double r0[W0], r1[W1], r2[W2], r3[W3];
#pragma unroll
    for (int k = 0; k < W0; k++) r0[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W1; k++) r1[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W2; k++) r2[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W3; k++) r3[k] = 0.0;
    // tick of the last positive / negative hit of stage j (far past: none in reach)
    int sp[4] = {-64, -64, -64, -64}, sn[4] = {-64, -64, -64, -64};
    // accumulated flags by position (mod 16): stage j ORs its dilated hits in,
    // stage j+1 reads them for its clamp, the last stage's position is output
    unsigned accP[UN], accN[UN];
#pragma unroll
    for (int k = 0; k < UN; k++) { accP[k] = 0; accN[k] = 0; }
    float xf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) xf[k] = 0.0f;

    const int nticks = Lp + DOUT;
    float cur[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) cur[u] = (u < Lp) ? x[(size_t)u * Cs] : 0.0f;

    auto block = [&](auto fastc, const int base) {
        constexpr bool fast = decltype(fastc)::value;
#pragma unroll
        for (int PH = 0; PH < UN; PH++) {
            const int n = base + PH;
            xf[PH & 7] = cur[PH];
            // rolling prefetch: the slot just consumed is refilled 16 ticks ahead
            cur[PH] = (n + UN < Lp) ? x[(size_t)(n + UN) * Cs] : 0.0f;
            // stages in reverse order: stage j reads the flags stage j-1 wrote
            // at the previous tick
#pragma unroll
            for (int j = 3; j >= 0; j--) {
#ifdef ST_EXP_STAGES
                if (j >= ST_EXP_STAGES) continue;
#endif
                const int w = W[j];
                const int i = n - D[j];        // ingest position
                const int e = i + 1 - w;       // emit position
                const bool ing = fast || (i >= 0 && i < Lp);
                const bool emi = fast || (e >= 0 && e < Lp);
                bool hp = false, hn = false;
                if (ing) {
                    const float xs = xf[(PH - D[j]) & 7];
                    double cl = (double)xs;
                    if (j > 0) {
                        const bool cp = (accP[(PH - D[j]) & (UN - 1)] != 0) && (xs > lim[j]);
                        const bool cn = !cp && (accN[(PH - D[j]) & (UN - 1)] != 0) && (xs < -lim[j]);
                        cl = cp ? thr[j] : (cn ? -thr[j] : cl);
                    }
                    const double cum = cumlast[j] + cl;
                    cumlast[j] = cum;
                    const int slot = (PH - D[j] + 1) & (w - 1);
                    double old;
                    if (j == 0) { old = r0[slot & (W0 - 1)]; r0[slot & (W0 - 1)] = cum; }
                    else if (j == 1) { old = r1[slot & (W1 - 1)]; r1[slot & (W1 - 1)] = cum; }
                    else if (j == 2) { old = r2[slot & (W2 - 1)]; r2[slot & (W2 - 1)] = cum; }
                    else { old = r3[slot & (W3 - 1)]; r3[slot & (W3 - 1)] = cum; }
                    const double S = cum - old;
                    const bool valid = fast || e >= 0;
                    hp = valid && (S > T[j]);
                    hn = valid && (S < -T[j]);
                }
                if (emi) {
                    bool ap, an;
                    if (w == 1) { ap = hp; an = hn; }
                    else {
                        sp[j] = hp ? n : sp[j];      // tick of the last hit
                        sn[j] = hn ? n : sn[j];
                        ap = sp[j] > n - w;
                        an = sn[j] > n - w;
                    }
                    const int es = (PH - D[j] + 1 - w) & (UN - 1);
                    accP[es] = ap ? 1u : accP[es];
                    accN[es] = an ? 1u : accN[es];
                }
            }
            const int ef = n - DOUT;
            const int fs = (PH - DOUT) & (UN - 1);
            if (fast || (ef >= o0 && ef < o1)) o[(size_t)ef * Cs] = (uint8_t)(accP[fs] | accN[fs]);
            accP[fs] = 0;
            accN[fs] = 0;
        }
    };

    for (int base = 0; base < nticks; base += UN) {
        const bool fast = base >= UN && base + UN - 1 < Lp && base - DOUT >= o0 && base + UN - 1 - DOUT < o1;
        if (fast) block(std::true_type{}, base);
        else block(std::false_type{}, base);
    }
}

// ---------------------------------------------------------------------------
// K8  _combine_flags + _unaverage_freq (flagging.py:784-918), TF layout.
// comb[t][fa] = any over t' in [t - e/2, t - e/2 + e) of (spec|time|freq).
// ---------------------------------------------------------------------------
__global__ void k_combine(const uint8_t* __restrict__ spec, const uint8_t* __restrict__ tflags,
                          const uint8_t* __restrict__ fflags, uint8_t* __restrict__ comb, int T,
                          int Fa, int Wn, int lo, int hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Fa) return;
    size_t win = blockIdx.y;
    int t = (int)(i / Fa), f = (int)(i % Fa);
    size_t base = win * (size_t)T * Fa;
    int t0 = max(t + lo, 0), t1 = min(t + hi, T);
    uint8_t v = 0;
    if (t1 > t0) {
        if (spec[(size_t)f * Wn + win]) v = 1;
        for (int tt = t0; tt < t1 && !v; tt++) {
            size_t a = base + (size_t)tt * Fa + f;
            v = (tflags[a] | fflags[a]) ? 1 : 0;
        }
    }
    comb[base + i] = v;
}

// dil[t][f] = any comb[t][f'/avg] for f' in [f - e/2, f - e/2 + e) clamped;
// per-row and per-column counts of dil (flagging.py:896-908).
// grid (ceil(F/256), T, W), block 256
__global__ void k_unaverage(const uint8_t* __restrict__ comb, uint8_t* __restrict__ dil,
                            int* __restrict__ rowcnt, int* __restrict__ colcnt, int T, int Fa,
                            int F, int avg, int lo, int hi) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    size_t win = blockIdx.z;
    int v = 0;
    if (f < F) {
        int f0 = max(f + lo, 0), f1 = min(f + hi, F);
        const uint8_t* row = comb + win * (size_t)T * Fa + (size_t)t * Fa;
        for (int ff = f0; ff < f1 && !v; ff++) v = row[ff / avg] ? 1 : 0;
        dil[win * (size_t)T * F + (size_t)t * F + f] = (uint8_t)v;
        if (v) atomicAdd(&colcnt[win * (size_t)F + f], 1);
    }
    // row count: wave ballot + one atomic per wave
    unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&rowcnt[win * (size_t)T + t], __popcll(b));
}

// out = dil | row rule | column rule | isnan(vis); iter |= out
// (flagging.py:910-918, 777-781, 1193)
template <int VD>
__global__ void k_final(const uint8_t* __restrict__ dil, const int* __restrict__ rowcnt,
                        const int* __restrict__ colcnt, const void* __restrict__ vis,
                        uint8_t* __restrict__ out, uint8_t* __restrict__ iter, int T, int F,
                        double row_limit, double col_limit, int update_iter) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * F) return;
    size_t win = blockIdx.y;
    int t = (int)(i / F), f = (int)(i % F);
    size_t a = win * (size_t)T * F + i;
    bool v = dil[a] != 0;
    v = v || ((double)rowcnt[win * (size_t)T + t] > row_limit);
    v = v || ((double)colcnt[win * (size_t)F + f] > col_limit);
    v = v || load_isnan<VD>(vis, a);
    out[a] = v ? 1 : 0;
    if (update_iter && v) iter[a] = 1;
}

// ---------------------------------------------------------------------------
// Vectorised (16 bytes of flags / 4 floats per thread) forms of the
// elementwise kernels above, used when the row lengths are multiples of 16
// (every production shape); the scalar kernels remain the general fallback.
// Flags are 0/1 bytes, so byte-wise OR is a plain bitwise OR of the words.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint4 or4(uint4 a, uint4 b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
// per byte: x != 0 ? 1 : 0 (no cross-byte carries)
__device__ __forceinline__ unsigned nz_bytes(unsigned x) {
    unsigned t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ((t | x) & 0x80808080u) >> 7;
}

template <int VD>
__global__ void k_prepare4(const void* __restrict__ vis, const uint8_t* __restrict__ iflags,
                           float* __restrict__ data, uint8_t* __restrict__ flags, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 samples, flat over the batch
    if (i >= n4) return;
    uchar4 f = reinterpret_cast<const uchar4*>(iflags)[i];
    float a[4];
    if (VD == TRI_VIS_C64) {
        float4 z0 = reinterpret_cast<const float4*>(vis)[2 * i];
        float4 z1 = reinterpret_cast<const float4*>(vis)[2 * i + 1];
        a[0] = tri_hypotf(z0.x, z0.y); a[1] = tri_hypotf(z0.z, z0.w);
        a[2] = tri_hypotf(z1.x, z1.y); a[3] = tri_hypotf(z1.z, z1.w);
    } else {
        float4 z = reinterpret_cast<const float4*>(vis)[i];
        a[0] = fabsf(z.x); a[1] = fabsf(z.y); a[2] = fabsf(z.z); a[3] = fabsf(z.w);
    }
    unsigned char fl[4] = {f.x, f.y, f.z, f.w};
    float o[4];
    unsigned char of[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        bool bad = fl[k] != 0 || isnan(a[k]);
        // factor 1: sum = 0 + a, count 1, a / 1.0f = a (flagging.py:858-870)
        o[k] = bad ? 0.0f : (0.0f + a[k]) / 1.0f;
        of[k] = bad ? 1 : 0;
    }
    reinterpret_cast<float4*>(data)[i] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<uchar4*>(flags)[i] = make_uchar4(of[0], of[1], of[2], of[3]);
}

// OP 0: b = a   OP 1: b |= a   OP 2: b = (a != 0)     (16 bytes per thread)
template <int OP>
__global__ void k_u8_op16(const uint8_t* __restrict__ a, uint8_t* __restrict__ b, size_t n16per,
                          size_t ws_a, size_t ws_b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16per) return;
    size_t win = blockIdx.y;
    uint4 va = reinterpret_cast<const uint4*>(a + win * ws_a)[i];
    uint4* pb = reinterpret_cast<uint4*>(b + win * ws_b) + i;
    if (OP == 0) *pb = va;
    else if (OP == 1) *pb = or4(*pb, va);
    else *pb = make_uint4(nz_bytes(va.x), nz_bytes(va.y), nz_bytes(va.z), nz_bytes(va.w));
}

// spectrum flags [Fa][Wn] -> rows [Wn][Fa] (tiny), so that the per-window
// kernels below can read 16 channels at a time
__global__ void k_spec_rows(const uint8_t* __restrict__ spec, uint8_t* __restrict__ rows, int Fa, int Wn) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Fa * Wn) return;
    int w = (int)(i / Fa), f = (int)(i % Fa);
    rows[i] = spec[(size_t)f * Wn + w];
}

// flags[w][t][f..f+15] |= spec_rows[w][f..f+15]
__global__ void k_or_spec16(uint8_t* __restrict__ flags, const uint8_t* __restrict__ spec_rows, int T, int Fa16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Fa16) return;
    size_t win = blockIdx.y;
    int f16 = (int)(i % Fa16);
    uint4 sp = reinterpret_cast<const uint4*>(spec_rows + win * (size_t)Fa16 * 16)[f16];
    uint4* pf = reinterpret_cast<uint4*>(flags + win * (size_t)T * Fa16 * 16) + i;
    *pf = or4(*pf, sp);
}

// _combine_flags (flagging.py:784-816), 16 channels per thread
__global__ void k_combine16(const uint8_t* __restrict__ spec_rows, const uint8_t* __restrict__ tflags,
                            const uint8_t* __restrict__ fflags, uint8_t* __restrict__ comb, int T,
                            int Fa16, int lo, int hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Fa16) return;
    size_t win = blockIdx.y;
    int t = (int)(i / Fa16), f16 = (int)(i % Fa16);
    size_t base = win * (size_t)T * Fa16;
    int t0 = max(t + lo, 0), t1 = min(t + hi, T);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t1 > t0) {
        v = reinterpret_cast<const uint4*>(spec_rows + win * (size_t)Fa16 * 16)[f16];
        const uint4* tp = reinterpret_cast<const uint4*>(tflags) + base;
        const uint4* fp = reinterpret_cast<const uint4*>(fflags) + base;
        for (int tt = t0; tt < t1; tt++) {
            size_t a = (size_t)tt * Fa16 + f16;
            v = or4(v, or4(tp[a], fp[a]));
        }
    }
    reinterpret_cast<uint4*>(comb)[base + i] = v;
}

// _unaverage_freq (flagging.py:896-908) for average_freq == 1, frequency
// dilation over [f + LO, f + LO + E), 16 channels per thread; row counts by
// popcount + one atomic per wave; column counts by k_colcount.
// grid (ceil(F16/64), T, W), block 64
template <int LO, int E>
__global__ void k_unaverage16(const uint8_t* __restrict__ comb, uint8_t* __restrict__ dil,
                              int* __restrict__ rowcnt, int T, int F16) {
    int f16 = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    size_t win = blockIdx.z;
    int cnt = 0;
    if (f16 < F16) {
        const uint4* row = reinterpret_cast<const uint4*>(comb) + (win * (size_t)T + t) * F16;
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 p = f16 > 0 ? row[f16 - 1] : z;
        uint4 c = row[f16];
        uint4 n = f16 + 1 < F16 ? row[f16 + 1] : z;
        unsigned w[12] = {p.x, p.y, p.z, p.w, c.x, c.y, c.z, c.w, n.x, n.y, n.z, n.w};
        unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            unsigned v = 0;
#pragma unroll
            for (int sft = LO; sft < LO + E; sft++) {
                const int idx = 16 + k + sft;   // byte index into the 48-byte window
                v |= (w[idx >> 2] >> (8 * (idx & 3))) & 0xFFu;
            }
            o[k >> 2] |= (v & 1u) << (8 * (k & 3));
        }
        reinterpret_cast<uint4*>(dil)[(win * (size_t)T + t) * F16 + f16] = make_uint4(o[0], o[1], o[2], o[3]);
        cnt = __popc(o[0]) + __popc(o[1]) + __popc(o[2]) + __popc(o[3]);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&rowcnt[win * (size_t)T + t], cnt);
}

// column counts of a [T][F] 0/1 byte image: one thread per 4 columns
// grid (ceil(F4/256), W)
__global__ void k_colcount(const uint8_t* __restrict__ dil, int* __restrict__ colcnt, int T, int F4) {
    int f4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (f4 >= F4) return;
    size_t win = blockIdx.y;
    const unsigned* p = reinterpret_cast<const unsigned*>(dil) + win * (size_t)T * F4 + f4;
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int t = 0;
    for (; t + 8 <= T; t += 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(size_t)(t + u) * F4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            c0 += v[u] & 0xFFu; c1 += (v[u] >> 8) & 0xFFu; c2 += (v[u] >> 16) & 0xFFu; c3 += v[u] >> 24;
        }
    }
    for (; t < T; t++) {
        unsigned v = p[(size_t)t * F4];
        c0 += v & 0xFFu; c1 += (v >> 8) & 0xFFu; c2 += (v >> 16) & 0xFFu; c3 += v >> 24;
    }
    reinterpret_cast<int4*>(colcnt)[win * (size_t)F4 + f4] = make_int4((int)c0, (int)c1, (int)c2, (int)c3);
}

// k_final, 16 samples per thread
template <int VD>
__global__ void k_final16(const uint8_t* __restrict__ dil, const int* __restrict__ rowcnt,
                          const int* __restrict__ colcnt, const void* __restrict__ vis,
                          uint8_t* __restrict__ out, uint8_t* __restrict__ iter, int T, int F16,
                          double row_limit, double col_limit, int update_iter) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * F16) return;
    size_t win = blockIdx.y;
    int t = (int)(i / F16), f16 = (int)(i % F16);
    size_t a16 = win * (size_t)T * F16 + i;
    uint4 d = reinterpret_cast<const uint4*>(dil)[a16];
    unsigned dw[4] = {d.x, d.y, d.z, d.w};
    bool rowall = (double)rowcnt[win * (size_t)T + t] > row_limit;
    const int4* cc = reinterpret_cast<const int4*>(colcnt + win * (size_t)F16 * 16) + (size_t)f16 * 4;
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int4 c4 = cc[q];
        int cv[4] = {c4.x, c4.y, c4.z, c4.w};
        unsigned nanb = 0;
        if (VD == TRI_VIS_C64) {
            const float4* vp = reinterpret_cast<const float4*>(vis) + (a16 * 16 + q * 4) / 2;
            float4 z0 = vp[0], z1 = vp[1];
            nanb = ((isnan(z0.x) || isnan(z0.y)) ? 1u : 0u) | ((isnan(z0.z) || isnan(z0.w)) ? 0x100u : 0u) |
                   ((isnan(z1.x) || isnan(z1.y)) ? 0x10000u : 0u) | ((isnan(z1.z) || isnan(z1.w)) ? 0x1000000u : 0u);
        } else {
            float4 z = reinterpret_cast<const float4*>(vis)[(a16 * 16 + q * 4) / 4];
            nanb = (isnan(z.x) ? 1u : 0u) | (isnan(z.y) ? 0x100u : 0u) | (isnan(z.z) ? 0x10000u : 0u) |
                   (isnan(z.w) ? 0x1000000u : 0u);
        }
        unsigned colb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) colb |= ((double)cv[k] > col_limit ? 1u : 0u) << (8 * k);
        o[q] = rowall ? 0x01010101u : (dw[q] | colb | nanb);
    }
    uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<uint4*>(out)[a16] = ov;
    if (update_iter) {
        uint4* ip = reinterpret_cast<uint4*>(iter) + a16;
        *ip = or4(*ip, ov);
    }
}

template <int MODE>
__global__ void k_masked_div4(const float* __restrict__ w, float* __restrict__ o,
                              const float* __restrict__ data, size_t n4per, size_t ws_wo, size_t ws_data,
                              float denom, uint8_t* __restrict__ nanflag, int C) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4per) return;
    size_t win = blockIdx.y;
    float4 wv = reinterpret_cast<const float4*>(w + win * ws_wo)[i];
    float4* po = reinterpret_cast<float4*>(o + win * ws_wo) + i;
    float4 ov = *po;
    if (denom != 0.0f) {   // deferred flagging.py:419
        wv = make_float4(wv.x / denom, wv.y / denom, wv.z / denom, wv.w / denom);
        ov = make_float4(ov.x / denom, ov.y / denom, ov.z / denom, ov.w / denom);
    }
    float bg[4] = {(wv.x == 0.0f) ? NAN : ov.x / wv.x, (wv.y == 0.0f) ? NAN : ov.y / wv.y,
                   (wv.z == 0.0f) ? NAN : ov.z / wv.z, (wv.w == 0.0f) ? NAN : ov.w / wv.w};
    if (MODE == 0 && nanflag) {
        int cb = (int)((i * 4) % C);   // C % 4 == 0: the four samples are columns cb .. cb + 3
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (isnan(bg[k])) nanflag[win * (size_t)C + cb + k] = 1;
    }
    if (MODE == 1) {
        float4 dv = reinterpret_cast<const float4*>(data + win * ws_data)[i];
        bg[0] = fabsf(dv.x - bg[0]); bg[1] = fabsf(dv.y - bg[1]);
        bg[2] = fabsf(dv.z - bg[2]); bg[3] = fabsf(dv.w - bg[3]);
    }
    *po = make_float4(bg[0], bg[1], bg[2], bg[3]);
}

__global__ void k_sub4(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                       size_t n4per, size_t ws_a, size_t ws_b, size_t ws_o) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4per) return;
    size_t win = blockIdx.y;
    float4 x = reinterpret_cast<const float4*>(a + win * ws_a)[i];
    float4 y = reinterpret_cast<const float4*>(b + win * ws_b)[i];
    reinterpret_cast<float4*>(out + win * ws_o)[i] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
}

// k_reject<true>, 4 samples per thread (C % 4 == 0 keeps a group in one line)
__global__ void k_reject4(const float* __restrict__ resid, uint8_t* __restrict__ flags,
                          const double* __restrict__ med, const int* __restrict__ chunk_of,
                          double scale, int C4, int G, size_t n4per, size_t ws_resid, size_t ws_flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4per) return;
    size_t win = blockIdx.y;
    int l = (int)(i / C4);
    double thr = med[win * G + chunk_of[l]] * scale;
    float4 rv = reinterpret_cast<const float4*>(resid + win * ws_resid)[i];
    uchar4* pf = reinterpret_cast<uchar4*>(flags + win * ws_flags) + i;
    uchar4 f = *pf;
    if ((double)rv.x > thr) f.x = 1;
    if ((double)rv.y > thr) f.y = 1;
    if ((double)rv.z > thr) f.z = 1;
    if ((double)rv.w > thr) f.w = 1;
    *pf = f;
}

// ---------------------------------------------------------------------------
// pack / unpack (packing.py:243-278, 369-415) with a precomputed row map
// ---------------------------------------------------------------------------
__global__ void k_fill_windows(float2* __restrict__ vis, uint8_t* __restrict__ flags, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vis[i] = make_float2(NAN, NAN);
    flags[i] = 1;
}

// one thread per (row, chan); loops over corr. grid (ceil(nchan/256), rows)
__global__ void k_pack(const float2* __restrict__ data, const uint8_t* __restrict__ flag,
                       const int32_t* __restrict__ row_bl, const int32_t* __restrict__ row_time,
                       int nchan, int ncorr, int nbl, int ntime, float2* __restrict__ vw,
                       uint8_t* __restrict__ fw) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    size_t r = blockIdx.y;
    if (f >= nchan) return;
    int bl = row_bl[r], t = row_time[r];
    if (bl < 0 || bl >= nbl || t < 0 || t >= ntime) return;
    for (int c = 0; c < ncorr; c++) {
        size_t i = (r * nchan + f) * (size_t)ncorr + c;
        size_t o = (((size_t)bl * ncorr + c) * ntime + t) * (size_t)nchan + f;
        vw[o] = data[i];
        fw[o] = flag[i];
    }
}

__global__ void k_unpack(const uint8_t* __restrict__ fw, const int32_t* __restrict__ row_bl,
                         const int32_t* __restrict__ row_time, int nchan, int ncorr, int nbl,
                         int ntime, uint8_t* __restrict__ out) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    size_t r = blockIdx.y;
    if (f >= nchan) return;
    int bl = row_bl[r], t = row_time[r];
    bool ok = !(bl < 0 || bl >= nbl || t < 0 || t >= ntime);
    for (int c = 0; c < ncorr; c++) {
        size_t i = (r * nchan + f) * (size_t)ncorr + c;
        out[i] = ok ? fw[(((size_t)bl * ncorr + c) * ntime + t) * (size_t)nchan + f] : 0;
    }
}

// ===========================================================================
// host side
// ===========================================================================
namespace {

struct Bump {
    char* base;
    size_t cap, off;
    bool dry;
    Bump(void* b, size_t c, bool d) : base((char*)b), cap(c), off(0), dry(d) {}
    template <typename T>
    T* get(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        size_t o = off;
        off += bytes;
        if (dry) return nullptr;
        return reinterpret_cast<T*>(base + o);
    }
};

// int(0.5 * sqrt(12 sigma^2 / passes + 1)), flagging.py:451
int64_t box_radius(double sigma) {
    return (int64_t)(0.5 * std::sqrt(12.0 * (sigma * sigma) / 4.0 + 1.0));
}

// float32(d) ** 4 as numba evaluates it: square-and-multiply in float32
// (numba cpython/numbers.py:207-245)
float box_denominator(int64_t r) {
    volatile float a = (float)(2 * r + 1);
    volatile float a2 = a * a;
    volatile float a4 = a2 * a2;
    volatile float res = 1.0f * a4;
    return res;
}

struct Plan {
    int64_t T, F, Fa, avg, G;
    int64_t r0max, r1max;   // largest box radii along time / frequency
    int64_t PT, PF;         // padded line lengths
    int nit;                // background_iterations
    bool vec;               // 16-byte vectorised elementwise kernels usable
    int64_t maxchunk;       // longest frequency chunk (averaged channels)
    StWin swT, swF;
};

int make_stwin(const int64_t* w, int64_t nw, double rho, StWin* s) {
    memset(s, 0, sizeof(*s));
    if (nw <= 0) return set_err(TRI_EINVAL, "no SumThreshold window fits the data (np.max of an empty window list)");
    if (nw > TRI_MAX_WINDOWS) return set_err(TRI_EINVAL, "too many windows");
    s->nw = (int)nw;
    int off = 0, d = 0, maxw = 0;
    for (int j = 0; j < nw; j++) {
        if (w[j] <= 0) return set_err(TRI_EINVAL, "SumThreshold window of size %lld (the reference fails here with a broadcasting ValueError)", (long long)w[j]);
        s->w[j] = (int)w[j];
        s->tf[j] = std::pow(rho, std::log2((double)w[j]));
        s->scale[j] = (double)(float)(1.0 / (double)w[j]);
        s->ringoff[j] = off;
        off += (int)w[j];
        s->delay[j] = d;
        d += (int)w[j] - 1;
        maxw = std::max(maxw, (int)w[j]);
    }
    s->delay[nw] = d;
    s->ringtot = off;
    s->acccap = d + 1;
    s->maxw = maxw;
    return TRI_OK;
}

int make_plan(int64_t T, int64_t F, const tri_params* p, Plan* pl) {
    if (T <= 0 || F <= 0) return set_err(TRI_EINVAL, "empty window (ntime=%lld, nchan=%lld)", (long long)T, (long long)F);
    if (p->average_freq < 1 || p->average_freq > 255) return set_err(TRI_EUNSUPPORTED, "average_freq must be in [1, 255]");
    if (p->n_chunk_ends < 2 || !p->chunk_ends) return set_err(TRI_EINVAL, "freq_chunks must be >= 1");
    if (p->background_iterations < 0 || p->num_major_iterations < 0) return set_err(TRI_EINVAL, "negative iteration count");
    pl->T = T; pl->F = F; pl->avg = p->average_freq;
    pl->Fa = (F + pl->avg - 1) / pl->avg;
    pl->G = p->n_chunk_ends - 1;
    for (int64_t g = 0; g < p->n_chunk_ends; g++) {
        if (p->chunk_ends[g] < 0 || p->chunk_ends[g] > pl->Fa || (g > 0 && p->chunk_ends[g] < p->chunk_ends[g - 1]))
            return set_err(TRI_EINVAL, "freq chunk ends must be non-decreasing within [0, averaged channels]");
    }
    if (p->chunk_ends[0] != 0 || p->chunk_ends[pl->G] != pl->Fa) return set_err(TRI_EINVAL, "freq chunk ends must start at 0 and end at the averaged channel count");
    pl->nit = (int)p->background_iterations;
    pl->maxchunk = 0;
    for (int64_t g = 0; g + 1 < p->n_chunk_ends; g++) pl->maxchunk = std::max(pl->maxchunk, p->chunk_ends[g + 1] - p->chunk_ends[g]);
    pl->vec = (pl->avg == 1 && F % 16 == 0 && T % 4 == 0);
    int64_t emax = std::max<int64_t>(pl->nit, 1);
    pl->r0max = box_radius((double)emax * p->spike_width_time);
    pl->r1max = box_radius((double)emax * p->spike_width_freq);
    if (pl->r0max > (1 << 20) || pl->r1max > (1 << 20)) return set_err(TRI_EUNSUPPORTED, "spike width too large");
    pl->PT = T + 4 * pl->r0max;
    pl->PF = pl->Fa + 4 * pl->r1max;
    int rc = make_stwin(p->windows_time, p->n_windows_time, p->rho, &pl->swT);
    if (rc) return rc;
    rc = make_stwin(p->windows_freq, p->n_windows_freq, p->rho, &pl->swF);
    if (rc) return rc;
    if (T * pl->Fa >= ((int64_t)1 << 31) || T * F >= ((int64_t)1 << 31))
        return set_err(TRI_EUNSUPPORTED, "a single window must hold fewer than 2^31 samples");
    return TRI_OK;
}

// Workspace carve-up for a batch of Wb windows.  `dry` only measures.
#define TRI_MAX_CHUNKS 255
struct ChunkTab {
    int n;                       // number of chunk ends
    int64_t ends[TRI_MAX_CHUNKS + 1];
};

struct Ws {
    // per-window images
    float *dataTF, *dataFT, *Aw, *Ao, *Bw, *Bo;
    uint8_t *iter, *flagsTF, *flagsFT, *bgfTF, *bgfFT, *tflTF, *fflFT, *fflTF, *comb, *dil;
    int *rowcnt, *colcnt;
    double* med;      // medians: max(Fa, T*G) per window
    // spectrum layout [Fa][Wb]
    float *sdata, *sw, *so, *sres;
    uint8_t *sflags, *sbgf, *sout, *srows;   // srows: sout as rows [Wb][Fa]
    double* smed;     // [Wb][G]
    // SumThreshold scratch
    double* ring;
    uint8_t* acc;
    // small tables (device)
    int64_t* d_chunk_ends;   // G+1, averaged-channel units
    int64_t* d_tends;        // {0, T}
    int64_t *segC_start, *segC_len;   // chunk segments in line units
    int64_t *segB_start, *segB_len;   // chunk blocks of the FT layout (x T)
    int64_t *segT_start, *segT_len;   // the single segment [0, T)
    int* d_chunk_of;         // Fa
    size_t total;
};

static void carve(const Plan& pl, int64_t Wb, void* base, size_t cap, bool dry, Ws* ws) {
    Bump b(base, cap, dry);
    size_t W = (size_t)Wb, T = (size_t)pl.T, F = (size_t)pl.F, Fa = (size_t)pl.Fa, G = (size_t)pl.G;
    size_t N = T * Fa;
    // small tables first so that their addresses do not depend on the batch
    ws->d_chunk_ends = b.get<int64_t>(G + 1);
    ws->d_tends = b.get<int64_t>(2);
    ws->segC_start = b.get<int64_t>(G);
    ws->segC_len = b.get<int64_t>(G);
    ws->segB_start = b.get<int64_t>(G);
    ws->segB_len = b.get<int64_t>(G);
    ws->segT_start = b.get<int64_t>(1);
    ws->segT_len = b.get<int64_t>(1);
    ws->d_chunk_of = b.get<int>(Fa);
    ws->iter = b.get<uint8_t>(W * T * F);
    ws->dataTF = b.get<float>(W * N);
    ws->dataFT = b.get<float>(W * N);
    ws->Aw = b.get<float>(W * (size_t)pl.PT * Fa);
    ws->Ao = b.get<float>(W * (size_t)pl.PT * Fa);
    ws->Bw = b.get<float>(W * (size_t)pl.PF * T);
    ws->Bo = b.get<float>(W * (size_t)pl.PF * T);
    ws->flagsTF = b.get<uint8_t>(W * N);
    ws->flagsFT = b.get<uint8_t>(W * N);
    ws->bgfTF = b.get<uint8_t>(W * N);
    ws->bgfFT = b.get<uint8_t>(W * N);
    ws->tflTF = b.get<uint8_t>(W * N);
    ws->fflFT = b.get<uint8_t>(W * N);
    ws->fflTF = b.get<uint8_t>(W * N);
    ws->comb = b.get<uint8_t>(W * N);
    ws->dil = b.get<uint8_t>(W * T * F);
    ws->rowcnt = b.get<int>(W * T);
    ws->colcnt = b.get<int>(W * F);
    ws->med = b.get<double>(W * std::max(Fa, T * G));
    size_t PS = (size_t)pl.PF;  // spectrum padded length
    ws->sdata = b.get<float>(Fa * W);
    ws->sw = b.get<float>(PS * W);
    ws->so = b.get<float>(PS * W);
    ws->sres = b.get<float>(Fa * W);
    ws->sflags = b.get<uint8_t>(Fa * W);
    ws->sbgf = b.get<uint8_t>(Fa * W);
    ws->sout = b.get<uint8_t>(Fa * W);
    ws->srows = b.get<uint8_t>(Fa * W);
    ws->smed = b.get<double>(W * G);
    // SumThreshold scratch: one slot set per (window, chunk, column) thread
    size_t thrT = W * Fa, thrF = W * T * G;
    size_t ringn = std::max(thrT * (size_t)pl.swT.ringtot, thrF * (size_t)pl.swF.ringtot);
    size_t accn = std::max(thrT * (size_t)pl.swT.acccap, thrF * (size_t)pl.swF.acccap);
    ws->ring = b.get<double>(ringn);
    ws->acc = b.get<uint8_t>(accn);
    ws->total = b.off;
}

static inline dim3 grid1(size_t n, size_t W) { return dim3((unsigned)cdiv((int64_t)n, 256), (unsigned)W, 1); }

}  // namespace

__global__ void k_tables(ChunkTab ct, int T, int Fa, int64_t* chunk_ends, int64_t* tends,
                         int64_t* segC_start, int64_t* segC_len, int64_t* segB_start,
                         int64_t* segB_len, int64_t* segT_start, int64_t* segT_len,
                         int* chunk_of) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int G = ct.n - 1;
    if (i <= G) chunk_ends[i] = ct.ends[i];
    if (i < G) {
        segC_start[i] = ct.ends[i];
        segC_len[i] = ct.ends[i + 1] - ct.ends[i];
        segB_start[i] = ct.ends[i] * T;
        segB_len[i] = (ct.ends[i + 1] - ct.ends[i]) * T;
    }
    if (i == 0) { tends[0] = 0; tends[1] = T; segT_start[0] = 0; segT_len[0] = T; }
    if (i < Fa) {
        int g = 0;
        // chunk containing channel i: ends[g] <= i < ends[g+1] (skips empty chunks)
        while (g < G - 1 && !(i >= ct.ends[g] && i < ct.ends[g + 1])) g++;
        chunk_of[i] = g;
    }
}

// copies column `col` of a [L][C] array into a contiguous vector (debug taps)
__global__ void k_gather_col_f32(const float* __restrict__ a, float* __restrict__ out, int L, int C, int col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) out[i] = a[(size_t)i * C + col];
}
__global__ void k_gather_col_u8(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, int L, int C, int col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) out[i] = a[(size_t)i * C + col];
}
__global__ void k_normalise_flags(const uint8_t* __restrict__ a, uint8_t* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] ? 1 : 0;
}

extern "C" size_t tri_workspace_bytes(int64_t batch_windows, int64_t ntime, int64_t nchan,
                                      const tri_params* p) {
    Plan pl;
    if (!p || batch_windows <= 0 || make_plan(ntime, nchan, p, &pl) != TRI_OK) return 0;
    Ws ws;
    carve(pl, batch_windows, nullptr, 0, true, &ws);
    return ws.total;
}

extern "C" int tri_prepare_params(int64_t ntime, int64_t nchan, double outlier_nsigma,
                                  const double* windows_time, int64_t n_windows_time,
                                  const double* windows_freq, int64_t n_windows_freq,
                                  double background_reject, int64_t background_iterations,
                                  double spike_width_time, double spike_width_freq,
                                  int64_t time_extend, int64_t freq_extend, int64_t freq_chunks,
                                  int64_t average_freq, double flag_all_time_frac,
                                  double flag_all_freq_frac, double rho,
                                  int64_t num_major_iterations, int64_t* chunk_ends_buf,
                                  int64_t chunk_cap, tri_params* out) {
    if (!out || !chunk_ends_buf) return set_err(TRI_EINVAL, "NULL output");
    if (ntime <= 0 || nchan <= 0) return set_err(TRI_EINVAL, "empty window");
    if (average_freq < 1) return set_err(TRI_EINVAL, "average_freq must be >= 1");
    if (freq_chunks < 1) return set_err(TRI_EINVAL, "freq_chunks must be >= 1");
    if (chunk_cap < freq_chunks + 1) return set_err(TRI_EINVAL, "chunk_ends_buf too small");
    if (n_windows_time > TRI_MAX_WINDOWS || n_windows_freq > TRI_MAX_WINDOWS || n_windows_time < 0 || n_windows_freq < 0)
        return set_err(TRI_EINVAL, "at most %d windows per axis", TRI_MAX_WINDOWS);
    memset(out, 0, sizeof(*out));
    int64_t fa = (nchan + average_freq - 1) / average_freq;
    // flagging.py:1160-1162: float32 ceil, float32 division, truncation, unique
    std::vector<int64_t> wf;
    for (int64_t i = 0; i < n_windows_freq; i++) {
        float v = std::ceil((float)windows_freq[i]) / (float)average_freq;
        wf.push_back((int64_t)v);
    }
    std::sort(wf.begin(), wf.end());
    wf.erase(std::unique(wf.begin(), wf.end()), wf.end());
    // flagging.py:1172-1173: linspace(0, fa, freq_chunks + 1).astype(int)
    double step = (double)fa / (double)freq_chunks;
    for (int64_t i = 0; i <= freq_chunks; i++) chunk_ends_buf[i] = (int64_t)((double)i * step);
    chunk_ends_buf[freq_chunks] = fa;
    // flagging.py:1176-1179
    out->n_windows_time = 0;
    for (int64_t i = 0; i < n_windows_time; i++)
        if (windows_time[i] <= (double)ntime) out->windows_time[out->n_windows_time++] = (int64_t)windows_time[i];
    out->n_windows_freq = 0;
    for (size_t i = 0; i < wf.size(); i++)
        if (wf[i] <= fa) out->windows_freq[out->n_windows_freq++] = wf[i];
    for (int64_t i = 0; i < out->n_windows_freq; i++)
        if (out->windows_freq[i] <= 0)
            return set_err(TRI_EINVAL, "windows_freq / average_freq produced a window of size %lld; the reference fails here (operands could not be broadcast together, flagging.py:663)", (long long)out->windows_freq[i]);
    for (int64_t i = 0; i < out->n_windows_time; i++)
        if (out->windows_time[i] <= 0) return set_err(TRI_EINVAL, "windows_time must be positive");
    out->outlier_nsigma = outlier_nsigma;
    out->background_reject = background_reject;
    out->background_iterations = background_iterations;
    out->spike_width_time = spike_width_time;
    out->spike_width_freq = spike_width_freq;
    out->time_extend = time_extend;
    out->freq_extend = freq_extend;
    out->n_chunk_ends = freq_chunks + 1;
    out->chunk_ends = chunk_ends_buf;
    out->average_freq = average_freq;
    out->flag_all_time_frac = flag_all_time_frac;
    out->flag_all_freq_frac = flag_all_freq_frac;
    out->rho = rho;
    out->num_major_iterations = num_major_iterations;
    return TRI_OK;
}

namespace {

struct Debug {
    float* f32;     // [spec_resid Fa][background FT Fa*T][residual TF T*Fa]
    uint8_t* u8;    // [spec_flags Fa][time_flags TF][freq_flags TF]
};

struct Run {
    hipStream_t st;
    Plan pl;
    Ws ws;
    const tri_params* p;
    int64_t Wb;      // windows in this batch
    Debug* dbg;      // taps of window 0 (tests only), or NULL
};

// The register cascade covers windows exactly (1,2,4,8) in that order; the
// environment variable TRI_ST_GENERIC=1 forces the generic kernel (tests).
bool st_use_fused(const StWin& sw) {
    static const bool force_generic = [] {
        const char* e = getenv("TRI_ST_GENERIC");
        return e && e[0] == '1';
    }();
    if (force_generic) return false;
    return sw.nw == 4 && sw.w[0] == 1 && sw.w[1] == 2 && sw.w[2] == 4 && sw.w[3] == 8;
}

int launch_median(const Run& r, const float* data, const uint8_t* flags, double* med, size_t WSd,
                  size_t WSf, size_t RS, size_t ES, const int64_t* seg_start,
                  const int64_t* seg_len, int R, int G, int64_t W, int64_t max_len, bool vec_ok = false,
                  bool rows_aligned = false, bool segs_aligned = false) {
    if ((int64_t)R * G <= 0 || W <= 0) return TRI_OK;
    if ((int64_t)R * G > 0x7FFFFFFF || W > 65535) return set_err(TRI_EUNSUPPORTED, "median grid too large");
    // segments of contiguous 4-aligned rows can be loaded 16 bytes at a time
    // (misaligned segment ends are masked, costing up to 3 extra slots)
    const bool row4 = ES == 1 && RS % 4 == 0 && WSd % 4 == 0 && WSf % 4 == 0 &&
                      ((uintptr_t)data % 16 == 0) && ((uintptr_t)flags % 4 == 0) && rows_aligned;
    const int64_t slack = segs_aligned ? 0 : 3;   // misaligned segment starts cost up to 3 masked slots
    if (max_len + slack <= 64 * 8 && row4)
        hipLaunchKernelGGL((k_median_wave<8, true>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    else if (max_len <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, false>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    else if (max_len + slack <= 64 * MW_K && row4)
        hipLaunchKernelGGL((k_median_wave<MW_K, true>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    else if (max_len <= 64 * MW_K)
        hipLaunchKernelGGL((k_median_wave<MW_K, false>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    else if (vec_ok)
        hipLaunchKernelGGL(k_median<true>, dim3((unsigned)(R * G), (unsigned)W), dim3(256), 0, r.st, data,
                           flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    else
        hipLaunchKernelGGL(k_median<false>, dim3((unsigned)(R * G), (unsigned)W), dim3(256), 0, r.st, data,
                           flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
    LAUNCHCHK();
    return TRI_OK;
}

// LDS budget of the single-sweep filter: 4 stages x 2r slots x BT threads x 4 B.
// Returns the block size to use, or 0 when the radius is too large and the
// in-place multi-pass kernel must be used instead.
// Radius limits of the single-sweep kernels.  r <= LDS_R_MAX: K4b with 256
// threads (4 rings per thread); r <= LANE4_R_MAX: K4c (one ring per thread);
// beyond: the in-place multi-pass kernel.
#define LDS_R_MAX 20   // measured: 4 rings per thread win up to here (256 threads to r = 10, 128 beyond)
#define LANE4_R_MAX 80
int colfilter_lds_block(int rad, int C) {
    static const bool disabled = [] {
        const char* e = getenv("TRI_FILTER_MULTIPASS");
        return e && e[0] == '1';
    }();
    static const bool no_lane4 = [] {
        const char* e = getenv("TRI_FILTER_NO_LANE4");
        return e && e[0] == '1';
    }();
    if (disabled || rad <= 0) return 0;
    if (no_lane4) {
        int bt = rad <= 10 ? 256 : (rad <= 20 ? 128 : (rad <= 40 ? 64 : 0));
        while (bt > 64 && bt / 2 >= C) bt /= 2;
        return bt;
    }
    if (rad <= LDS_R_MAX) {
        int bt = rad <= 10 ? 256 : 128;
        while (bt > 64 && bt / 2 >= C) bt /= 2;
        return bt;
    }
    return rad <= LANE4_R_MAX ? 64 : 0;
}
inline bool colfilter_use_lane4(int rad) {
    static const bool no_lane4 = [] {
        const char* e = getenv("TRI_FILTER_NO_LANE4");
        return e && e[0] == '1';
    }();
    return !no_lane4 && rad > LDS_R_MAX && rad <= LANE4_R_MAX;
}

// One axis of masked_gaussian_filter's two box filters (weight image and
// data image) on a column-layout image pair.
//   srcmode 0: images are built on the fly from (srcData, srcFlags) [n][C]
//   srcmode 1: images are float arrays; for the multi-pass kernel they sit in
//              rows [4r, 4r+n) of bufW / bufO, for the LDS kernel in rows [0,n)
// Output: rows [0,n) of dstW / dstO.
// `deferred_denom` (optional): when the single-sweep kernel is used the final
// division by float32(d)**4 is left to the consumer (transpose / masked_div) and
// *deferred_denom receives the denominator; otherwise it is set to 0.
int launch_colfilter(const Run& r, int srcmode, float* bufW, float* bufO, const float* srcData,
                     const uint8_t* srcFlags, float* dstW, float* dstO, int n, int C, int rad,
                     size_t bws, size_t sws, size_t dws, int64_t W, float* deferred_denom = nullptr,
                     bool transposed_out = false) {
    float denom = box_denominator(rad);
    int bt = colfilter_lds_block(rad, C);
    if (deferred_denom) *deferred_denom = 0.0f;
    if (bt > 0 && colfilter_use_lane4(rad) && !transposed_out) {
        size_t lds = (size_t)2 * rad * 64 * sizeof(float);
        dim3 grid((unsigned)cdiv(C, 16), (unsigned)W, 2);
        // thread-safe one-time setup (C++11 static initialisation)
        static const hipError_t attr4 = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lane4<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lane4<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lane4<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lane4<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lane4<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return e;
        }();
        HIPCHK(attr4);
        if (srcmode == 0) {
            // byte flags + data (spectrum path): build-free variant is not provided;
            // fall through to the 4-ring kernel below when it fits, else multi-pass
        } else if (srcmode == 2) {
            if (deferred_denom) {
                *deferred_denom = denom;
                hipLaunchKernelGGL((k_colfilter_lane4<2, false>), grid, dim3(64), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            } else {
                hipLaunchKernelGGL((k_colfilter_lane4<2, true>), grid, dim3(64), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            }
            LAUNCHCHK();
            return TRI_OK;
        } else if (deferred_denom) {
            *deferred_denom = denom;
            hipLaunchKernelGGL((k_colfilter_lane4<1, false>), grid, dim3(64), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
            LAUNCHCHK();
            return TRI_OK;
        } else {
            hipLaunchKernelGGL((k_colfilter_lane4<1, true>), grid, dim3(64), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
            LAUNCHCHK();
            return TRI_OK;
        }
    }
    if (bt > 0 && colfilter_use_lane4(rad) && srcmode == 0) {
        // spectrum path with a medium radius: 4-ring kernel with a small block if it fits
        bt = rad <= 20 ? 128 : (rad <= 40 ? 64 : 0);
        while (bt > 64 && bt / 2 >= C) bt /= 2;
    }
    if (bt > 0) {
        size_t lds = (size_t)4 * 2 * rad * bt * sizeof(float);
        dim3 grid((unsigned)cdiv(C, bt), (unsigned)W, 2);
        static const hipError_t attr_set = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds<2, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return e;
        }();
        HIPCHK(attr_set);
        if (srcmode == 2) {
            if (deferred_denom) {
                *deferred_denom = denom;
                hipLaunchKernelGGL((k_colfilter_lds<2, false, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            } else {
                hipLaunchKernelGGL((k_colfilter_lds<2, true, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            }
            LAUNCHCHK();
            return TRI_OK;
        }
        if (transposed_out)
            hipLaunchKernelGGL((k_colfilter_lds<1, true, true>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
        else if (srcmode == 0)
            hipLaunchKernelGGL((k_colfilter_lds<0, true, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
        else if (deferred_denom) {
            *deferred_denom = denom;
            hipLaunchKernelGGL((k_colfilter_lds<1, false, false>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
        } else
            hipLaunchKernelGGL((k_colfilter_lds<1, true, false>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
        LAUNCHCHK();
        return TRI_OK;
    }
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    dim3 grid((unsigned)cdiv(C, blk), (unsigned)W, 2);
    if (srcmode == 0)
        hipLaunchKernelGGL(k_colfilter<0>, grid, dim3(blk), 0, r.st, bufW, bufO, srcData, srcFlags,
                           dstW, dstO, n, C, rad, denom, bws, sws, dws);
    else
        hipLaunchKernelGGL(k_colfilter<1>, grid, dim3(blk), 0, r.st, bufW, bufO, srcData, srcFlags,
                           dstW, dstO, n, C, rad, denom, bws, sws, dws);
    LAUNCHCHK();
    return TRI_OK;
}

int launch_colst(const Run& r, const StWin& sw, const float* data, const double* med,
                 uint8_t* out, const int64_t* d_chunk_ends, int L, int C, int G, size_t ws_data,
                 size_t ws_out, int64_t W) {
    double thr_scale = r.p->outlier_nsigma * TRI_MAD_NORMAL;  // flagging.py:623
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    dim3 grid((unsigned)cdiv(C, blk), (unsigned)G, (unsigned)W);
    if (st_use_fused(sw)) {
        StFusedArgs fa;
        for (int j = 0; j < 4; j++) fa.tf[j] = sw.tf[j];
        hipLaunchKernelGGL((k_colst_fused<1, 2, 4, 8>), grid, dim3(blk), 0, r.st, data, med, out,
                           d_chunk_ends, fa, thr_scale, L, C, G, ws_data, ws_out);
    } else {
        hipLaunchKernelGGL(k_colst_dyn, grid, dim3(blk), 0, r.st, data, med, out, r.ws.ring, r.ws.acc,
                           d_chunk_ends, sw, thr_scale, L, C, G, ws_data, ws_out);
    }
    LAUNCHCHK();
    return TRI_OK;
}

template <typename T>
int launch_transpose(const Run& r, const T* src, T* dst, int R, int C, size_t sws, size_t dws,
                     int64_t W, float denom = 0.0f) {
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64), (unsigned)W);
    if (sizeof(T) == 1 && R % 4 == 0 && C % 4 == 0 && sws % 4 == 0 && dws % 4 == 0 &&
        ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0)) {
        hipLaunchKernelGGL(k_transpose_u8x4, grid, dim3(16, 16), 0, r.st, (const uint8_t*)src, (uint8_t*)dst, R, C, sws, dws);
        LAUNCHCHK();
        return TRI_OK;
    }
    hipLaunchKernelGGL(k_transpose<T>, grid, dim3(64, 4), 0, r.st, src, dst, R, C, sws, dws, denom);
    LAUNCHCHK();
    return TRI_OK;
}


// ---- elementwise launch helpers (vector path when the plan allows) ----------
template <int OP>
int launch_u8(const Run& r, const uint8_t* a, uint8_t* b, size_t nper, size_t ws_a, size_t ws_b, int64_t W) {
    if (r.pl.vec && nper % 16 == 0 && ws_a % 16 == 0 && ws_b % 16 == 0) {
        hipLaunchKernelGGL(k_u8_op16<OP>, grid1(nper / 16, W), dim3(256), 0, r.st, a, b, nper / 16, ws_a, ws_b);
    } else if (OP == 0) {
        for (int64_t w = 0; w < W; w++)
            hipLaunchKernelGGL(k_copy_u8, dim3((unsigned)cdiv(nper, 256)), dim3(256), 0, r.st, a + w * ws_a, b + w * ws_b, nper);
    } else if (OP == 1) {
        hipLaunchKernelGGL(k_or, grid1(nper, W), dim3(256), 0, r.st, b, a, nper, ws_b, ws_a);
    } else {
        for (int64_t w = 0; w < W; w++)
            hipLaunchKernelGGL(k_normalise_flags, dim3((unsigned)cdiv(nper, 256)), dim3(256), 0, r.st, a + w * ws_a, b + w * ws_b, nper);
    }
    LAUNCHCHK();
    return TRI_OK;
}

template <int MODE>
int launch_masked_div(const Run& r, const float* w, float* o, const float* data, size_t nper, size_t ws_wo, size_t ws_data, int64_t W, float denom = 0.0f,
                      uint8_t* nanflag = nullptr, int C = 1) {
    if (nper % 4 == 0 && ws_wo % 4 == 0 && ws_data % 4 == 0 && (nanflag == nullptr || C % 4 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)o % 16 == 0))
        hipLaunchKernelGGL(k_masked_div4<MODE>, grid1(nper / 4, W), dim3(256), 0, r.st, w, o, data, nper / 4, ws_wo, ws_data, denom, nanflag, C);
    else
        hipLaunchKernelGGL(k_masked_div<MODE>, grid1(nper, W), dim3(256), 0, r.st, w, o, data, nper, ws_wo, ws_data, denom, nanflag, C);
    LAUNCHCHK();
    return TRI_OK;
}

int launch_sub(const Run& r, const float* a, const float* b, float* out, size_t nper, size_t ws_a, size_t ws_b, size_t ws_o, int64_t W) {
    if (nper % 4 == 0 && ws_a % 4 == 0 && ws_b % 4 == 0 && ws_o % 4 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)out % 16 == 0))
        hipLaunchKernelGGL(k_sub4, grid1(nper / 4, W), dim3(256), 0, r.st, a, b, out, nper / 4, ws_a, ws_b, ws_o);
    else
        hipLaunchKernelGGL(k_sub, grid1(nper, W), dim3(256), 0, r.st, a, b, out, nper, ws_a, ws_b, ws_o);
    LAUNCHCHK();
    return TRI_OK;
}

// Frequency-axis stage reading the time-axis stage's TF images directly
// (k_colfilter_lds_t).  Usable for radii whose four rings fit LDS at 128 threads.
bool colfilter_t_usable(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_TIN"); return e && e[0] == '1'; }();
    return !off && rad > 0 && rad <= 16;
}

int launch_colfilter_t(const Run& r, const float* srcW, const float* srcO, float* dstW, float* dstO,
                       int n, int C, int ld, int rad, size_t sws_img, size_t dws, int64_t W, float* deferred_denom) {
    float denom = box_denominator(rad);
    size_t lds = ((size_t)4 * 2 * rad * CFT_BT + (size_t)CFT_PF * (CFT_BT + 1)) * sizeof(float);
    static const hipError_t attr = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_colfilter_lds_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, CFT_BT), (unsigned)W, 2);
    if (deferred_denom) {
        *deferred_denom = denom;
        hipLaunchKernelGGL(k_colfilter_lds_t<false>, grid, dim3(CFT_BT), lds, r.st, srcW, srcO, dstW, dstO, n, C, ld, rad, denom, sws_img, dws);
    } else {
        hipLaunchKernelGGL(k_colfilter_lds_t<true>, grid, dim3(CFT_BT), lds, r.st, srcW, srcO, dstW, dstO, n, C, ld, rad, denom, sws_img, dws);
    }
    LAUNCHCHK();
    return TRI_OK;
}

// _get_background2d (flagging.py:516-579) for the median spectra of the batch,
// held in spectrum layout [Fa][Wb] (line axis = channel, column = window).
// Result: rows [0,Fa) of ws.so hold the background.
int spectrum_background(const Run& r) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    int Fa = (int)pl.Fa, Wn = (int)r.Wb, G = (int)pl.G;
    size_t nS = (size_t)Fa * Wn;
    hipLaunchKernelGGL(k_copy_u8, dim3((unsigned)cdiv(nS, 256)), dim3(256), 0, r.st, ws.sflags, ws.sbgf, nS);
    LAUNCHCHK();
    double rej = TRI_MAD_NORMAL * r.p->background_reject;  // flagging.py:568
    for (int ext = pl.nit; ext >= 0; ext--) {
        bool final_pass = ext == 0;
        double sigma = (double)(final_pass ? 1 : ext) * r.p->spike_width_freq;  // flagging.py:554, 576
        int rad = (int)box_radius(sigma);
        if (rad > 0) {
            int rc = launch_colfilter(r, 0, ws.sw, ws.so, ws.sdata, ws.sbgf, ws.sw, ws.so, Fa, Wn, rad, 0, 0, 0, 1);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_build_wo, grid1(nS, 1), dim3(256), 0, r.st, ws.sdata, ws.sbgf, ws.sw, ws.so, nS, (size_t)0, (size_t)0);
            LAUNCHCHK();
        }
        if (final_pass) {
            { int rc2 = launch_masked_div<0>(r, ws.sw, ws.so, ws.sdata, nS, 0, 0, 1); if (rc2) return rc2; }
        } else {
            { int rc2 = launch_masked_div<1>(r, ws.sw, ws.so, ws.sdata, nS, 0, 0, 1); if (rc2) return rc2; }
            // per (window, chunk) median of the residual: element (f, w) at f*Wn + w
            // -> row = w (RS 1), element stride Wn
            int rc = launch_median(r, ws.so, ws.sbgf, ws.smed, 0, 0, 1, (size_t)Wn, ws.segC_start, ws.segC_len, Wn, G, 1, pl.maxchunk);
            if (rc) return rc;
            hipLaunchKernelGGL(k_reject<false>, grid1(nS, 1), dim3(256), 0, r.st, ws.so, ws.sbgf, ws.smed, ws.d_chunk_of, rej, Fa, Wn, G, (size_t)0, (size_t)0);
            LAUNCHCHK();
        }
    }
    hipLaunchKernelGGL(k_colinterp, dim3((unsigned)cdiv(Wn, 256), 1), dim3(256), 0, r.st, ws.so, Fa, Wn, (size_t)0, (const uint8_t*)nullptr);
    LAUNCHCHK();
    return TRI_OK;
}

// _get_background2d (flagging.py:516-579) for every window of the batch.
// In: dataTF / dataFT, flagsTF (spectral flags already OR-ed in).  Out: the
// background in FT layout in rows [0,Fa) of ws.Bo (window stride PF*T).
int background2d(const Run& r) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    int T = (int)pl.T, Fa = (int)pl.Fa, G = (int)pl.G;
    int64_t W = r.Wb;
    size_t N = (size_t)T * Fa;
    size_t wsA = (size_t)pl.PT * Fa, wsB = (size_t)pl.PF * T;
    // Background flags live in FT layout (ws.bgfFT).  For the time-axis stage
    // they are needed per (time, channel) column thread: either as a TF byte
    // image (general case) or, when T % 4 == 0, packed four times per word
    // ("TF4": [T/4][Fa] uint32) -- which is exactly the 32-bit transpose of the
    // FT byte image viewed as [Fa][T/4] words.
    static const bool no_pack = [] { const char* e = getenv("TRI_NO_PACKED_FLAGS"); return e && e[0] == '1'; }();
    const bool packed = !no_pack && (T % 4 == 0) && (N % 4 == 0);
    int rc = launch_transpose<uint8_t>(r, ws.flagsTF, ws.bgfFT, T, Fa, N, N, W);
    if (rc) return rc;
    if (packed) {
        rc = launch_transpose<float>(r, reinterpret_cast<const float*>(ws.bgfFT), reinterpret_cast<float*>(ws.bgfTF), Fa, T / 4, N / 4, N / 4, W);
        if (rc) return rc;
    } else {
        rc = launch_u8<0>(r, ws.flagsTF, ws.bgfTF, N, N, N, W);
        if (rc) return rc;
    }
    double rej = TRI_MAD_NORMAL * r.p->background_reject;
    for (int ext = pl.nit; ext >= 0; ext--) {
        bool final_pass = ext == 0;
        double e = (double)(final_pass ? 1 : ext);
        int r0 = (int)box_radius(e * r.p->spike_width_time);
        int r1 = (int)box_radius(e * r.p->spike_width_freq);
        // --- time axis (TF layout: line = time, column = channel) ---
        // Building the weight / data images first (13 B/sample, vectorised) and
        // streaming float images through the sequential kernel is faster than
        // byte loads of the flags inside its per-line loop.
        static const bool prebuild = [] { const char* e = getenv("TRI_TIME_PREBUILD"); return !(e && e[0] == '0'); }();
        float den_t = 0.0f, den_f = 0.0f;   // divisions deferred to the transposes / masked_div
        bool direct_ft = false;
        // frequency stage able to read the time stage's TF images itself (no transposes)
        const bool tin = colfilter_t_usable(r1);
        float* den_t_ptr = tin ? nullptr : &den_t;
        // (for the in-place multi-pass kernel, used at large radii, building on
        //  the fly measured faster: 11.4 vs 16.8 ms per call at 128 windows)
        if (r0 > 0 && packed && colfilter_lds_block(r0, Fa) > 0) {
            // single sweep straight from (data, packed flags): no image build
            rc = launch_colfilter(r, 2, ws.Aw, ws.Ao, ws.dataTF, ws.bgfTF, ws.Aw, ws.Ao, T, Fa, r0, wsA, N, wsA, W, den_t_ptr);
            if (rc) return rc;
        } else if (r0 > 0 && !packed && prebuild && colfilter_lds_block(r0, Fa) > 0) {
            const bool lds_path = true;
            const size_t boff = lds_path ? 0 : (size_t)4 * r0 * Fa;
            if (N % 4 == 0 && wsA % 4 == 0)
                hipLaunchKernelGGL(k_build_wo4, grid1(N / 4, W), dim3(256), 0, r.st, ws.dataTF, ws.bgfTF, ws.Aw + boff, ws.Ao + boff, N / 4, N, wsA);
            else
                hipLaunchKernelGGL(k_build_wo, grid1(N, W), dim3(256), 0, r.st, ws.dataTF, ws.bgfTF, ws.Aw + boff, ws.Ao + boff, N, N, wsA);
            LAUNCHCHK();
            // (TRI_FILTER_DIRECT_FT=1: write the filtered images straight into FT
            //  layout; measured no faster than the two transposes it replaces)
            static const bool want_direct = [] { const char* e = getenv("TRI_FILTER_DIRECT_FT"); return e && e[0] == '1'; }();
            direct_ft = lds_path && want_direct && (T % 4 == 0) && (wsB % 4 == 0);
            if (direct_ft) {
                size_t off2 = colfilter_lds_block(r1, T) > 0 ? 0 : (size_t)4 * r1 * T;
                rc = launch_colfilter(r, 1, ws.Aw, ws.Ao, nullptr, nullptr, ws.Bw + off2, ws.Bo + off2, T, Fa, r0, wsA, 0, wsB, W, nullptr, true);
            } else {
                rc = launch_colfilter(r, 1, ws.Aw, ws.Ao, nullptr, nullptr, ws.Aw, ws.Ao, T, Fa, r0, wsA, 0, wsA, W, den_t_ptr);
            }
            if (rc) return rc;
        } else if (r0 > 0) {
            const uint8_t* fl = ws.bgfTF;
            if (packed) {   // byte image needed: rebuild it next to the packed one
                rc = launch_transpose<uint8_t>(r, ws.bgfFT, ws.comb, Fa, T, N, N, W);
                if (rc) return rc;
                fl = ws.comb;
            }
            rc = launch_colfilter(r, 0, ws.Aw, ws.Ao, ws.dataTF, fl, ws.Aw, ws.Ao, T, Fa, r0, wsA, N, wsA, W);
            if (rc) return rc;
        } else {
            const uint8_t* fl = ws.bgfTF;
            if (packed) {
                rc = launch_transpose<uint8_t>(r, ws.bgfFT, ws.comb, Fa, T, N, N, W);
                if (rc) return rc;
                fl = ws.comb;
            }
            hipLaunchKernelGGL(k_build_wo, grid1(N, W), dim3(256), 0, r.st, ws.dataTF, fl, ws.Aw, ws.Ao, N, N, wsA);
            LAUNCHCHK();
        }
        // --- to FT layout: rows [4 r1, 4 r1 + Fa) of the padded buffers for the
        //     in-place multi-pass filter, rows [0, Fa) for the single-sweep one ---
        size_t off = colfilter_lds_block(r1, T) > 0 ? 0 : (size_t)4 * r1 * T;
        if (tin && !direct_ft) {
            rc = launch_colfilter_t(r, ws.Aw, ws.Ao, ws.Bw, ws.Bo, Fa, T, Fa, r1, wsA, wsB, W, &den_f);
            if (rc) return rc;
        } else if (!direct_ft) {
            rc = launch_transpose<float>(r, ws.Aw, ws.Bw + off, T, Fa, wsA, wsB, W, den_t);
            if (rc) return rc;
            rc = launch_transpose<float>(r, ws.Ao, ws.Bo + off, T, Fa, wsA, wsB, W, den_t);
            if (rc) return rc;
        }
        // --- frequency axis (FT layout: line = channel, column = time) ---
        if (r1 > 0 && !(tin && !direct_ft)) {
            rc = launch_colfilter(r, 1, ws.Bw, ws.Bo, nullptr, nullptr, ws.Bw, ws.Bo, Fa, T, r1, wsB, 0, wsB, W, &den_f);
            if (rc) return rc;
        }
        if (final_pass) {
            // ws.rowcnt (W * T ints, idle until the end of the iteration) doubles as the
            // per-line "background holds a NaN" marker
            HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T, r.st));
            rc = launch_masked_div<0>(r, ws.Bw, ws.Bo, ws.dataFT, N, wsB, N, W, den_f, reinterpret_cast<uint8_t*>(ws.rowcnt), T);
            if (rc) return rc;
        } else {
            rc = launch_masked_div<1>(r, ws.Bw, ws.Bo, ws.dataFT, N, wsB, N, W, den_f);
            if (rc) return rc;
            // block medians over (all times) x (chunk channels): contiguous in FT
            rc = launch_median(r, ws.Bo, ws.bgfFT, ws.med, wsB, N, 0, 1, ws.segB_start, ws.segB_len, 1, G, W, pl.maxchunk * pl.T,
                               T % 4 == 0 && wsB % 4 == 0 && N % 4 == 0);
            if (rc) return rc;
            if (r.pl.vec && wsB % 4 == 0)
                hipLaunchKernelGGL(k_reject4, grid1(N / 4, W), dim3(256), 0, r.st, ws.Bo, ws.bgfFT, ws.med, ws.d_chunk_of, rej, T / 4, G, N / 4, wsB, N);
            else
                hipLaunchKernelGGL(k_reject<true>, grid1(N, W), dim3(256), 0, r.st, ws.Bo, ws.bgfFT, ws.med, ws.d_chunk_of, rej, Fa, T, G, wsB, N);
            LAUNCHCHK();
            if (packed)
                rc = launch_transpose<float>(r, reinterpret_cast<const float*>(ws.bgfFT), reinterpret_cast<float*>(ws.bgfTF), Fa, T / 4, N / 4, N / 4, W);
            else
                rc = launch_transpose<uint8_t>(r, ws.bgfFT, ws.bgfTF, Fa, T, N, N, W);
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(k_colinterp, dim3((unsigned)cdiv(T, 256), (unsigned)W), dim3(256), 0, r.st, ws.Bo, Fa, T, wsB, reinterpret_cast<const uint8_t*>(ws.rowcnt));
    LAUNCHCHK();
    return TRI_OK;
}

// One major iteration (_get_flags_impl, flagging.py:745-781) for a batch.
template <int VD>
int run_iteration(Run& r, const void* vis, uint8_t* iter_flags, uint8_t* out_flags, bool update_iter, bool tap) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    const tri_params* p = r.p;
    int T = (int)pl.T, F = (int)pl.F, Fa = (int)pl.Fa, G = (int)pl.G;
    int64_t W = r.Wb;
    int Wn = (int)W;
    size_t N = (size_t)T * Fa, NF = (size_t)T * F;
    size_t wsB = (size_t)pl.PF * T;
    int rc;

    // flagging.py:756  _average_freq
    if (pl.vec)
        hipLaunchKernelGGL(k_prepare4<VD>, dim3((unsigned)cdiv(N * (size_t)W / 4, 256)), dim3(256), 0, r.st, vis, iter_flags, ws.dataTF, ws.flagsTF, N * (size_t)W / 4);
    else
        hipLaunchKernelGGL(k_prepare<VD>, grid1(N, W), dim3(256), 0, r.st, vis, iter_flags, ws.dataTF, ws.flagsTF, T, F, Fa, (int)pl.avg);
    LAUNCHCHK();
    rc = launch_transpose<float>(r, ws.dataTF, ws.dataFT, T, Fa, N, N, W);
    if (rc) return rc;
    rc = launch_transpose<uint8_t>(r, ws.flagsTF, ws.flagsFT, T, Fa, N, N, W);
    if (rc) return rc;

    // flagging.py:944  _time_median: rows of the FT layout are contiguous in time
    rc = launch_median(r, ws.dataFT, ws.flagsFT, ws.med, N, N, (size_t)T, 1, ws.segT_start, ws.segT_len, Fa, 1, W, pl.T, false, T % 4 == 0, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_spec_from_med, dim3((unsigned)cdiv((size_t)Fa * Wn, 256)), dim3(256), 0, r.st, ws.med, ws.sdata, ws.sflags, Fa, Wn);
    LAUNCHCHK();

    // flagging.py:945-952  spectrum background, subtraction, SumThreshold
    rc = spectrum_background(r);
    if (rc) return rc;
    size_t nS = (size_t)Fa * Wn;
    rc = launch_sub(r, ws.sdata, ws.so, ws.sres, nS, 0, 0, 0, 1);
    if (rc) return rc;
    rc = launch_median(r, ws.sres, ws.sflags, ws.smed, 0, 0, 1, (size_t)Wn, ws.segC_start, ws.segC_len, Wn, G, 1, pl.maxchunk);
    if (rc) return rc;
    rc = launch_colst(r, pl.swF, ws.sres, ws.smed, ws.sout, ws.d_chunk_ends, Fa, Wn, G, 0, 0, 1);
    if (rc) return rc;

    // flagging.py:954  flags |= spec_flags
    if (pl.vec) {
        hipLaunchKernelGGL(k_spec_rows, dim3((unsigned)cdiv(nS, 256)), dim3(256), 0, r.st, ws.sout, ws.srows, Fa, Wn);
        hipLaunchKernelGGL(k_or_spec16, grid1(N / 16, W), dim3(256), 0, r.st, ws.flagsTF, ws.srows, T, Fa / 16);
    } else {
        hipLaunchKernelGGL(k_or_spec, grid1(N, W), dim3(256), 0, r.st, ws.flagsTF, ws.sout, T, Fa, Wn);
    }
    LAUNCHCHK();

    // flagging.py:957-962  2-D background (FT layout, ws.Bo), then the residual
    rc = background2d(r);
    if (rc) return rc;
    if (tap && r.dbg) {
        HIPCHK(hipMemcpyAsync(r.dbg->f32 + Fa, ws.Bo, N * sizeof(float), hipMemcpyDeviceToDevice, r.st));
    }
    rc = launch_sub(r, ws.dataFT, ws.Bo, ws.Bo, N, N, wsB, wsB, W);
    if (rc) return rc;
    float* residFT = ws.Bo;   // window stride wsB
    float* residTF = ws.Aw;   // window stride N (the time-axis scratch is free again)
    rc = launch_transpose<float>(r, residFT, residTF, Fa, T, wsB, N, W);
    if (rc) return rc;

    // flagging.py:964  SumThreshold along time.  MAD per channel over time =
    // contiguous rows of the FT layout; flags = input | spectral flags.
    rc = launch_transpose<uint8_t>(r, ws.flagsTF, ws.flagsFT, T, Fa, N, N, W);
    if (rc) return rc;
    rc = launch_median(r, residFT, ws.flagsFT, ws.med, wsB, N, (size_t)T, 1, ws.segT_start, ws.segT_len, Fa, 1, W, pl.T, false, T % 4 == 0, true);
    if (rc) return rc;
    rc = launch_colst(r, pl.swT, residTF, ws.med, ws.tflTF, ws.d_tends, T, Fa, 1, N, N, W);
    if (rc) return rc;

    // flagging.py:967-969  flags |= time_flags; SumThreshold along frequency.
    // MAD per (time, chunk) = contiguous row segments of the TF layout.
    rc = launch_u8<1>(r, ws.tflTF, ws.flagsTF, N, N, N, W);
    if (rc) return rc;
    rc = launch_median(r, residTF, ws.flagsTF, ws.med, N, N, (size_t)Fa, 1, ws.segC_start, ws.segC_len, T, G, W, pl.maxchunk, false, Fa % 4 == 0);
    if (rc) return rc;
    rc = launch_colst(r, pl.swF, residFT, ws.med, ws.fflFT, ws.d_chunk_ends, Fa, T, G, wsB, N, W);
    if (rc) return rc;
    rc = launch_transpose<uint8_t>(r, ws.fflFT, ws.fflTF, Fa, T, N, N, W);
    if (rc) return rc;

    if (tap && r.dbg) {
        hipLaunchKernelGGL(k_gather_col_f32, dim3((unsigned)cdiv(Fa, 256)), dim3(256), 0, r.st, ws.sres, r.dbg->f32, Fa, Wn, 0);
        hipLaunchKernelGGL(k_gather_col_u8, dim3((unsigned)cdiv(Fa, 256)), dim3(256), 0, r.st, ws.sout, r.dbg->u8, Fa, Wn, 0);
        LAUNCHCHK();
        HIPCHK(hipMemcpyAsync(r.dbg->f32 + Fa + N, residTF, N * sizeof(float), hipMemcpyDeviceToDevice, r.st));
        HIPCHK(hipMemcpyAsync(r.dbg->u8 + Fa, ws.tflTF, N, hipMemcpyDeviceToDevice, r.st));
        HIPCHK(hipMemcpyAsync(r.dbg->u8 + Fa + N, ws.fflTF, N, hipMemcpyDeviceToDevice, r.st));
    }

    // flagging.py:973  _combine_flags (time smearing)
    {
        int64_t e = p->time_extend;
        int64_t half = e >= 0 ? e / 2 : -((-e + 1) / 2);   // Python floor division
        int lo = (int)-half, hi = (int)(-half + e);
        if (pl.vec)
            hipLaunchKernelGGL(k_combine16, grid1(N / 16, W), dim3(256), 0, r.st, ws.srows, ws.tflTF, ws.fflTF, ws.comb, T, Fa / 16, lo, hi);
        else
            hipLaunchKernelGGL(k_combine, grid1(N, W), dim3(256), 0, r.st, ws.sout, ws.tflTF, ws.fflTF, ws.comb, T, Fa, Wn, lo, hi);
        LAUNCHCHK();
    }
    // flagging.py:975  _unaverage_freq (replication, frequency smearing, counts)
    {
        int64_t e = p->freq_extend;
        int64_t half = e >= 0 ? e / 2 : -((-e + 1) / 2);
        int lo = (int)-half, hi = (int)(-half + e);
        HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T * sizeof(int), r.st));
        if (pl.vec && lo == -1 && hi == 2) {
            dim3 grid((unsigned)cdiv(F / 16, 64), (unsigned)T, (unsigned)W);
            hipLaunchKernelGGL((k_unaverage16<-1, 3>), grid, dim3(64), 0, r.st, ws.comb, ws.dil, ws.rowcnt, T, F / 16);
            hipLaunchKernelGGL(k_colcount, grid1(F / 4, W), dim3(256), 0, r.st, ws.dil, ws.colcnt, T, F / 4);
        } else {
            HIPCHK(hipMemsetAsync(ws.colcnt, 0, (size_t)W * F * sizeof(int), r.st));
            dim3 grid((unsigned)cdiv(F, 256), (unsigned)T, (unsigned)W);
            hipLaunchKernelGGL(k_unaverage, grid, dim3(256), 0, r.st, ws.comb, ws.dil, ws.rowcnt, ws.colcnt, T, Fa, F, (int)pl.avg, lo, hi);
        }
        LAUNCHCHK();
    }
    // flagging.py:910-918 whole-row / whole-column rules; :777-781 NaN OR; :1193
    double row_limit = p->flag_all_freq_frac * (double)F;
    double col_limit = (double)T * p->flag_all_time_frac;
    if (pl.vec)
        hipLaunchKernelGGL(k_final16<VD>, grid1(NF / 16, W), dim3(256), 0, r.st, ws.dil, ws.rowcnt, ws.colcnt, vis, out_flags, iter_flags, T, F / 16, row_limit, col_limit, update_iter ? 1 : 0);
    else
        hipLaunchKernelGGL(k_final<VD>, grid1(NF, W), dim3(256), 0, r.st, ws.dil, ws.rowcnt, ws.colcnt, vis, out_flags, iter_flags, T, F, row_limit, col_limit, update_iter ? 1 : 0);
    LAUNCHCHK();
    return TRI_OK;
}

int flagger_impl(const void* vis, int vis_dtype, const uint8_t* flags, uint8_t* out_flags,
                 int64_t n_cp, int64_t T, int64_t F, const tri_params* p, void* workspace,
                 size_t workspace_bytes, void* stream, Debug* dbg) {
    if (!vis || !flags || !out_flags || !p) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_cp < 0) return set_err(TRI_EINVAL, "negative window count");
    if (vis_dtype != TRI_VIS_C64 && vis_dtype != TRI_VIS_F32) return set_err(TRI_EUNSUPPORTED, "vis dtype must be complex64 or float32");
    Run r;
    r.st = (hipStream_t)stream;
    r.p = p;
    r.dbg = dbg;
    int rc = make_plan(T, F, p, &r.pl);
    if (rc) return rc;
    if (r.pl.G > TRI_MAX_CHUNKS) return set_err(TRI_EUNSUPPORTED, "at most %d frequency chunks", TRI_MAX_CHUNKS);
    // the 16-byte kernels need 16-byte aligned user buffers (torch allocations are)
    if ((((uintptr_t)vis) | ((uintptr_t)flags) | ((uintptr_t)out_flags)) & 15) r.pl.vec = false;
    if (n_cp == 0) return TRI_OK;
    if (!workspace) return set_err(TRI_EWORKSPACE, "NULL workspace");
    if (((uintptr_t)workspace & 255) != 0) return set_err(TRI_EINVAL, "workspace must be 256-byte aligned");
    // largest batch the workspace admits (the carve-up is monotone in Wb)
    Ws probe;
    carve(r.pl, 1, nullptr, 0, true, &probe);
    if (probe.total > workspace_bytes)
        return set_err(TRI_EWORKSPACE, "workspace of %zu bytes is smaller than the %zu needed for one window", workspace_bytes, probe.total);
    int64_t lo = 1, hi = std::min<int64_t>(n_cp, 16384);
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) / 2;
        carve(r.pl, mid, nullptr, 0, true, &probe);
        if (probe.total <= workspace_bytes) lo = mid; else hi = mid - 1;
    }
    int64_t Wb = lo;
    carve(r.pl, Wb, workspace, workspace_bytes, false, &r.ws);

    // device tables
    ChunkTab ct;
    ct.n = (int)p->n_chunk_ends;
    for (int i = 0; i < ct.n; i++) ct.ends[i] = p->chunk_ends[i];
    {
        int nthr = (int)std::max<int64_t>(r.pl.Fa, ct.n);
        hipLaunchKernelGGL(k_tables, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, r.st, ct, (int)T, (int)r.pl.Fa,
                           r.ws.d_chunk_ends, r.ws.d_tends, r.ws.segC_start, r.ws.segC_len, r.ws.segB_start,
                           r.ws.segB_len, r.ws.segT_start, r.ws.segT_len, r.ws.d_chunk_of);
        LAUNCHCHK();
    }
    size_t NF = (size_t)T * F;
    size_t esz = vis_dtype == TRI_VIS_C64 ? 8 : 4;
    if (p->num_major_iterations == 0)
        HIPCHK(hipMemsetAsync(out_flags, 0, (size_t)n_cp * NF, r.st));
    for (int64_t w0 = 0; w0 < n_cp; w0 += Wb) {
        r.Wb = std::min(Wb, n_cp - w0);
        const char* vis_b = (const char*)vis + (size_t)w0 * NF * esz;
        uint8_t* out_b = out_flags + (size_t)w0 * NF;
        size_t nb = (size_t)r.Wb * NF;
        // flagging.py:1182  iter_flags = flags.copy()  (non-zero = flagged)
        rc = launch_u8<2>(r, flags + (size_t)w0 * NF, r.ws.iter, NF, NF, NF, r.Wb);
        if (rc) return rc;
        for (int64_t it = 0; it < p->num_major_iterations; it++) {
            bool last = it == p->num_major_iterations - 1;
            bool tap = last && w0 == 0;
            if (vis_dtype == TRI_VIS_C64) rc = run_iteration<TRI_VIS_C64>(r, vis_b, r.ws.iter, out_b, !last, tap);
            else rc = run_iteration<TRI_VIS_F32>(r, vis_b, r.ws.iter, out_b, !last, tap);
            if (rc) return rc;
        }
    }
    return TRI_OK;
}

}  // namespace

extern "C" int tri_sum_threshold_flagger(const void* vis, int vis_dtype, const uint8_t* flags,
                                         uint8_t* out_flags, int64_t n_cp, int64_t ntime,
                                         int64_t nchan, const tri_params* p, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    return flagger_impl(vis, vis_dtype, flags, out_flags, n_cp, ntime, nchan, p, workspace,
                        workspace_bytes, stream, nullptr);
}

// Test hook: as above, additionally taps the last major iteration's
// intermediates of window 0 into caller-provided device buffers:
//   dbg_f32: [spec_resid (Fa)][background, FT layout (Fa*T)][residual, TF layout (T*Fa)]
//   dbg_u8 : [spec_flags (Fa)][time_flags TF (T*Fa)][freq_flags TF (T*Fa)]
extern "C" int tri_sum_threshold_flagger_debug(const void* vis, int vis_dtype, const uint8_t* flags,
                                               uint8_t* out_flags, int64_t n_cp, int64_t ntime,
                                               int64_t nchan, const tri_params* p, void* workspace,
                                               size_t workspace_bytes, void* stream,
                                               float* dbg_f32, uint8_t* dbg_u8) {
    Debug d{dbg_f32, dbg_u8};
    return flagger_impl(vis, vis_dtype, flags, out_flags, n_cp, ntime, nchan, p, workspace,
                        workspace_bytes, stream, (dbg_f32 && dbg_u8) ? &d : nullptr);
}

extern "C" int tri_abs_c64(const void* z, float* out, int64_t n, void* stream) {
    if (!z || !out || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (n == 0) return TRI_OK;
    hipLaunchKernelGGL(k_abs_c64, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)z, out, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_fill_windows(void* vis_windows_c64, uint8_t* flag_windows, int64_t n, void* stream) {
    if (!vis_windows_c64 || !flag_windows || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (n == 0) return TRI_OK;
    hipLaunchKernelGGL(k_fill_windows, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (float2*)vis_windows_c64, flag_windows, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_pack_data(const void* data_c64, const uint8_t* flag, const int32_t* row_bl,
                             const int32_t* row_time, int64_t rows, int64_t nchan, int64_t ncorr,
                             int64_t nbl, int64_t ntime, void* vis_windows_c64,
                             uint8_t* flag_windows, void* stream) {
    if (!data_c64 || !flag || !row_bl || !row_time || !vis_windows_c64 || !flag_windows)
        return set_err(TRI_EINVAL, "NULL pointer argument");
    if (rows < 0 || nchan <= 0 || ncorr <= 0 || nbl < 0 || ntime < 0) return set_err(TRI_EINVAL, "bad shape");
    if (rows == 0) return TRI_OK;
    if (rows > 0x7FFFFFFF) return set_err(TRI_EUNSUPPORTED, "too many rows in one call");
    dim3 grid((unsigned)cdiv(nchan, 256), 1, 1);
    // gridDim.y is limited to 65535: walk the rows in slabs
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        int64_t nr = std::min<int64_t>(65535, rows - r0);
        grid.y = (unsigned)nr;
        hipLaunchKernelGGL(k_pack, grid, dim3(256), 0, (hipStream_t)stream,
                           (const float2*)data_c64 + (size_t)r0 * nchan * ncorr, flag + (size_t)r0 * nchan * ncorr,
                           row_bl + r0, row_time + r0, (int)nchan, (int)ncorr, (int)nbl, (int)ntime,
                           (float2*)vis_windows_c64, flag_windows);
        LAUNCHCHK();
    }
    return TRI_OK;
}

extern "C" int tri_unpack_data(const uint8_t* flag_windows, const int32_t* row_bl,
                               const int32_t* row_time, int64_t rows, int64_t nchan, int64_t ncorr,
                               int64_t nbl, int64_t ntime, uint8_t* out_flags, void* stream) {
    if (!flag_windows || !row_bl || !row_time || !out_flags) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (rows < 0 || nchan <= 0 || ncorr <= 0 || nbl < 0 || ntime < 0) return set_err(TRI_EINVAL, "bad shape");
    dim3 grid((unsigned)cdiv(nchan, 256), 1, 1);
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        int64_t nr = std::min<int64_t>(65535, rows - r0);
        grid.y = (unsigned)nr;
        hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, (hipStream_t)stream, flag_windows, row_bl + r0, row_time + r0,
                           (int)nchan, (int)ncorr, (int)nbl, (int)ntime, out_flags + (size_t)r0 * nchan * ncorr);
        LAUNCHCHK();
    }
    return TRI_OK;
}

extern "C" int tri_bench_sumthreshold(const float* data, const double* mad, uint8_t* out,
                                      int64_t n_win, int64_t n_line, int64_t n_col,
                                      const int64_t* windows, int64_t n_windows,
                                      double outlier_nsigma, double rho, int variant, int repeats,
                                      float* ms_per_launch, void* stream) {
    if (!data || !mad || !out || !windows || !ms_per_launch) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_win <= 0 || n_line <= 0 || n_col <= 0 || repeats <= 0 || n_win > 65535) return set_err(TRI_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    StWin sw;
    int rc = make_stwin(windows, n_windows, rho, &sw);
    if (rc) return rc;
    double thr_scale = outlier_nsigma * TRI_MAD_NORMAL;
    size_t nthreads = (size_t)n_win * n_col;
    double* ring = nullptr;
    uint8_t* acc = nullptr;
    int64_t* d_ends = nullptr;
    HIPCHK(hipMalloc(&ring, nthreads * sw.ringtot * sizeof(double)));
    HIPCHK(hipMalloc(&acc, nthreads * sw.acccap));
    HIPCHK(hipMalloc(&d_ends, 2 * sizeof(int64_t)));
    int64_t ends[2] = {0, n_line};
    HIPCHK(hipMemcpyAsync(d_ends, ends, sizeof(ends), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    int C = (int)n_col, L = (int)n_line;
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    dim3 grid((unsigned)cdiv(C, blk), 1, (unsigned)n_win);
    size_t ws = (size_t)n_line * n_col;
    bool can_fuse = sw.nw == 4 && sw.w[0] == 1 && sw.w[1] == 2 && sw.w[2] == 4 && sw.w[3] == 8;
    if (variant == 2 && !can_fuse) return set_err(TRI_EUNSUPPORTED, "register cascade needs windows (1,2,4,8)");
    bool fused = variant == 2 || (variant == 0 && can_fuse);
    StFusedArgs fa;
    for (int j = 0; j < 4; j++) fa.tf[j] = sw.tf[j < sw.nw ? j : 0];
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < repeats; i++) {
        if (fused)
            hipLaunchKernelGGL((k_colst_fused<1, 2, 4, 8>), grid, dim3(blk), 0, st, data, mad, out, d_ends, fa, thr_scale, L, C, 1, ws, ws);
        else
            hipLaunchKernelGGL(k_colst_dyn, grid, dim3(blk), 0, st, data, mad, out, ring, acc, d_ends, sw, thr_scale, L, C, 1, ws, ws);
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    LAUNCHCHK();
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / repeats;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(ring);
    (void)hipFree(acc);
    (void)hipFree(d_ends);
    return TRI_OK;
}

// Test hook: exact segmented medians of |x| over unflagged samples of a
// [n_win][rows][row_len] array; segments [ends[g], ends[g+1]) along each row.
// variant: 0 = automatic choice, 1 = wave kernel (segments <= 1024),
// 2 = workgroup kernel with scalar loads, 3 = workgroup kernel with vector
// loads (needs row_len, segment bounds % 4 == 0).  med: [n_win][rows][G].
extern "C" int tri_test_median(const float* data, const uint8_t* flags, double* med,
                               int64_t n_win, int64_t rows, int64_t row_len,
                               const int64_t* seg_ends, int64_t n_seg_ends, int variant,
                               void* stream) {
    if (!data || !flags || !med || !seg_ends || n_seg_ends < 2 || n_win <= 0 || rows <= 0)
        return set_err(TRI_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int G = (int)n_seg_ends - 1;
    std::vector<int64_t> start(G), len(G);
    int64_t maxlen = 0;
    bool al4 = row_len % 4 == 0;
    for (int g = 0; g < G; g++) {
        start[g] = seg_ends[g];
        len[g] = seg_ends[g + 1] - seg_ends[g];
        if (len[g] < 0 || seg_ends[g] < 0 || seg_ends[g + 1] > row_len) return set_err(TRI_EINVAL, "bad segment");
        maxlen = std::max(maxlen, len[g]);
        al4 = al4 && start[g] % 4 == 0 && len[g] % 4 == 0;
    }
    int64_t *d_start = nullptr, *d_len = nullptr;
    HIPCHK(hipMalloc(&d_start, G * sizeof(int64_t)));
    HIPCHK(hipMalloc(&d_len, G * sizeof(int64_t)));
    HIPCHK(hipMemcpy(d_start, start.data(), G * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_len, len.data(), G * sizeof(int64_t), hipMemcpyHostToDevice));
    size_t WS = (size_t)rows * row_len, RS = (size_t)row_len;
    int R = (int)rows;
    if (variant == 0) variant = maxlen <= 64 * MW_K ? 1 : (al4 ? 3 : 2);
    if ((variant == 1 || variant == 4) && maxlen > 64 * MW_K) return set_err(TRI_EINVAL, "wave kernel handles segments <= 1024");
    if (variant == 4 && (row_len % 4 != 0 || maxlen + 3 > 64 * MW_K)) return set_err(TRI_EINVAL, "masked vector variant needs row_len % 4 == 0 and segments <= 1021");
    if (variant == 3 && !al4) return set_err(TRI_EINVAL, "vector loads need 4-aligned segments");
    if (variant == 1 && maxlen <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, false>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, true>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if ((variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * MW_K) ||
             (variant == 1 && G == 1 && al4 && seg_ends[0] == 0 && seg_ends[1] == row_len))
        hipLaunchKernelGGL((k_median_wave<MW_K, true>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 1)
        hipLaunchKernelGGL((k_median_wave<MW_K, false>), dim3((unsigned)cdiv((int64_t)R * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 3)
        hipLaunchKernelGGL(k_median<true>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else
        hipLaunchKernelGGL(k_median<false>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d_start);
    (void)hipFree(d_len);
    return TRI_OK;
}

// ===========================================================================
// "Next" rows: the cheap strategy steps that surround
// sum_threshold in conf/default.yaml, so a whole strategy chain can stay
// device-resident.
// ===========================================================================
// flag_nans_and_zeros (flagging.py:29-62): out = vis == 0 | isnan(vis) | flags != 0
template <int VD>
__global__ void k_flag_nans_zeros(const void* __restrict__ vis, const uint8_t* __restrict__ flags,
                                  uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool f;
    if (VD == TRI_VIS_C64) {
        float2 z = reinterpret_cast<const float2*>(vis)[i];
        f = (z.x == 0.0f && z.y == 0.0f) || isnan(z.x) || isnan(z.y);
    } else {
        float x = reinterpret_cast<const float*>(vis)[i];
        f = x == 0.0f || isnan(x);
    }
    out[i] = (f || flags[i] != 0) ? 1 : 0;
}

// out = flags, then for every selected baseline: out |= chan_mask (mode 0) or
// out = chan_mask (mode 1), broadcast over corr and time.  Serves
// apply_static_mask (flagging.py:151-172, one call per mask) and flag_autos
// (flagging.py:90-93: all-ones mask on the auto-correlation baselines).
// In-place safe (out == flags).  grid (ceil(nchan/256), ncorr*ntime, nbl)
__global__ void k_apply_bl_chan_mask(const uint8_t* __restrict__ flags, uint8_t* __restrict__ out,
                                     const uint8_t* __restrict__ bl_sel,
                                     const uint8_t* __restrict__ chan_mask, int mode, int nchan,
                                     size_t rows_per_bl) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nchan) return;
    size_t bl = blockIdx.z;
    size_t a = (bl * rows_per_bl + blockIdx.y) * (size_t)nchan + f;
    uint8_t v = flags[a];
    if (bl_sel[bl]) v = mode == 0 ? (uint8_t)((v | chan_mask[f]) ? 1 : 0) : (uint8_t)(chan_mask[f] ? 1 : 0);
    out[a] = v;
}

extern "C" int tri_flag_nans_and_zeros(const void* vis, int vis_dtype, const uint8_t* flags,
                                       uint8_t* out_flags, int64_t n, void* stream) {
    if (!vis || !flags || !out_flags || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (vis_dtype != TRI_VIS_C64 && vis_dtype != TRI_VIS_F32) return set_err(TRI_EUNSUPPORTED, "vis dtype must be complex64 or float32");
    if (n == 0) return TRI_OK;
    dim3 grid((unsigned)cdiv(n, 256));
    if (vis_dtype == TRI_VIS_C64)
        hipLaunchKernelGGL(k_flag_nans_zeros<TRI_VIS_C64>, grid, dim3(256), 0, (hipStream_t)stream, vis, flags, out_flags, (size_t)n);
    else
        hipLaunchKernelGGL(k_flag_nans_zeros<TRI_VIS_F32>, grid, dim3(256), 0, (hipStream_t)stream, vis, flags, out_flags, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_apply_baseline_channel_mask(const uint8_t* flags, uint8_t* out_flags,
                                               const uint8_t* bl_sel, const uint8_t* chan_mask,
                                               int mode, int64_t nbl, int64_t ncorr, int64_t ntime,
                                               int64_t nchan, void* stream) {
    if (!flags || !out_flags || !bl_sel || !chan_mask) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (mode != 0 && mode != 1) return set_err(TRI_EINVAL, "mode must be 0 (or) or 1 (override)");
    if (nbl < 0 || ncorr < 0 || ntime < 0 || nchan < 0) return set_err(TRI_EINVAL, "bad shape");
    if (nbl == 0 || ncorr * ntime == 0 || nchan == 0) return TRI_OK;
    if (ncorr * ntime > 65535 || nbl > 65535) return set_err(TRI_EUNSUPPORTED, "corr*time and bl must each be <= 65535 per call");
    dim3 grid((unsigned)cdiv(nchan, 256), (unsigned)(ncorr * ntime), (unsigned)nbl);
    hipLaunchKernelGGL(k_apply_bl_chan_mask, grid, dim3(256), 0, (hipStream_t)stream, flags, out_flags, bl_sel,
                       chan_mask, mode, (int)nchan, (size_t)(ncorr * ntime));
    LAUNCHCHK();
    return TRI_OK;
}
