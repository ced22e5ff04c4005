// kernels_elementwise.hpp -- amplitude / flag preconditioning, transposes, masked division, rejection, NaN interpolation, flag combination, pack / unpack, strategy steps
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once

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
        if (VD == TRI_VIS_F64) {
            // float64 amplitude + float32 accumulator: the sum is formed in float64 and stored as float32 (flagging.py:858-859)
            const double a = fabs(reinterpret_cast<const double*>(vis)[base + f]);
            if (!iflags[base + f] && !isnan(a)) { sum = (float)((double)sum + a); cnt++; }
            continue;
        }
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
// PANEL: the output [C][R] is written as COLUMN PANELS [R / 64][C][64] -- the 64 columns a wave of a column kernel owns
// are contiguous, and so are the rows it visits one after the other: its walk becomes one linear stream (host: R % 64 == 0).
// Element (line c, column r) sits at ((r >> 6) * C + c) * 64 + (r & 63).
template <typename T, bool PANEL = false>
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
            if (PANEL) d[((size_t)blockIdx.y * C + c) * 64 + tx] = v;
            else d[(size_t)c * R + r] = v;
        }
    }
}

// rows [L][C] -> panel image [C / 64][L][64] (measurement hook only: the flagger's panels come out of k_transpose<T, true>)
template <typename T>
__global__ void k_panelize(const T* __restrict__ src, T* __restrict__ dst, int L, int C, size_t ws) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const int l = (int)(i / C), c = (int)(i % C);
    dst[blockIdx.y * ws + ((size_t)(c >> 6) * L + l) * 64 + (c & 63)] = src[blockIdx.y * ws + i];
}
template <typename T>
__global__ void k_unpanel_w(const T* __restrict__ src, T* __restrict__ dst, int L, int C, size_t ws) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const int l = (int)(i / C), c = (int)(i % C);
    dst[blockIdx.y * ws + i] = src[blockIdx.y * ws + ((size_t)(c >> 6) * L + l) * 64 + (c & 63)];
}
// panel image [C / 64][L][64] -> rows [L][C] (debug taps of the panel-layout SumThreshold images)
template <typename T>
__global__ void k_unpanel(const T* __restrict__ src, T* __restrict__ dst, int L, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const int l = (int)(i / C), c = (int)(i % C);
    dst[i] = src[((size_t)(c >> 6) * L + l) * 64 + (c & 63)];
}

// uint8 transpose with 4-byte accesses on both sides (R % 4 == 0, C % 4 == 0):
// a 128 x 64 byte tile goes through LDS as words; each thread then takes 4x4
// byte blocks (four words of four consecutive source rows), transposes them in
// registers with byte permutes and writes four words -- 128-byte runs along
// every output row.
//   COPY: the source words are also written, untransposed, to `copy`
//   ZERO: data[c][r] (float, the output's layout) is zeroed wherever the
//         transposed byte is non-zero (k_zero_flagged4's job, flags in hand)
// Together: the start of a major iteration (flagsTF = running flags, flagsFT
// = their transpose, dataFT masked) in one pass over the flags.
template <bool COPY, bool ZERO>
__global__ __launch_bounds__(256) void k_transpose_u8w(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       uint8_t* __restrict__ copy, float* __restrict__ data, int R, int C,
                                                       size_t src_ws, size_t dst_ws, size_t copy_ws, size_t data_ws) {
    __shared__ unsigned tile[128][17];
    const uint8_t* s = src + (size_t)blockIdx.z * src_ws;
    uint8_t* d = dst + (size_t)blockIdx.z * dst_ws;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 128;
    const int tid = threadIdx.x;
    {
        const int tx = tid & 15, ty = tid >> 4;
        const int c = c0 + 4 * tx;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int rr = ty + 16 * j, r = r0 + rr;
            unsigned v = 0;
            if (r < R && c < C) {
                v = *reinterpret_cast<const unsigned*>(s + (size_t)r * C + c);
                if (COPY) *reinterpret_cast<unsigned*>(copy + (size_t)blockIdx.z * copy_ws + (size_t)r * C + c) = v;
            }
            tile[rr][tx] = v;
        }
    }
    __syncthreads();
    const int bx = tid & 31, q0 = tid >> 5;
    const int r = r0 + 4 * bx;   // four source rows = one output word
    if (r >= R) return;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int q = q0 + 8 * h;
        const unsigned a0 = tile[4 * bx][q], a1 = tile[4 * bx + 1][q], a2 = tile[4 * bx + 2][q], a3 = tile[4 * bx + 3][q];
        // byte b of a_k = src[r + k][c0 + 4q + b]; output word b = bytes b of a0..a3
        const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
        const unsigned u0 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), u1 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
        const unsigned o[4] = {__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                               __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u)};
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = c0 + 4 * q + b;
            if (c >= C) break;
            *reinterpret_cast<unsigned*>(d + (size_t)c * R + r) = o[b];
            if (ZERO && o[b]) {
                const unsigned f = o[b];
                float4* pd = reinterpret_cast<float4*>(data + (size_t)blockIdx.z * data_ws + (size_t)c * R + r);
                float4 x = *pd;
                // the running flags only grow: from the second iteration on most of these are zero already
                const bool dirty = ((f & 0xFFu) && __float_as_uint(x.x)) || ((f & 0xFF00u) && __float_as_uint(x.y)) ||
                                   ((f & 0xFF0000u) && __float_as_uint(x.z)) || ((f & 0xFF000000u) && __float_as_uint(x.w));
                if (dirty)
                    *pd = make_float4((f & 0xFFu) ? 0.0f : x.x, (f & 0xFF00u) ? 0.0f : x.y, (f & 0xFF0000u) ? 0.0f : x.z,
                                      (f & 0xFF000000u) ? 0.0f : x.w);
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
    if (MODE != 1 && nanflag && isnan(bg)) nanflag[win * (size_t)C + (i % C)] = 1;
    if (MODE == 1) bg = fabsf(data[win * ws_data + i] - bg);
    o[win * ws_wo + i] = bg;
    // MODE 2: additionally the signed residual data - bg (flagging.py:962), written over
    // the weight image, which is dead from here on
    if (MODE == 2) const_cast<float*>(w)[win * ws_wo + i] = data[win * ws_data + i] - bg;
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
                            const uint8_t* __restrict__ nanflag, const float* __restrict__ data,
                            size_t ws_data, float* __restrict__ resid) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (nanflag && !nanflag[(size_t)blockIdx.y * C + c]) return;   // no NaN in this line
    float* x = a + (size_t)blockIdx.y * ws + c;
    const size_t Cs = (size_t)C;
    int last = -1;       // index of the last valid sample
    float lastv = 0.0f;
    int run = 0;         // start of the current NaN run
    // The scan is sequential, but the samples AHEAD of it are never rewritten (repairs only touch
    // positions behind the scan): read them 64 at a time, so that a line costs L / 64 memory
    // round trips instead of L (SKA-sized windows have 65536-long lines on a few thousand threads).
    constexpr int PFI = 64;
    for (int i0 = 0; i0 < L; i0 += PFI) {
        float vv[PFI];
#pragma unroll
        for (int u = 0; u < PFI; u++) vv[u] = (i0 + u < L) ? x[(size_t)(i0 + u) * Cs] : NAN;
#pragma unroll
        for (int u = 0; u < PFI; u++) {
            const int i = i0 + u;
            const float v = vv[u];
            if (i >= L || isnan(v)) continue;
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
    }
    if (run < L) {
        float fill = last < 0 ? 0.0f : lastv;  // all NaN -> zeros; else extrapolate forwards
        for (int j = run; j < L; j++) x[(size_t)j * Cs] = fill;
    }
    // the residual of a repaired line was formed from the NaN background: redo it
    if (resid) {
        const float* d = data + (size_t)blockIdx.y * ws_data + c;
        float* rr = resid + (size_t)blockIdx.y * ws + c;
        int j = 0;
        for (; j + PFI <= L; j += PFI) {
            float dv[PFI], xv[PFI];
#pragma unroll
            for (int u = 0; u < PFI; u++) { dv[u] = d[(size_t)(j + u) * Cs]; xv[u] = x[(size_t)(j + u) * Cs]; }
#pragma unroll
            for (int u = 0; u < PFI; u++) rr[(size_t)(j + u) * Cs] = dv[u] - xv[u];
        }
        for (; j < L; j++) rr[(size_t)j * Cs] = d[(size_t)j * Cs] - x[(size_t)j * Cs];
    }
}

// K6p: the same interpolation with the line cut into segments of INTERP_SEG positions that are
// repaired independently (a line of an SKA-sized window is 65536 channels long and there may be
// only a few hundred lines: one thread per line is a long, latency-bound walk).
//   k_interp_scan: per (segment, line) the first and the last valid position and whether any NaN
//   k_interp_fix : repairs the NaNs of a segment; the valid neighbours of runs that touch its ends
//                  come from the table (valid samples are never rewritten, NaNs only by their own
//                  segment's thread, so there is no ordering between segments)
// Every repaired value is computed by k_colinterp's expressions from the same (last valid, next
// valid) pair -- identical results.  The residual data - background is redone only where the
// background was repaired.  tab: int [W][nseg][3][C].  grid (ceil(C/64), nseg, W), block 64
#define INTERP_SEG 512
__global__ void k_interp_scan(const float* __restrict__ a, int L, int C, size_t ws, const uint8_t* __restrict__ nanflag,
                              int* __restrict__ tab) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int seg = blockIdx.y, nseg = gridDim.y;
    const int s0 = seg * INTERP_SEG, s1 = min(L, s0 + INTERP_SEG);
    int* t = tab + (((size_t)blockIdx.z * nseg + seg) * 3) * C + c;
    if (nanflag && !nanflag[(size_t)blockIdx.z * C + c]) {   // no NaN anywhere in this line
        t[0] = s0; t[C] = s1 - 1; t[2 * (size_t)C] = 0;
        return;
    }
    const float* x = a + (size_t)blockIdx.z * ws + c;
    const size_t Cs = (size_t)C;
    int first = -1, last = -1, nans = 0;
    constexpr int PFI = 64;
    for (int i0 = s0; i0 < s1; i0 += PFI) {
        float vv[PFI];
#pragma unroll
        for (int u = 0; u < PFI; u++) vv[u] = (i0 + u < s1) ? x[(size_t)(i0 + u) * Cs] : 0.0f;
#pragma unroll
        for (int u = 0; u < PFI; u++) {
            const bool nn = isnan(vv[u]);                      // (padding beyond s1 is valid-looking but not counted)
            if (i0 + u < s1) {
                if (nn) nans = 1;
                else { if (first < 0) first = i0 + u; last = i0 + u; }
            }
        }
    }
    t[0] = first; t[C] = last; t[2 * (size_t)C] = nans;
}

__global__ void k_interp_fix(float* __restrict__ a, int L, int C, size_t ws, const int* __restrict__ tab,
                             const float* __restrict__ data, size_t ws_data, float* __restrict__ resid) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int seg = blockIdx.y, nseg = gridDim.y;
    const int* tw = tab + (size_t)blockIdx.z * nseg * 3 * C + c;   // entry (sg, k) at (sg * 3 + k) * C
    if (!tw[((size_t)seg * 3 + 2) * C]) return;                  // nothing to repair here
    const int s0 = seg * INTERP_SEG, s1 = min(L, s0 + INTERP_SEG);
    float* x = a + (size_t)blockIdx.z * ws + c;
    const float* d = resid ? data + (size_t)blockIdx.z * ws_data + c : nullptr;
    float* rr = resid ? resid + (size_t)blockIdx.z * ws + c : nullptr;
    const size_t Cs = (size_t)C;
    auto put = [&](int j, float v) {
        x[(size_t)j * Cs] = v;
        if (resid) rr[(size_t)j * Cs] = d[(size_t)j * Cs] - v;
    };
    // positions [lo, hi) between the valid samples (last, lastv) and (i, v); last < 0: none before
    auto fill = [&](int lo, int hi, int last, float lastv, int i, float v) {
        if (last < 0) {
            for (int j = lo; j < hi; j++) put(j, v);             // extrapolate backwards
        } else {
            float diff = v - lastv;
            double grad = (double)diff / (double)(i - last);
            for (int j = lo; j < hi; j++) put(j, (float)((double)lastv + (double)(j - last) * grad));
        }
    };
    int last = -1;
    for (int sg = seg - 1; sg >= 0 && last < 0; sg--) last = tw[((size_t)sg * 3 + 1) * C];
    float lastv = last >= 0 ? x[(size_t)last * Cs] : 0.0f;
    int run = s0;
    constexpr int PFI = 64;
    for (int i0 = s0; i0 < s1; i0 += PFI) {
        float vv[PFI];
#pragma unroll
        for (int u = 0; u < PFI; u++) vv[u] = (i0 + u < s1) ? x[(size_t)(i0 + u) * Cs] : NAN;
#pragma unroll
        for (int u = 0; u < PFI; u++) {
            const int i = i0 + u;
            const float v = vv[u];
            if (i >= s1 || isnan(v)) continue;
            if (run < i) fill(run, i, last, lastv, i, v);
            last = i;
            lastv = v;
            run = i + 1;
        }
    }
    if (run < s1) {                                              // the run continues into the next segments
        int nxt = -1;
        for (int sg = seg + 1; sg < nseg && nxt < 0; sg++) nxt = tw[((size_t)sg * 3) * C];
        if (nxt >= 0) {
            fill(run, s1, last, lastv, nxt, x[(size_t)nxt * Cs]);
        } else {
            const float f = last < 0 ? 0.0f : lastv;             // all NaN -> zeros; else extrapolate forwards
            for (int j = run; j < s1; j++) put(j, f);
        }
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

// Amplitudes of a batch, computed ONCE per call when no channel averaging is
// requested: |vis| does not change between major iterations, only the flags
// do.  A NaN amplitude is flagged by every iteration's _average_freq
// (flagging.py:856-861), so it is folded into the running flags here.  The
// time-axis filter masks flagged samples itself (its flags contain the running
// flags); the FT copy is masked in place by k_zero_flagged4 every iteration --
// the running flags only grow, so zeros once written stay valid
// (flagged -> sum 0, count 0 -> value 0, flagging.py:858-870).
template <int VD>
__global__ void k_amplitude4(const void* __restrict__ vis, float* __restrict__ ampl,
                             uint8_t* __restrict__ iter, uint16_t* __restrict__ nanmask, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 samples, flat over the batch
    if (i >= n4) return;
    float a[4];
    // vn: the reference's final isnan(in_data) (flagging.py:777-781) -- of the VISIBILITY, i.e. either part
    // NaN; it differs from isnan(amplitude) for a sample with one infinite and one NaN part, whose
    // amplitude is +inf (C99 hypot) and so is NOT flagged by _average_freq (flagging.py:856-861)
    bool vn[4];
    if (VD == TRI_VIS_C64) {
        float4 z0 = reinterpret_cast<const float4*>(vis)[2 * i];
        float4 z1 = reinterpret_cast<const float4*>(vis)[2 * i + 1];
        a[0] = tri_hypotf(z0.x, z0.y); a[1] = tri_hypotf(z0.z, z0.w);
        a[2] = tri_hypotf(z1.x, z1.y); a[3] = tri_hypotf(z1.z, z1.w);
        vn[0] = isnan(z0.x) || isnan(z0.y); vn[1] = isnan(z0.z) || isnan(z0.w);
        vn[2] = isnan(z1.x) || isnan(z1.y); vn[3] = isnan(z1.z) || isnan(z1.w);
    } else {
        float4 z = reinterpret_cast<const float4*>(vis)[i];
        a[0] = fabsf(z.x); a[1] = fabsf(z.y); a[2] = fabsf(z.z); a[3] = fabsf(z.w);
        vn[0] = isnan(z.x); vn[1] = isnan(z.y); vn[2] = isnan(z.z); vn[3] = isnan(z.w);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = (0.0f + a[k]) / 1.0f;   // factor 1: sum = 0 + a, count 1 (flagging.py:858-870)
    reinterpret_cast<float4*>(ampl)[i] = make_float4(a[0], a[1], a[2], a[3]);
    const unsigned nanb = (isnan(a[0]) ? 1u : 0u) | (isnan(a[1]) ? 0x100u : 0u) | (isnan(a[2]) ? 0x10000u : 0u) |
                          (isnan(a[3]) ? 0x1000000u : 0u);
    if (nanb) reinterpret_cast<unsigned*>(iter)[i] |= nanb;
    // NaN bitmap, 16 samples (4 neighbouring threads; n4 % 4 == 0) per word: k_final16's NaN test
    // then costs 1/8 B per sample instead of re-reading the visibilities
    unsigned m = ((vn[0] ? 1u : 0u) | (vn[1] ? 2u : 0u) | (vn[2] ? 4u : 0u) | (vn[3] ? 8u : 0u)) << (4 * (threadIdx.x & 3));
    m |= __shfl_xor(m, 1, 64);
    m |= __shfl_xor(m, 2, 64);
    if ((threadIdx.x & 3) == 0) nanmask[i >> 2] = (uint16_t)m;
}

// data[i] = 0 where flags[i] != 0, four samples per thread; groups without a flag
// cost one byte-quad read and nothing else
__global__ void k_zero_flagged4(const uint8_t* __restrict__ flags, float* __restrict__ data, size_t n4per,
                                size_t ws_flags, size_t ws_data) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4per) return;
    size_t win = blockIdx.y;
    const unsigned f = reinterpret_cast<const unsigned*>(flags + win * ws_flags)[i];
    if (f == 0) return;
    float4* pd = reinterpret_cast<float4*>(data + win * ws_data) + i;
    float4 d = *pd;
    *pd = make_float4((f & 0xFFu) ? 0.0f : d.x, (f & 0xFF00u) ? 0.0f : d.y, (f & 0xFF0000u) ? 0.0f : d.z,
                      (f & 0xFF000000u) ? 0.0f : d.w);
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

// flags[w][t][f..f+15] |= spec_rows[w][f..f+15] | more[w][t][f..f+15]
__global__ void k_or_spec_more16(uint8_t* __restrict__ flags, const uint8_t* __restrict__ spec_rows,
                                 const uint8_t* __restrict__ more, int T, int Fa16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * Fa16) return;
    size_t win = blockIdx.y;
    int f16 = (int)(i % Fa16);
    uint4 sp = reinterpret_cast<const uint4*>(spec_rows + win * (size_t)Fa16 * 16)[f16];
    size_t a = win * (size_t)T * Fa16 + i;
    uint4* pf = reinterpret_cast<uint4*>(flags) + a;
    *pf = or4(or4(*pf, sp), reinterpret_cast<const uint4*>(more)[a]);
}

// The same update on the FT image: flagsFT[w][f][t..t+15] |= spec_rows[w][f].
// A spectral flag covers its channel's whole row, so only the rows of flagged
// channels are written -- the FT image stays current without transposing the
// TF one again.
__global__ void k_or_spec_ft16(uint8_t* __restrict__ flagsFT, const uint8_t* __restrict__ spec_rows, int T16, int Fa) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T16 * Fa) return;
    size_t win = blockIdx.y;
    int f = (int)(i / T16);
    if (!spec_rows[win * (size_t)Fa + f]) return;
    reinterpret_cast<uint4*>(flagsFT + win * (size_t)T16 * 16 * Fa)[i] = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
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

// k_combine16 followed by k_unaverage16<-1, 3> in one pass (time smearing over
// [t + lo, t + hi), then frequency smearing over [f - 1, f + 1]) without the
// intermediate image: the two bytes a 16-channel group needs from its
// neighbours come from the adjacent lanes (or, at wave edges, from memory).
// grid (ceil(F16/64), T, W), block 64
// TPANEL: the time-axis SumThreshold flags arrive as column panels [F / 64][T][64] (see k_transpose<T, true>): a 16-channel
// group is still 16 contiguous bytes, at ((f16 >> 2) * T + t) * 4 + (f16 & 3) in 16-byte units.
template <bool TPANEL>
__global__ __launch_bounds__(64) void k_combine_dilate16(const uint8_t* __restrict__ spec_rows, const uint8_t* __restrict__ tflags,
                                                        const uint8_t* __restrict__ fflags, uint8_t* __restrict__ dil,
                                                        int* __restrict__ rowcnt, int T, int F16, int lo, int hi) {
    const int lane = threadIdx.x;
    const int f16 = blockIdx.x * 64 + lane;
    const int t = blockIdx.y;
    const size_t win = blockIdx.z;
    const size_t base = win * (size_t)T * F16;
    const int t0 = max(t + lo, 0), t1 = min(t + hi, T);
    const bool in = f16 < F16;
    uint4 c = make_uint4(0, 0, 0, 0);
    if (in && t1 > t0) {
        c = reinterpret_cast<const uint4*>(spec_rows + win * (size_t)F16 * 16)[f16];
        const uint4* tp = reinterpret_cast<const uint4*>(tflags) + base;
        const uint4* fp = reinterpret_cast<const uint4*>(fflags) + base;
        for (int tt = t0; tt < t1; tt++) {
            size_t a = (size_t)tt * F16 + f16;
            size_t at = TPANEL ? ((size_t)(f16 >> 2) * T + tt) * 4 + (f16 & 3) : a;
            c = or4(c, or4(tp[at], fp[a]));
        }
    }
    auto edge = [&](int g, int b) -> unsigned {   // combined byte b of group g
        if (g < 0 || g >= F16 || t1 <= t0) return 0u;
        unsigned v = spec_rows[(win * (size_t)F16 + g) * 16 + b];
        for (int tt = t0; tt < t1; tt++) {
            size_t a = (base + (size_t)tt * F16 + g) * 16 + b;
            size_t at = TPANEL ? (base + ((size_t)(g >> 2) * T + tt) * 4 + (g & 3)) * 16 + b : a;
            v |= (unsigned)tflags[at] | (unsigned)fflags[a];
        }
        return v;
    };
    unsigned prev = __shfl_up(c.w >> 24, 1, 64), next = __shfl_down(c.x & 0xFFu, 1, 64);
    if (lane == 0) prev = edge(f16 - 1, 15);
    if (lane == 63) next = edge(f16 + 1, 0);
    int cnt = 0;
    if (in) {
        const unsigned w[4] = {c.x, c.y, c.z, c.w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned left = (w[j] << 8) | (j > 0 ? w[j - 1] >> 24 : prev & 0xFFu);
            const unsigned right = (w[j] >> 8) | ((j < 3 ? w[j + 1] & 0xFFu : next & 0xFFu) << 24);
            o[j] = (w[j] | left | right) & 0x01010101u;
            cnt += __popc(o[j]);
        }
        reinterpret_cast<uint4*>(dil)[base + (size_t)t * F16 + f16] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if (lane == 0 && cnt) atomicAdd(&rowcnt[win * (size_t)T + t], cnt);
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

// k_final, 16 samples per thread.  VD == TRI_VIS_NANMASK: `vis` is k_amplitude4's
// NaN bitmap of the batch (one uint16 per thread here).
#define TRI_VIS_NANMASK (-1000)
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
    unsigned nm = 0;
    if (VD == TRI_VIS_NANMASK) nm = reinterpret_cast<const uint16_t*>(vis)[a16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int4 c4 = cc[q];
        int cv[4] = {c4.x, c4.y, c4.z, c4.w};
        unsigned nanb = 0;
        if (VD == TRI_VIS_NANMASK) {
            const unsigned b = nm >> (4 * q);
            nanb = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
        } else if (VD == TRI_VIS_C64) {
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
    if (MODE != 1 && nanflag) {
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
    if (MODE == 2) {   // signed residual over the (dead) weight image
        float4 dv = reinterpret_cast<const float4*>(data + win * ws_data)[i];
        reinterpret_cast<float4*>(const_cast<float*>(w) + win * ws_wo)[i] =
            make_float4(dv.x - bg[0], dv.y - bg[1], dv.z - bg[2], dv.w - bg[3]);
    }
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
// k_reject4 fused with the TF4 re-pack that follows it in the rejection loop: a workgroup takes a tile of
// 64 lines (channels) x 64 flag words (256 times), updates the FT flag words in place and writes the same
// words transposed -- [T/4][C] -- for the next time-axis filter (one pass instead of two, 7 instead of 8 B/sample).
// grid (ceil(C4 / 64), ceil(L / 64), W), block (64, 4); L lines of C4 words
__global__ void __launch_bounds__(256)
k_reject4_t(const float* __restrict__ resid, uint8_t* __restrict__ flags, uint8_t* __restrict__ flags_t4,
            const double* __restrict__ med, const int* __restrict__ chunk_of, double scale, int L, int C4, int G,
            size_t ws_resid, size_t ws_flags) {
    __shared__ unsigned tile[64][65];
    const size_t win = blockIdx.z;
    const int w0 = blockIdx.x * 64, l0 = blockIdx.y * 64;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const float4* r4 = reinterpret_cast<const float4*>(resid + win * ws_resid);
    unsigned* fw = reinterpret_cast<unsigned*>(flags + win * ws_flags);
    unsigned* ft = reinterpret_cast<unsigned*>(flags_t4 + win * ws_flags);
    for (int j = ty; j < 64; j += 4) {
        const int l = l0 + j, w = w0 + tx;
        if (l < L && w < C4) {
            const double thr = med[win * G + chunk_of[l]] * scale;
            const size_t i = (size_t)l * C4 + w;
            const float4 rv = r4[i];
            unsigned f = fw[i];
            if ((double)rv.x > thr) f = (f & 0xFFFFFF00u) | 0x00000001u;
            if ((double)rv.y > thr) f = (f & 0xFFFF00FFu) | 0x00000100u;
            if ((double)rv.z > thr) f = (f & 0xFF00FFFFu) | 0x00010000u;
            if ((double)rv.w > thr) f = (f & 0x00FFFFFFu) | 0x01000000u;
            fw[i] = f;
            tile[j][tx] = f;
        }
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int w = w0 + j, l = l0 + tx;
        if (l < L && w < C4) ft[(size_t)w * L + l] = tile[tx][j];
    }
}

// The rejection where the residual lies as ROWS [T][ld] (the exact row filter's output) and the background flags as
// TF4 words [T / 4][C]: one thread per word = four times of one channel (flagging.py:567-574).  No FT image involved.
// grid (ceil(C / 256), T4, W)
__global__ void __launch_bounds__(256)
k_reject_tf(const float* __restrict__ resid, unsigned* __restrict__ flags_t4, const double* __restrict__ med,
            const int* __restrict__ chunk_of, double scale, int T4, int C, int ld, int G, size_t ws_resid, size_t ws_words) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int q = blockIdx.y;
    const size_t win = blockIdx.z;
    const double thr = med[win * G + chunk_of[c]] * scale;
    const float* rr = resid + win * ws_resid + (size_t)(4 * q) * ld + c;
    unsigned* pw = flags_t4 + win * ws_words + (size_t)q * C + c;
    unsigned f = *pw;
    if ((double)rr[0] > thr) f = (f & 0xFFFFFF00u) | 0x00000001u;
    if ((double)rr[(size_t)ld] > thr) f = (f & 0xFFFF00FFu) | 0x00000100u;
    if ((double)rr[(size_t)2 * ld] > thr) f = (f & 0xFF00FFFFu) | 0x00010000u;
    if ((double)rr[(size_t)3 * ld] > thr) f = (f & 0x00FFFFFFu) | 0x01000000u;
    *pw = f;
}

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

// NC = 1, 2 or 4 correlations: the (chan, corr) samples of a thread are one 8 / 16 / 32-byte piece of the row --
// vector loads (the scalar form reads every cache line of the row once per correlation)
template <int NC>
__global__ void k_pack_v(const float2* __restrict__ data, const uint8_t* __restrict__ flag,
                         const int32_t* __restrict__ row_bl, const int32_t* __restrict__ row_time,
                         int nchan, int nbl, int ntime, float2* __restrict__ vw, uint8_t* __restrict__ fw) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    if (f >= nchan) return;
    const int bl = row_bl[r], t = row_time[r];
    if (bl < 0 || bl >= nbl || t < 0 || t >= ntime) return;
    const size_t i = (r * nchan + f) * (size_t)NC;
    float2 v[NC];
    uint8_t fl[NC];
    if (NC == 4) {
        const float4 a = reinterpret_cast<const float4*>(data + i)[0], b = reinterpret_cast<const float4*>(data + i)[1];
        v[0] = make_float2(a.x, a.y); v[1 % NC] = make_float2(a.z, a.w);
        v[2 % NC] = make_float2(b.x, b.y); v[3 % NC] = make_float2(b.z, b.w);
        const uchar4 q = *reinterpret_cast<const uchar4*>(flag + i);
        fl[0] = q.x; fl[1 % NC] = q.y; fl[2 % NC] = q.z; fl[3 % NC] = q.w;
    } else if (NC == 2) {
        const float4 a = *reinterpret_cast<const float4*>(data + i);
        v[0] = make_float2(a.x, a.y); v[1 % NC] = make_float2(a.z, a.w);
        const uchar2 q = *reinterpret_cast<const uchar2*>(flag + i);
        fl[0] = q.x; fl[1 % NC] = q.y;
    } else {
        v[0] = data[i];
        fl[0] = flag[i];
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const size_t o = (((size_t)bl * NC + c) * ntime + t) * (size_t)nchan + f;
        vw[o] = v[c];
        fw[o] = fl[c];
    }
}

template <int NC>
__global__ void k_unpack_v(const uint8_t* __restrict__ fw, const int32_t* __restrict__ row_bl,
                           const int32_t* __restrict__ row_time, int nchan, int nbl, int ntime,
                           uint8_t* __restrict__ out, int any_corr) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    if (f >= nchan) return;
    const int bl = row_bl[r], t = row_time[r];
    const bool ok = !(bl < 0 || bl >= nbl || t < 0 || t >= ntime);
    uint8_t v[NC];
    uint8_t any = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        v[c] = ok ? fw[(((size_t)bl * NC + c) * ntime + t) * (size_t)nchan + f] : 0;
        any |= v[c] ? 1 : 0;
    }
    const size_t i = (r * nchan + f) * (size_t)NC;
    if (NC == 4) *reinterpret_cast<uchar4*>(out + i) = any_corr ? make_uchar4(any, any, any, any) : make_uchar4(v[0], v[1 % NC], v[2 % NC], v[3 % NC]);
    else if (NC == 2) *reinterpret_cast<uchar2*>(out + i) = any_corr ? make_uchar2(any, any) : make_uchar2(v[0], v[1 % NC]);
    else out[i] = any_corr ? any : v[0];
}

// any_corr: "flag entire visibility if any correlations are flagged"
// (apps/tricolour/app.py:479-480) fused into the gather
__global__ void k_unpack(const uint8_t* __restrict__ fw, const int32_t* __restrict__ row_bl,
                         const int32_t* __restrict__ row_time, int nchan, int ncorr, int nbl,
                         int ntime, uint8_t* __restrict__ out, int any_corr) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    size_t r = blockIdx.y;
    if (f >= nchan) return;
    int bl = row_bl[r], t = row_time[r];
    bool ok = !(bl < 0 || bl >= nbl || t < 0 || t >= ntime);
    uint8_t any = 0;
    if (any_corr && ok)
        for (int c = 0; c < ncorr; c++)
            any |= fw[(((size_t)bl * ncorr + c) * ntime + t) * (size_t)nchan + f] ? 1 : 0;
    for (int c = 0; c < ncorr; c++) {
        size_t i = (r * nchan + f) * (size_t)ncorr + c;
        uint8_t v = ok ? fw[(((size_t)bl * ncorr + c) * ntime + t) * (size_t)nchan + f] : 0;
        out[i] = any_corr ? any : v;
    }
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


// ===========================================================================
// uvcontsub_flagger (flagging.py:989-1073) -- a pure-NumPy
// routine in the reference, evaluated here with the float32 semantics of
// NumPy >= 2 (complex64 FFT, float32 residual).  Tolerance parity only: the
// reference's FFT rounding cannot be reproduced bit for bit.
// ===========================================================================
// flagged-sample count per correlation product (all-flagged products are skipped, :1033-1035)
// grid (ceil(N/4096), n_cp), block 256
__global__ void k_uv_count(const uint8_t* __restrict__ rflags, unsigned* __restrict__ cnt, size_t N) {
    size_t cp = blockIdx.y;
    size_t i0 = (size_t)blockIdx.x * 4096;
    unsigned c = 0;
    for (size_t i = i0 + threadIdx.x; i < min(N, i0 + 4096); i += 256) c += rflags[cp * N + i] ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[cp], c);
}

// nanmean over time per channel (:1037-1044): complex64 accumulation in time
// order, count of samples that are neither flagged nor NaN; none -> 0.
// grid (ceil(F/256), n_cp)
__global__ void k_uv_mean(const float2* __restrict__ vis, const uint8_t* __restrict__ rflags,
                          float2* __restrict__ avg, int T, int F) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    size_t cp = blockIdx.y;
    size_t base = cp * (size_t)T * F + f;
    float sr = 0.0f, si = 0.0f;
    long long n = 0;
    for (int t = 0; t < T; t++) {
        size_t a = base + (size_t)t * F;
        float2 z = vis[a];
        if (!rflags[a] && !isnan(z.x) && !isnan(z.y)) { sr += z.x; si += z.y; n++; }
    }
    float2 m = make_float2(0.0f, 0.0f);
    if (n > 0) m = make_float2((float)((double)sr / (double)n), (float)((double)si / (double)n));
    avg[cp * (size_t)F + f] = m;
}

// keep the first K Fourier bins of the mean spectrum (:1046-1055): direct DFT
// in float64 (K is 20..25), result rounded to complex64.  One workgroup per
// correlation product.  grid (n_cp), block 256
__global__ void __launch_bounds__(256)
k_uv_lowpass(const float2* __restrict__ avg, float2* __restrict__ smooth, int F, int K) {
    __shared__ double xr[64], xi[64];
    __shared__ double red[2][4];
    size_t cp = blockIdx.x;
    const float2* a = avg + cp * (size_t)F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < K; k++) {
        double pr = 0.0, pi_ = 0.0;
        for (int f = tid; f < F; f += 256) {
            double s, c;
            long long kf = ((long long)k * f) % F;
            sincospi(2.0 * (double)kf / (double)F, &s, &c);
            double re = a[f].x, im = a[f].y;
            // (re + i im) * (c - i s)
            pr += re * c + im * s;
            pi_ += im * c - re * s;
        }
        for (int o = 32; o > 0; o >>= 1) { pr += __shfl_down(pr, o, 64); pi_ += __shfl_down(pi_, o, 64); }
        if (lane == 0) { red[0][wave] = pr; red[1][wave] = pi_; }
        __syncthreads();
        if (tid == 0) {
            xr[k] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            xi[k] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        }
        __syncthreads();
    }
    for (int f = tid; f < F; f += 256) {
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < K; k++) {
            double s, c;
            long long kf = ((long long)k * f) % F;
            sincospi(2.0 * (double)kf / (double)F, &s, &c);
            // X[k] * (c + i s)
            sr += xr[k] * c - xi[k] * s;
            si += xr[k] * s + xi[k] * c;
        }
        smooth[cp * (size_t)F + f] = make_float2((float)(sr / (double)F), (float)(si / (double)F));
    }
}

// absresidual = |vis - smooth| as float32 (:1056) and the mask of samples the
// MAD ignores: prior flags or NaN residual (:1058-1062).  grid (ceil(N/256), n_cp)
__global__ void k_uv_resid(const float2* __restrict__ vis, const uint8_t* __restrict__ rflags,
                           const float2* __restrict__ smooth, float* __restrict__ absres,
                           uint8_t* __restrict__ mflags, unsigned* __restrict__ cnt, int T, int F) {
    __shared__ unsigned s_cnt;
    const size_t N = (size_t)T * F;
    const size_t cp = blockIdx.y;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned nfl = 0;
    // 2048 samples per workgroup, 8 per thread
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const size_t i = (size_t)blockIdx.x * 2048 + (size_t)u * 256 + threadIdx.x;
        if (i < N) {
            const int f = (int)(i % F);
            const float2 z = vis[cp * N + i], s = smooth[cp * (size_t)F + f];
            const float r = tri_hypotf(z.x - s.x, z.y - s.y);
            const bool fl = rflags[cp * N + i] != 0;
            absres[cp * N + i] = r;
            mflags[cp * N + i] = (fl || isnan(r)) ? 1 : 0;
            nfl += fl ? 1u : 0u;
        }
    }
    // flagged samples of the product (all flagged -> the cycle leaves its flags alone, k_uv_apply): one
    // global atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) nfl += __shfl_down(nfl, o, 64);
    if ((threadIdx.x & 63) == 0 && nfl) atomicAdd(&s_cnt, nfl);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&cnt[cp], s_cnt);
}
// k_uv_resid, four channels per thread (F % 4 == 0, 16-byte aligned images): 2048 samples per workgroup
__global__ void k_uv_resid4(const float2* __restrict__ vis, const uint8_t* __restrict__ rflags,
                            const float2* __restrict__ smooth, float* __restrict__ absres,
                            uint8_t* __restrict__ mflags, unsigned* __restrict__ cnt, int T, int F) {
    __shared__ unsigned s_cnt;
    const size_t N4 = (size_t)T * F / 4;
    const int F4 = F / 4;
    const size_t cp = blockIdx.y;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned nfl = 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const size_t i = (size_t)blockIdx.x * 512 + (size_t)u * 256 + threadIdx.x;   // group of 4 samples
        if (i < N4) {
            const int f4 = (int)(i % F4);
            const float4* vp = reinterpret_cast<const float4*>(vis) + (cp * N4 + i) * 2;
            const float4* sp = reinterpret_cast<const float4*>(smooth) + (cp * (size_t)F4 + f4) * 2;
            const float4 z0 = vp[0], z1 = vp[1], s0 = sp[0], s1 = sp[1];
            const unsigned fw = reinterpret_cast<const unsigned*>(rflags)[cp * N4 + i];
            const float r0 = tri_hypotf(z0.x - s0.x, z0.y - s0.y), r1 = tri_hypotf(z0.z - s0.z, z0.w - s0.w);
            const float r2 = tri_hypotf(z1.x - s1.x, z1.y - s1.y), r3 = tri_hypotf(z1.z - s1.z, z1.w - s1.w);
            reinterpret_cast<float4*>(absres)[cp * N4 + i] = make_float4(r0, r1, r2, r3);
            const unsigned fl = ((fw & 0xFFu) ? 1u : 0u) | ((fw & 0xFF00u) ? 0x100u : 0u) | ((fw & 0xFF0000u) ? 0x10000u : 0u) |
                                ((fw & 0xFF000000u) ? 0x1000000u : 0u);
            const unsigned nn = (isnan(r0) ? 1u : 0u) | (isnan(r1) ? 0x100u : 0u) | (isnan(r2) ? 0x10000u : 0u) | (isnan(r3) ? 0x1000000u : 0u);
            reinterpret_cast<unsigned*>(mflags)[cp * N4 + i] = fl | nn;
            nfl += __popc(fl);
        }
    }
    for (int o = 32; o > 0; o >>= 1) nfl += __shfl_down(nfl, o, 64);
    if ((threadIdx.x & 63) == 0 && nfl) atomicAdd(&s_cnt, nfl);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&cnt[cp], s_cnt);
}
// (kept for the A/B of the difference image; the flagger forms | |residual| - median | inside the median kernels)
__global__ void k_uv_diff(const float* __restrict__ absres, const double* __restrict__ med1,
                          float* __restrict__ diff, size_t N) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    size_t cp = blockIdx.y;
    float m = (float)med1[cp];
    diff[cp * N + i] = fabsf(fabsf(absres[cp * N + i]) - m);
}

// newflags = absres > sigma * mad (:1065); cycles >= or_from OR with the prior
// flags, earlier ones replace them (:1067-1071); fully flagged products untouched.
__global__ void k_uv_apply(const float* __restrict__ absres, const double* __restrict__ mad,
                           const unsigned* __restrict__ cnt, uint8_t* __restrict__ rflags,
                           float sigma, int do_or, size_t N) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    size_t cp = blockIdx.y;
    if (cnt[cp] == (unsigned)N) return;
    float thr = sigma * (float)mad[cp];
    bool nf = absres[cp * N + i] > thr;
    uint8_t old = rflags[cp * N + i];
    rflags[cp * N + i] = do_or ? (uint8_t)((old || nf) ? 1 : 0) : (uint8_t)(nf ? 1 : 0);
}

// k_uv_apply, four samples per thread (N % 4 == 0, 16-byte aligned images)
__global__ void k_uv_apply4(const float* __restrict__ absres, const double* __restrict__ mad,
                            const unsigned* __restrict__ cnt, uint8_t* __restrict__ rflags,
                            float sigma, int do_or, size_t N4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N4) return;
    size_t cp = blockIdx.y;
    if (cnt[cp] == (unsigned)(N4 * 4)) return;
    const float thr = sigma * (float)mad[cp];
    const float4 a = reinterpret_cast<const float4*>(absres)[cp * N4 + i];
    unsigned* pf = reinterpret_cast<unsigned*>(rflags) + cp * N4 + i;
    const unsigned nf = (a.x > thr ? 1u : 0u) | (a.y > thr ? 0x100u : 0u) | (a.z > thr ? 0x10000u : 0u) | (a.w > thr ? 0x1000000u : 0u);
    if (do_or) {
        const unsigned old = *pf;                               // (0 / 1 bytes: normalised at the start of the call)
        if ((old | nf) != old) *pf = old | nf;
    } else {
        *pf = nf;
    }
}

// ---------------------------------------------------------------------------
// Flag counts of a (bl, corr, time, chan) window for the flag summary
// (window_statistics.py:12-66): every per-antenna / per-baseline / per-scan /
// per-field number of the reference is a sum of per-baseline counts, every
// per-channel-bin number a sum of per-channel counts, so one pass produces
//   per_bl[bl]     = # of set flags of baseline bl
//   per_chan[chan] = # of set flags of channel chan over bl, corr, time
// A block owns ROWS rows of one baseline and a tile of 4 * 256 channels; a
// thread counts four adjacent channels from 32-bit loads (VEC) or one channel
// (scalar fallback), then adds its column counts to per_chan and, through a
// wave + block reduction, its total to per_bl.
// grid (ceil(nchan / (VEC ? 1024 : 256)), ceil(rows / ROWS), nbl), block 256
// ---------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
k_window_counts(const uint8_t* __restrict__ flags, unsigned long long* __restrict__ per_bl,
                unsigned long long* __restrict__ per_chan, int rows, int nchan, int rows_per_block) {
    __shared__ unsigned long long part[4];
    const int bl = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(r0 + rows_per_block, rows);
    const uint8_t* base = flags + ((size_t)bl * rows + r0) * (size_t)nchan;
    unsigned long long mine = 0;
    if (VEC) {
        const int c4 = blockIdx.x * 256 + threadIdx.x;          // group of four channels
        if (c4 * 4 < nchan) {
            const unsigned* row = reinterpret_cast<const unsigned*>(base) + c4;
            const size_t stride = (size_t)nchan / 4;
            unsigned cnt[4] = {0, 0, 0, 0};
            unsigned acc = 0;                                    // four byte counters
            int held = 0;
            for (int r = r0; r < r1; r++, row += stride) {
                unsigned w = *row;
                // bytes != 0 -> 1 (flags are 0/1, anything else still counts once)
                unsigned t = ((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w;
                acc += (t >> 7) & 0x01010101u;
                if (++held == 255) {
#pragma unroll
                    for (int k = 0; k < 4; k++) cnt[k] += (acc >> (8 * k)) & 0xFFu;
                    acc = 0;
                    held = 0;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                cnt[k] += (acc >> (8 * k)) & 0xFFu;
                if (cnt[k]) atomicAdd(&per_chan[(size_t)c4 * 4 + k], (unsigned long long)cnt[k]);
                mine += cnt[k];
            }
        }
    } else {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < nchan) {
            unsigned cnt = 0;
            const uint8_t* q = base + c;
            for (int r = r0; r < r1; r++, q += nchan) cnt += (*q != 0) ? 1u : 0u;
            if (cnt) atomicAdd(&per_chan[c], (unsigned long long)cnt);
            mine = cnt;
        }
    }
    // block total -> per_bl
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = part[0] + part[1] + part[2] + part[3];
        if (tot) atomicAdd(&per_bl[bl], tot);
    }
}

// ---------------------------------------------------------------------------
// Polarised / unpolarised intensity of (row, chan, corr) visibilities
// (stokes.py:157-209 and :79-153) with numba's typing: every term
//   value = a * (s1 * vis[c1] + s2 * vis[c2])
// is formed in complex128 (int64 * complex64 promotes), |value| is the double
// hypot, pol = sum |value|^2 and unpol = sum |value| accumulate in float64, and
//   mode 0: out = sqrt(pol)            mode 1: out = unpol - sqrt(pol)
// is cast to the visibility dtype (imaginary part 0).  One thread per
// (row, chan); the correlations of a sample are contiguous.
// ---------------------------------------------------------------------------
#define TRI_MAX_STOKES_TERMS 4
struct StokesTerm {
    int c1, c2, s1, s2;
    double ar, ai;
};
struct StokesTerms {
    int n_pol, n_unpol;
    StokesTerm pol[TRI_MAX_STOKES_TERMS];
    StokesTerm unpol[TRI_MAX_STOKES_TERMS];
};

template <typename T>   // float (complex64) or double (complex128)
__global__ void k_stokes_intensity(const T* __restrict__ vis, T* __restrict__ out, size_t n, int ncorr,
                                   StokesTerms terms, int mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T* v = vis + i * (size_t)ncorr * 2;
    auto magnitude = [&](const StokesTerm& t) {
        // s * (re + i im) with s = (double)s + 0i, then the sum, then a * (...): numba's complex products
        const double x1r = (double)v[2 * t.c1], x1i = (double)v[2 * t.c1 + 1];
        const double x2r = (double)v[2 * t.c2], x2i = (double)v[2 * t.c2 + 1];
        const double s1 = (double)t.s1, s2 = (double)t.s2;
        const double p1r = s1 * x1r - 0.0 * x1i, p1i = s1 * x1i + 0.0 * x1r;
        const double p2r = s2 * x2r - 0.0 * x2i, p2i = s2 * x2i + 0.0 * x2r;
        const double sr = p1r + p2r, si = p1i + p2i;
        const double vr = t.ar * sr - t.ai * si, vi = t.ar * si + t.ai * sr;
        return hypot(vr, vi);
    };
    double pol = 0.0;
    for (int k = 0; k < terms.n_pol; k++) {
        const double m = magnitude(terms.pol[k]);
        pol += m * m;
    }
    double res = sqrt(pol);
    if (mode == 1) {
        double unpol = 0.0;
        for (int k = 0; k < terms.n_unpol; k++) unpol += magnitude(terms.unpol[k]);
        res = unpol - res;
    }
    out[2 * i] = (T)res;
    out[2 * i + 1] = (T)0;
}
