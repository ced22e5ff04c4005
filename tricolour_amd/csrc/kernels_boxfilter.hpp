// kernels_boxfilter.hpp -- the box-Gaussian filter in its four forms (in-place multi-pass, single sweep, transposed-input single sweep, lane-per-stage)
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once

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
#ifndef CF_U
#define CF_U 8
#endif
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
                int n, int C, int r, float denom, size_t sws_img, size_t sws, size_t dws, int intw) {
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
    float* ring = cf_ring + threadIdx.x;            // element (slot, p) at ((slot*4)+p)*BT
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

    // The cascade, generic in its arithmetic: float32 values with float64
    // accumulators (the reference's types), or -- for the WEIGHT image of the
    // time-axis stage, whose values are small integers throughout (0/1 in,
    // pass sums <= (2r+1)^4 <= 2^24, all exactly representable in float32 and
    // float64) -- plain int32, which yields bit-identical outputs at a
    // fraction of the FP64 issue cost.
    auto cascade = [&](auto intw_tag) {
    using V = typename std::conditional<decltype(intw_tag)::value, int, float>::type;
    using A = typename std::conditional<decltype(intw_tag)::value, int, double>::type;
    A s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    V o1 = 0, o2 = 0, o3 = 0;                    // stage outputs of the previous step
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
    // The delay lines are indexed by the step counter: every stage writes slot
    // m mod 2r at EVERY step (inputs outside a stage's range are zero and leave
    // its sum unchanged), so the slot being written holds exactly the sample
    // that leaves the window, all four stages share one wave-uniform slot
    // index, and only the stores are range-checked.  The trailing samples are
    // read one step ahead (2r >= 2, so that cell is not the one written now):
    // their LDS latency hides behind the arithmetic of the current step.
    V* cells = reinterpret_cast<V*>(ring);      // element (slot, p) at ((slot*4)+p)*BT
    const size_t BT4 = (size_t)4 * BT;
    int slot = 0;
    V old1 = 0, old2 = 0, old3 = 0, old4 = 0;   // rings start zeroed (all-zero bits in either type)

    auto step = [&](auto fastc, const int m, const float xin) {
        constexpr bool FAST = decltype(fastc)::value;
        V* cell = cells + (size_t)slot * BT4;
        slot = (slot + 1 == R2) ? 0 : slot + 1;
        const V* ncell = cells + (size_t)slot * BT4;
        const V nold1 = ncell[0], nold2 = ncell[BT], nold3 = ncell[2 * (size_t)BT], nold4 = ncell[3 * (size_t)BT];
        // stage 4 (time t4 = m - 3): input out_3[t4], present for 2r <= t4 < n + 4r
        // stage 3 (t3 = m - 2): input out_2[t3];  stage 2 (t2 = m - 1): input out_1[t2]
        // for t2 < n + 2r;  stage 1 (t1 = m): input data[t1] (zero beyond the line end)
        const V in4 = (FAST || m - 3 >= R2) ? o3 : (V)0;
        const V in3 = o2;
        const V in2 = (FAST || m - 1 < n + R2) ? o1 : (V)0;
        const V in1 = (FAST || m < n) ? (V)xin : (V)0;
        cell[3 * (size_t)BT] = in4;
        cell[2 * (size_t)BT] = in3;
        cell[BT] = in2;
        cell[0] = in1;
        s4 += (A)in4;
        const float out = (float)s4;
        s4 -= (A)old4;
        s3 += (A)in3;
        o3 = (V)s3;
        s3 -= (A)old3;
        s2 += (A)in2;
        o2 = (V)s2;
        s2 -= (A)old2;
        s1 += (A)in1;
        o1 = (V)s1;
        s1 -= (A)old1;
        old1 = nold1; old2 = nold2; old3 = nold3; old4 = nold4;
        const int i = m - 3 - 4 * r;
        if (FAST || (i >= 0 && i < n)) {
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
    };
    if (intw && img == 0) cascade(std::true_type{});
    else cascade(std::false_type{});
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
// K4b''  K4b' fused with the masked division that follows the frequency-axis
// stage of masked_gaussian_filter (flagging.py:419, 506-513, 563-566, 962): a
// workgroup of two waves filters BOTH images of 64 lines -- wave 0 the weight
// image, wave 1 the weight * data image -- and after every 32 steps the two
// waves swap their 32 outputs per line through the (then idle) staging tiles,
// so that each thread finishes 16 positions of its line:
//     w = W / d^4, o = O / d^4, bg = (w == 0) ? NaN : o / w
//     MODE 1: dstO = |data - bg|                 (rejection loop)
//     MODE 2: dstO = bg, dstW = data - bg, nanflag[line] = 1 on a NaN
// The filtered images themselves are never written: 16 B/sample of HBM traffic
// and a kernel less per masked filter.  Arithmetic of the cascade: K4b.
// grid (ceil(C/64), W), block 128, dynamic LDS 2 * (4 * 2r * 64 + 32 * 65) floats
// ---------------------------------------------------------------------------
#define CFF_BT 64
template <int MODE>
__global__ void __launch_bounds__(2 * CFF_BT)
k_colfilter_lds_tf(const float* __restrict__ srcW, const float* __restrict__ srcO,
                   float* __restrict__ dstW, float* __restrict__ dstO, const float* __restrict__ data,
                   int n, int C, int ld, int r, float denom, size_t sws_img, size_t dws, size_t ws_data,
                   uint8_t* __restrict__ nanflag) {
    extern __shared__ float cf_ring[];
    const int half = threadIdx.x >> 6;                           // 0: weight image, 1: data image
    const int lt = threadIdx.x & 63;
    const int c0 = blockIdx.x * CFF_BT;
    const int c = c0 + lt;
    const bool colok = c < C;
    const size_t win = blockIdx.y;
    const int R2 = 2 * r;
    const size_t Cs = (size_t)C;
    const float* src = (half == 0 ? srcW : srcO) + win * sws_img;
    float* ring = cf_ring + (size_t)half * 4 * R2 * CFF_BT + lt;  // element (slot, p) at ((slot*4)+p)*BT
    float* tiles = cf_ring + (size_t)2 * 4 * R2 * CFF_BT;         // [2][CFT_PF][CFF_BT + 1]
    float* tile = tiles + (size_t)half * CFT_PF * (CFF_BT + 1);
    for (int k = 0; k < 4 * R2; k++) ring[(size_t)k * CFF_BT] = 0.0f;

    // staging: element e = j * 64 + lt of a [64 lines][32 positions] patch:
    // line = e / 32, position = e % 32  ->  lanes 0..31 read 128 contiguous bytes
    const int s_pos = lt & 31;
    const int s_line0 = lt >> 5;                                 // + 2 j
    float pre[CFT_PF];
    auto issue = [&](int t0) {
#pragma unroll
        for (int j = 0; j < CFT_PF; j++) {
            int line = c0 + 2 * j + s_line0;
            int t = t0 + s_pos;
            pre[j] = (line < C && t < n) ? src[(size_t)line * ld + t] : 0.0f;
        }
    };
    auto exchange = [&]() {
        __syncthreads();                                         // previous tile (and hand-over) fully consumed
#pragma unroll
        for (int j = 0; j < CFT_PF; j++) tile[s_pos * (CFF_BT + 1) + 2 * j + s_line0] = pre[j];
        __syncthreads();
    };
    // this thread finishes positions u in [16 half, 16 half + 16) of a block; their
    // data samples are requested at the top of the block
    const float* dcol = data + win * ws_data + (colok ? c : 0);
    float dpre[CFT_PF / 2];
    auto issue_data = [&](int m0) {
#pragma unroll
        for (int k = 0; k < CFT_PF / 2; k++) {
            const int i = m0 + 16 * half + k - 3 - 4 * r;
            dpre[k] = (i >= 0 && i < n && colok) ? dcol[(size_t)i * Cs] : 0.0f;
        }
    };

    // Delay lines indexed by the step counter: every stage writes slot m mod 2r at
    // every step (inputs outside a stage's range are zero and leave its float64 sum
    // unchanged), so the slot being written holds exactly the sample that leaves the
    // window, all four stages share one wave-uniform slot index, and the four cells of
    // a slot sit at immediate offsets of one LDS address ((slot, stage) interleaved).
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    float o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
    int slot = 0;
    float old1 = 0.0f, old2 = 0.0f, old3 = 0.0f, old4 = 0.0f;    // trailing samples of the coming step
    const int total = n + 4 * r + 3;

    // one step of the cascade (K4b arithmetic); returns the last stage's output
    auto step = [&](auto fastc, const int m, const float xin) -> float {
        constexpr bool FAST = decltype(fastc)::value;
        float* cell = ring + (size_t)slot * 4 * CFF_BT;
        slot = (slot + 1 == R2) ? 0 : slot + 1;
        const float* ncell = ring + (size_t)slot * 4 * CFF_BT;
        // the next step's trailing samples (2r >= 2: not the cell written below)
        const float nold1 = ncell[0], nold2 = ncell[CFF_BT], nold3 = ncell[2 * CFF_BT], nold4 = ncell[3 * CFF_BT];
        const float in4 = (FAST || m - 3 >= R2) ? o3 : 0.0f;
        const float in3 = o2;
        const float in2 = (FAST || m - 1 < n + R2) ? o1 : 0.0f;
        const float in1 = xin;                                   // zero beyond the line end by construction
        cell[3 * CFF_BT] = in4;
        cell[2 * CFF_BT] = in3;
        cell[CFF_BT] = in2;
        cell[0] = in1;
        s4 += (double)in4;
        const float out = (float)s4;
        s4 -= (double)old4;
        s3 += (double)in3;
        o3 = (float)s3;
        s3 -= (double)old3;
        s2 += (double)in2;
        o2 = (float)s2;
        s2 -= (double)old2;
        s1 += (double)in1;
        o1 = (float)s1;
        s1 -= (double)old1;
        old1 = nold1; old2 = nold2; old3 = nold3; old4 = nold4;
        return out;
    };

    bool line_nan = false;
    issue(0);
    for (int m0 = 0; m0 < total; m0 += CFT_PF) {
        exchange();                 // tile of positions [m0, m0 + 32) in LDS
        issue(m0 + CFT_PF);         // next tile's loads stay in flight during the arithmetic
        issue_data(m0);
        // every step takes its sample from tile slot (u, line) and leaves the last
        // stage's output in the same slot (one wave per tile: LDS operations in order)
        const bool fast = m0 >= 4 * r + 3 && m0 + CFT_PF <= n;
        // (the sample of step u + 1 is read before step u's LDS writes, so its latency hides
        //  behind a whole step of arithmetic)
        float xin = tile[lt];
        if (fast) {
#pragma unroll
            for (int u = 0; u < CFT_PF; u++) {
                const float xnext = (u + 1 < CFT_PF) ? tile[(u + 1) * (CFF_BT + 1) + lt] : 0.0f;
                tile[u * (CFF_BT + 1) + lt] = step(std::true_type{}, m0 + u, xin);
                xin = xnext;
            }
        } else {
#pragma unroll
            for (int u = 0; u < CFT_PF; u++) {
                const float xnext = (u + 1 < CFT_PF) ? tile[(u + 1) * (CFF_BT + 1) + lt] : 0.0f;
                tile[u * (CFF_BT + 1) + lt] = step(std::false_type{}, m0 + u, xin);
                xin = xnext;
            }
        }
        if (m0 + CFT_PF - 1 - 3 - 4 * r < 0) continue;          // no output position in this block yet (uniform)
        __syncthreads();
        const float* tw = tiles;
        const float* to = tiles + (size_t)CFT_PF * (CFF_BT + 1);
#pragma unroll
        for (int k = 0; k < CFT_PF / 2; k++) {
            const int u = 16 * half + k;
            const int i = m0 + u - 3 - 4 * r;
            if (!(fast || (i >= 0 && i < n)) || !colok) continue;
            const float wv = tw[u * (CFF_BT + 1) + lt] / denom;   // deferred flagging.py:419
            const float ov = to[u * (CFF_BT + 1) + lt] / denom;
            const float bg = (wv == 0.0f) ? NAN : ov / wv;
            if (MODE == 1) {
                dstO[win * dws + (size_t)i * Cs + c] = fabsf(dpre[k] - bg);
            } else {
                dstO[win * dws + (size_t)i * Cs + c] = bg;
                dstW[win * dws + (size_t)i * Cs + c] = dpre[k] - bg;
                line_nan |= isnan(bg);
            }
        }
    }
    if (MODE == 2 && line_nan && colok) nanflag[win * Cs + c] = 1;
}

// ---------------------------------------------------------------------------
// K4c  "Lane-per-stage" form of the single-sweep box filter for medium radii
// (the four 2r-deep delay lines of K4b no longer fit LDS at useful occupancy):
// the four cascade stages of one line run in the four lanes of a quad, every
// thread owning ONE stage and ONE LDS delay line, so the same LDS holds four
// times the threads.  Stage p takes its input from lane p-1's output of the
// previous step (DPP quad shuffle); the quad's four lanes prefetch four
// consecutive line positions per load instruction and the stage-0 lane picks
// them up by DPP broadcast.  Arithmetic per stage is identical to K4b (same
// causal running sums, same order) -- only the thread that executes a stage
// differs.  One wave (16 lines) per workgroup; no barriers.
//   * The delay lines are indexed by the wave's step counter m (write slot
//     m mod Rc, read slot (m - 2r) mod Rc, Rc = 2r rounded up to a multiple of
//     32), so every LDS address is lane offset + wave-uniform offset, and the
//     32 trailing samples of a block of steps are read in one burst before the
//     block's writes (none of them can hit a slot read later in the block).
//   * Every lane updates its running sum at every step (inputs outside a
//     stage's range are zero, which leaves the float64 sum unchanged), so only
//     the stores are range-checked, and interior blocks run without predicates.
//   * The last stage's outputs are handed round the quad: lane p keeps the
//     output of step 4k + p, and every fourth step all 64 lanes divide (when
//     the denominators are not deferred) and store one value each -- four
//     consecutive rows of the quad's line.
// grid (ceil(C/16), W, 2 images), block 64, dynamic LDS Rc * 64 floats (+ tile)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    // every lane of a quad_perm has a valid source lane: no fill value needed
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, false));
}
// quad_perm(a,b,c,d): lane i of each quad reads lane {a,b,c,d}[i]
#define QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

#ifndef LANE4_SKEW
#define LANE4_SKEW 2
#endif
__host__ __device__ inline int lane4_ring_capacity(int r) { return (2 * r + 31) / 32 * 32; }

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
    const bool colok_wave = (int)blockIdx.x * 16 + 16 <= C;   // all 16 lines of the wave exist
    const int cc = colok ? c : C - 1;               // out-of-range quads compute on a valid column, store nothing
    const size_t win = blockIdx.y;
    const int img = blockIdx.z;
    const int R2 = 2 * r;                           // >= 32 (host: r > LDS_R_MAX)
    const size_t Cs = (size_t)C;
    const float* src = SRCMODE == 1 ? ((img == 0 ? srcW : srcO) + win * sws_img + cc) : nullptr;
    // SRCMODE 3: input images stored transposed (line c = row c of a [C][ld]
    // array, ld passed in `sws`); the wave stages 32 positions of its 16 lines
    // through a small LDS tile (128-byte row segments in, one value per step out)
    const float* srct = SRCMODE == 3 ? ((img == 0 ? srcW : srcO) + win * sws_img) : nullptr;
    const int ld = (int)sws;
    const float* sd = (SRCMODE == 0 || SRCMODE == 2) ? srcData + win * sws + cc : nullptr;
    const unsigned* sf4 = SRCMODE == 2 ? reinterpret_cast<const unsigned*>(srcFlags) + win * (sws / 4) + cc : nullptr;
    float* dst = (img == 0 ? dstW : dstO) + win * dws + cc;
    const int Rc = lane4_ring_capacity(r);
    float* ring = cf_ring + lane;                   // slot k at ring[k * 64]
    float* tile = cf_ring + (size_t)Rc * 64;        // SRCMODE 3: [32 positions][16 lines + 1]
    for (int k = 0; k < Rc; k++) ring[k * 64] = 0.0f;

    // per-stage input range (see K4b): stage p takes the upstream value for its
    // time index t = m - p in [ilo, ihi), zero otherwise
    const int ilo = (p == 3) ? R2 : 0;
    const int ihi = (p == 0) ? n : ((p == 1) ? n + R2 : n + 4 * r);

    constexpr int PF = 32;                          // steps per block; each lane prefetches PF/4 positions
    float pre[PF / 4], cur[PF / 4];
    unsigned prew[PF / 4];
    // lane (line, p) loads positions t0 + 4 q + p
    auto issue = [&](int t0) {
        if (SRCMODE == 3) {
#pragma unroll
            for (int q = 0; q < PF / 4; q++) {
                const int e = q * 64 + lane;        // element of the [16 lines][32 positions] patch
                const int line = blockIdx.x * 16 + (e >> 5), t = t0 + (e & 31);
                pre[q] = (line < C && t < n) ? srct[(size_t)line * ld + t] : 0.0f;
            }
            return;
        }
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

    // Stage p runs SK steps behind stage p-1 (time index t = m - SK p) and takes
    // the upstream output of SK steps ago: consecutive steps are then tied only
    // by the running sum (two float64 adds), not by shuffle -> select ->
    // convert -> add -> convert through the quad.
    constexpr int SK = LANE4_SKEW;
    double s = 0.0;
    float oh[SK];                                   // this stage's outputs of the last SK steps
#pragma unroll
    for (int k = 0; k < SK; k++) oh[k] = 0.0f;
    float omine = 0.0f;                             // last-stage output kept by this lane
    const int total = n + 4 * r + 3 * SK;
    auto block = [&](auto fastc, const int m0) {
        constexpr bool fast = decltype(fastc)::value;
#pragma unroll
        for (int q = 0; q < PF / 4; q++) {
            if (SRCMODE == 2) {
                bool fl = ((prew[q] >> (8 * p)) & 0xFFu) != 0;   // byte p of the word = position 4q + p
                cur[q] = (img == 0) ? (fl ? 0.0f : 1.0f) : (fl ? 0.0f : pre[q]);
            } else {
                cur[q] = pre[q];
            }
        }
        if (SRCMODE == 3) {
            // registers -> tile (transposed); single-wave workgroup: LDS
            // operations of one wave execute in order
#pragma unroll
            for (int q = 0; q < PF / 4; q++) {
                const int e = q * 64 + lane;
                tile[(e & 31) * 17 + (e >> 5)] = cur[q];
            }
            __syncthreads();
        }
        issue(m0 + PF);
        // trailing samples of the 32 steps (written 2r steps ago), in one burst
        const int wp0 = m0 % Rc;                    // multiple of 32: no wrap inside the block
        int rp = (m0 + 8 * Rc - R2) % Rc;
        float oldv[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            oldv[u] = ring[rp * 64];
            rp = (rp + 1 == Rc) ? 0 : rp + 1;
        }
        float* wr = ring + wp0 * 64;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int m = m0 + u;
            // sample for stage 0: position m was loaded by lane (u & 3) of the quad
            float xs;
            if (SRCMODE == 3) xs = tile[u * 17 + (lane >> 2)];
            else if ((u & 3) == 0) xs = dpp_quad<QUAD_PERM(0, 0, 0, 0)>(cur[u >> 2]);
            else if ((u & 3) == 1) xs = dpp_quad<QUAD_PERM(1, 1, 1, 1)>(cur[u >> 2]);
            else if ((u & 3) == 2) xs = dpp_quad<QUAD_PERM(2, 2, 2, 2)>(cur[u >> 2]);
            else xs = dpp_quad<QUAD_PERM(3, 3, 3, 3)>(cur[u >> 2]);
            // upstream stage's output of SK steps ago
            const float up = dpp_quad<QUAD_PERM(0, 0, 1, 2)>(oh[u % SK]);
            float in = (p == 0) ? xs : up;
            if (!fast) {
                const int t = m - SK * p;           // this stage's time index
                in = (t >= ilo && t < ihi) ? in : 0.0f;
            }
            wr[u * 64] = in;
            s += (double)in;
            const float o_new = (float)s;
            oh[u % SK] = o_new;
            s -= (double)oldv[u];
            // the quad's lane 3 holds the filter output of line position m - 3 SK - 4r
            const float cap = dpp_quad<QUAD_PERM(3, 3, 3, 3)>(o_new);
            omine = ((u & 3) == p) ? cap : omine;
            if ((u & 3) == 3) {
                const int i = m - 3 + p - 3 * SK - 4 * r;   // lane p kept step m - 3 + p
                if (fast || (i >= 0 && i < n && colok)) dst[(size_t)i * Cs] = DIV ? omine / denom : omine;
            }
        }
        if (SRCMODE == 3) __syncthreads();          // tile fully consumed before it is rewritten
    };
    for (int m0 = 0; m0 < total; m0 += PF) {
        // interior: every stage inside its input range, every store inside the line
        const bool fast = colok_wave && m0 >= 4 * r + 3 * SK && m0 + PF <= n;
        if (fast) block(std::true_type{}, m0);
        else block(std::false_type{}, m0);
    }
}

