// kernels_boxline.hpp -- single-sweep box-Gaussian filter with the delay lines in REGISTERS
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once

// ---------------------------------------------------------------------------
// K4r  The single-sweep cascade of K4b (kernels_boxfilter.hpp: four causal running
// sums, float64 accumulators, float32 stage outputs, one pass over the line; the
// proof of equivalence with flagging.py:394-417 is in K4b's header) with each
// stage's 2r-deep delay line split in two:
//   * the first KS slots live in VGPRs.  The step loop is unrolled over a multiple
//     of KS steps, so the slot a step touches is a compile-time register: the
//     value that entered KS steps ago is read and the new one takes its place,
//     without a move;
//   * the remaining d = 2r - KS slots (d even, possibly 0) form a small LDS ring
//     addressed by the step counter, exactly as in K4b.
// A value therefore leaves the window after KS + d = 2r steps, as the reference's
// padded[i] does.  The register file of a CU is 512 KB against 160 KB of LDS, and
// a stage needs no LDS traffic at all when 2r == KS: the filter's occupancy is no
// longer capped by the LDS rings (1.5-2 waves per SIMD in K4b / K4b'', 2-10 waves
// per CU in the lane-per-stage form K4c), and the thread-per-line mapping needs a
// third of K4c's instructions per line step (no DPP hand-off, no per-lane selects).
//
// k_boxt   time-axis stage: line = time, column = channel (coalesced); input
//          (data, TF4-packed flags) as K4b's SRCMODE 2, one image per blockIdx.z;
//          the weight image (0/1 input) runs the cascade in int32 -- bit-identical
//          while (2r+1)^3 <= 2^24 and (2r+1)^4 < 2^31 (r <= 107): every stage
//          value is then an integer that float32 holds exactly and the last
//          stage's int32 -> float32 conversion rounds to nearest even like the
//          reference's float64 -> float32 store.
// k_boxf   frequency-axis stage fused with the masked division (K4b''): input images
//          stored transposed (TF), two waves per 64 lines (weight / data image),
//          outputs swapped through the staging tiles and finished as
//          |data - bg| (MODE 1) or bg + signed residual + NaN marks (MODE 2).
// ---------------------------------------------------------------------------
// The cascade state of one line: KS register slots per stage + running sums.
template <int KS, typename V, typename A>
struct BoxLine {
    V R1[KS], R2[KS], R3[KS], R4[KS];
    A s1, s2, s3, s4;
    V o1, o2, o3;
    V old1, old2, old3, old4;            // LDS part: trailing samples of the coming step
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KS; k++) { R1[k] = 0; R2[k] = 0; R3[k] = 0; R4[k] = 0; }
        s1 = 0; s2 = 0; s3 = 0; s4 = 0;
        o1 = 0; o2 = 0; o3 = 0;
        old1 = 0; old2 = 0; old3 = 0; old4 = 0;
    }
};

// One cascade step.  SLOT: register slot (compile-time after unrolling).  HASL: an LDS
// part of d slots follows the register part; `cell` is this step's LDS slot (written),
// `ncell` the next step's (read one step ahead: d >= 2, so not the cell written now).
// Returns the last stage's running sum converted to float32 (before any division).
template <int KS, typename V, typename A, bool HASL, bool FAST>
__device__ __forceinline__ float boxline_step(BoxLine<KS, V, A>& L, const int slot, const int m, const V xin,
                                              const int n, const int R2x, V* cell, const V* ncell, const int BT) {
    // stage inputs (see K4b): stage 4 takes out_3[t] for t >= 2r, stage 2 out_1[t] for t < n + 2r,
    // stage 1 the data for t < n (xin is zero beyond the line end by construction)
    const V in4 = (FAST || m - 3 >= R2x) ? L.o3 : (V)0;
    const V in3 = L.o2;
    const V in2 = (FAST || m - 1 < n + R2x) ? L.o1 : (V)0;
    const V in1 = xin;
    V t1 = L.R1[slot], t2 = L.R2[slot], t3 = L.R3[slot], t4 = L.R4[slot];   // entered KS steps ago
    L.R1[slot] = in1; L.R2[slot] = in2; L.R3[slot] = in3; L.R4[slot] = in4;
    if (HASL) {
        const V n1 = ncell[0], n2 = ncell[BT], n3 = ncell[2 * (size_t)BT], n4 = ncell[3 * (size_t)BT];
        cell[0] = t1; cell[BT] = t2; cell[2 * (size_t)BT] = t3; cell[3 * (size_t)BT] = t4;
        t1 = L.old1; t2 = L.old2; t3 = L.old3; t4 = L.old4;                  // entered 2r steps ago
        L.old1 = n1; L.old2 = n2; L.old3 = n3; L.old4 = n4;
    }
    L.s4 += (A)in4;
    const float out = (float)L.s4;
    L.s4 -= (A)t4;
    L.s3 += (A)in3;
    L.o3 = (V)L.s3;
    L.s3 -= (A)t3;
    L.s2 += (A)in2;
    L.o2 = (V)L.s2;
    L.s2 -= (A)t2;
    L.s1 += (A)in1;
    L.o1 = (V)L.s1;
    L.s1 -= (A)t1;
    return out;
}

// ---- time-axis stage ------------------------------------------------------
// Register budget per KS: rings 4 KS + prefetch 2 PF (+ PF / 4 flag words) + ~40.
//   KS  8 / 16: three waves per SIMD (168 registers), KS 32: two, KS 64 / 80: one (512)
// The 64 / 80-slot forms run ONE loop body with
// wave-uniform edge predicates (a second, predicate-free interior body would double a loop
// that is already ~40 KB of code; the instruction cache holds 64 KB).
#ifndef BOXR_PF_SMALL
#define BOXR_PF_SMALL 16                 // prefetch depth of the 8 / 16-slot forms
#endif
#ifndef BOXR_PF_32
#define BOXR_PF_32 16
#endif
#ifndef BOXR_SCHED_EVERY
#define BOXR_SCHED_EVERY 1               // > 0: a scheduling barrier after every so many steps (bounds live ranges)
#endif
#ifndef BOXR_WAVES_32
#define BOXR_WAVES_32 2
#endif
#ifndef BOXR_WAVES_16
#define BOXR_WAVES_16 3                  // waves per SIMD the 8 / 16-slot forms are compiled for
#endif
#define BOXR_KS_MAX 80
__host__ __device__ constexpr int boxr_pf(int ks) { return ks >= 64 ? 16 : (ks == 32 ? BOXR_PF_32 : BOXR_PF_SMALL); }
__host__ __device__ constexpr int boxr_waves(int ks) { return ks <= 16 ? BOXR_WAVES_16 : (ks <= 32 ? BOXR_WAVES_32 : 1); }
#ifndef BOXR_WAVES_F32
#define BOXR_WAVES_F32 2
#endif
__host__ __device__ constexpr int boxr_waves_f(int ks) { return ks < 32 ? 2 : (ks == 32 ? BOXR_WAVES_F32 : 1); }
__host__ __device__ constexpr int boxr_lcm(int a, int b) {
    int x = a, y = b;
    while (y) { int t = x % y; x = y; y = t; }
    return a / x * b;
}

// grid (ceil(C / 64), W), block 64, dynamic LDS 4 * d * 64 floats; IMG 0: weight image
// (0/1 input, int32 cascade), IMG 1: data image (flagged samples zeroed, float64 cascade)
template <int KS, bool HASL, int IMG>
__global__ void __launch_bounds__(64, boxr_waves(KS))
k_boxt(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
       float* __restrict__ dstImg, int n, int C, int r, float denom, size_t sws, size_t dws) {
    extern __shared__ float cf_ring[];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;                                  // no workgroup barrier in this kernel
    const size_t win = blockIdx.y;
    constexpr int BT = 64;
    const int R2x = 2 * r;
    const int d = R2x - KS;                              // LDS slots per stage (host: even, >= 2 when HASL, else 0)
    // buffer addressing (host: window below 2^31 bytes): lane offset c * 4 + scalar row offset,
    // no per-load 64-bit address arithmetic; data loads beyond the line end return 0 by the
    // descriptor's range check
    const unsigned rowb = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(srcData + win * sws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(srcFlags + win * sws), 0, (int)((unsigned)(n / 4) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstImg + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const int coff = c * 4;
    float* ring = cf_ring + threadIdx.x;                 // element (slot, p) at ((slot*4)+p)*BT
    if (HASL)
        for (int k = 0; k < 4 * d; k++) ring[(size_t)k * BT] = 0.0f;
    const int total = n + 4 * r + 3;
    constexpr int PF = boxr_pf(KS);
    constexpr int UNR = boxr_lcm(KS, PF);
    constexpr bool SPLIT = KS <= 32;                     // separate predicate-free interior body

    using V = typename std::conditional<IMG == 0, int, float>::type;
    using A = typename std::conditional<IMG == 0, int, double>::type;
    BoxLine<KS, V, A> L;
    L.init();
    V* cells = reinterpret_cast<V*>(ring);
    constexpr size_t BT4 = (size_t)4 * BT;
    int lslot = 0;
    float pre[PF];
    unsigned prew[PF / 4];
    auto issue = [&](int t0) {
#pragma unroll
        for (int q = 0; q < PF / 4; q++) {
            const int t = t0 + 4 * q;
            const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, coff, (int)((unsigned)(t >> 2) * rowb), 0);
            prew[q] = (t < n) ? w : 0x01010101u;         // beyond the line end: flagged (weight 0)
        }
        if (IMG == 1) {
#pragma unroll
            for (int u = 0; u < PF; u++)
                pre[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, coff, (int)((unsigned)(t0 + u) * rowb), 0));
        }
    };
    auto store = [&](int i, float y) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ors, coff, (int)((unsigned)i * rowb), 0);
    };
    issue(0);
    for (int mb = 0; mb < total; mb += UNR) {
#pragma unroll
        for (int b = 0; b < UNR / PF; b++) {
            const int m0 = mb + b * PF;
            if (m0 < total) {
                V cur[PF];
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const bool fl = ((prew[u >> 2] >> (8 * (u & 3))) & 0xFFu) != 0;
                    cur[u] = (IMG == 0) ? (fl ? (V)0 : (V)1) : (fl ? (V)0 : (V)pre[u]);
                }
                issue(m0 + PF);
                const bool fast = SPLIT && m0 >= 4 * r + 3 && m0 + PF <= n;
                if (fast) {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        V* cell = cells + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const V* ncell = cells + (size_t)lslot * BT4;
                        const float out = boxline_step<KS, V, A, HASL, true>(L, (b * PF + u) % KS, m0 + u, cur[u], n, R2x, cell, ncell, BT);
                        store(m0 + u - 3 - 4 * r, out / denom);
                        if (BOXR_SCHED_EVERY > 0 && (u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1)) == (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        V* cell = cells + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const V* ncell = cells + (size_t)lslot * BT4;
                        const float out = boxline_step<KS, V, A, HASL, false>(L, (b * PF + u) % KS, m0 + u, cur[u], n, R2x, cell, ncell, BT);
                        const int i = m0 + u - 3 - 4 * r;
                        if (i >= 0 && i < n) store(i, out / denom);
                        if (BOXR_SCHED_EVERY > 0 && (u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1)) == (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
}

// ---- frequency-axis stage fused with the masked division ---------------------------
// Input images stored transposed: line c is row c of a [C][ld] array (the time-axis
// stage's TF output), staged PF positions at a time through an LDS tile per image
// (row segments in, one value per thread and step out).  A workgroup of two waves
// filters BOTH images of 64 lines -- wave 0 the weight image, wave 1 the weight * data
// image -- each step leaves the last stage's output in the tile slot its input came
// from, and after every PF steps the two waves finish PF / 2 positions per line each:
//     w = W / d^4, o = O / d^4, bg = (w == 0) ? NaN : o / w      (flagging.py:419, 506-513)
//     MODE 1: dstO = |data - bg|                                  (rejection loop, :563-566)
//     MODE 2: dstO = bg, dstW = data - bg, nanflag[line] = 1 on a NaN   (:576-578, :962)
// The filtered images are never written.  Buffer addressing throughout (host: images and
// outputs below 2^31 bytes per window).
// grid (ceil(C / 64), W), block 128, dynamic LDS 2 * (4 * d * 64 + PF * 65) floats
template <int KS, bool HASL, int MODE>
__global__ void __launch_bounds__(128, boxr_waves_f(KS))
k_boxf(const float* __restrict__ srcW, const float* __restrict__ srcO,
       float* __restrict__ dstW, float* __restrict__ dstO, const float* __restrict__ data,
       int n, int C, int ld, int r, float denom, size_t sws_img, size_t dws, size_t ws_data,
       uint8_t* __restrict__ nanflag) {
    extern __shared__ float cf_ring[];
    constexpr int PF = boxr_pf(KS);
    constexpr int UNR = boxr_lcm(KS, PF);
    constexpr int BT = 64;
    constexpr int LPI = 64 / PF;                               // lines covered by one staging load instruction
    constexpr int TS = BT + 1;                                 // tile row stride (floats)
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: weight image, 1: data image
    const int lt = threadIdx.x & 63;
    const int c0 = blockIdx.x * BT;
    const int c = c0 + lt;
    const bool colok = c < C;
    const size_t win = blockIdx.y;
    const int R2x = 2 * r;
    const int d = R2x - KS;
    float* ring = cf_ring + (size_t)half * 4 * d * BT + lt;    // element (slot, p) at ((slot*4)+p)*BT
    float* tiles = cf_ring + (size_t)2 * 4 * d * BT;           // [2][PF][TS]
    float* tile = tiles + (size_t)half * PF * TS;
    if (HASL)
        for (int k = 0; k < 4 * d; k++) ring[(size_t)k * BT] = 0.0f;

    const unsigned ldb = (unsigned)ld * 4u, rowb = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((half == 0 ? srcW : srcO) + win * sws_img), 0, (int)((unsigned)C * ldb), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(data + win * ws_data), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstO + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstW + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);

    // staging: element e = j * 64 + lt of a [64 lines][PF positions] patch: line = e / PF,
    // position = e % PF -> PF consecutive lanes read PF * 4 contiguous bytes of one row.
    // Rows beyond C fall outside the descriptor and read 0; positions beyond n are masked.
    const int s_pos = lt % PF;
    const int s_line0 = lt / PF;                               // + LPI j
    const int s_off = (int)((unsigned)s_line0 * ldb) + s_pos * 4;
    float pre[PF];
    auto issue = [&](int t0) {
        const int sbase = (int)((unsigned)c0 * ldb) + t0 * 4;
        if (t0 + PF <= n) {
#pragma unroll
            for (int j = 0; j < PF; j++)
                pre[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, s_off, sbase + (int)((unsigned)(LPI * j) * ldb), 0));
        } else {
            const bool tok = t0 + s_pos < n;
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const float v = (t0 < n) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, s_off, sbase + (int)((unsigned)(LPI * j) * ldb), 0)) : 0.0f;
                pre[j] = tok ? v : 0.0f;
            }
        }
    };
    auto exchange = [&]() {
        __syncthreads();                                         // previous tile (and hand-over) fully consumed
#pragma unroll
        for (int j = 0; j < PF; j++) tile[s_pos * TS + LPI * j + s_line0] = pre[j];
        __syncthreads();
    };
    // this thread finishes positions u in [PF/2 half, PF/2 half + PF/2) of a block; their
    // data samples are requested at the top of the block
    const int coff = (colok ? c : 0) * 4;
    float dpre[PF / 2];
    auto issue_data = [&](int m0) {
#pragma unroll
        for (int k = 0; k < PF / 2; k++) {
            const int i = m0 + (PF / 2) * half + k - 3 - 4 * r;
            dpre[k] = (i >= 0 && i < n) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, coff, (int)((unsigned)i * rowb), 0)) : 0.0f;
        }
    };

    BoxLine<KS, float, double> L;
    L.init();
    constexpr size_t BT4 = (size_t)4 * BT;
    int lslot = 0;
    const int total = n + 4 * r + 3;
    bool line_nan = false;
    issue(0);
    for (int mb = 0; mb < total; mb += UNR) {
#pragma unroll
        for (int b = 0; b < UNR / PF; b++) {
            const int m0 = mb + b * PF;
            if (m0 < total) {                                    // uniform over the workgroup
                exchange();                                      // tile of positions [m0, m0 + PF) in LDS
                issue(m0 + PF);                                  // next tile's loads stay in flight during the arithmetic
                issue_data(m0);
                const bool fast = m0 >= 4 * r + 3 && m0 + PF <= n;
                // (the sample of step u + 1 is read before step u's LDS writes, so its latency hides
                //  behind a whole step of arithmetic)
                float xin = tile[lt];
                if (fast) {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        const float xnext = (u + 1 < PF) ? tile[(u + 1) * TS + lt] : 0.0f;
                        float* cell = ring + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const float* ncell = ring + (size_t)lslot * BT4;
                        tile[u * TS + lt] = boxline_step<KS, float, double, HASL, true>(L, (b * PF + u) % KS, m0 + u, xin, n, R2x, cell, ncell, BT);
                        xin = xnext;
                        if (BOXR_SCHED_EVERY > 0 && u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        const float xnext = (u + 1 < PF) ? tile[(u + 1) * TS + lt] : 0.0f;
                        float* cell = ring + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const float* ncell = ring + (size_t)lslot * BT4;
                        tile[u * TS + lt] = boxline_step<KS, float, double, HASL, false>(L, (b * PF + u) % KS, m0 + u, xin, n, R2x, cell, ncell, BT);
                        xin = xnext;
                        if (BOXR_SCHED_EVERY > 0 && u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (m0 + PF - 1 - 3 - 4 * r >= 0) {              // an output position in this block (uniform)
                    __syncthreads();
                    const float* tw = tiles;
                    const float* to = tiles + (size_t)PF * TS;
#pragma unroll
                    for (int k = 0; k < PF / 2; k++) {
                        const int u = (PF / 2) * half + k;
                        const int i = m0 + u - 3 - 4 * r;
                        if ((fast || (i >= 0 && i < n)) && colok) {
                            const float wv = tw[u * TS + lt] / denom;   // deferred flagging.py:419
                            const float ov = to[u * TS + lt] / denom;
                            const float bg = (wv == 0.0f) ? NAN : ov / wv;
                            const int so = (int)((unsigned)i * rowb);
                            if (MODE == 1) {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fabsf(dpre[k] - bg)), ors, coff, so, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bg), ors, coff, so, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dpre[k] - bg), wrs, coff, so, 0);
                                line_nan |= isnan(bg);
                            }
                        }
                        if (BOXR_SCHED_EVERY > 0) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    if (MODE == 2 && line_nan && colok) nanflag[win * (size_t)C + c] = 1;
}
