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
// ---------------------------------------------------------------------------
// Exact division by the launch constant b = float32(2r+1)**4 (flagging.py:419) through its
// correctly rounded reciprocal y = RN(1/b) (host, box_reciprocal()):
//     q = a y;  r = fma(-b, q, a);  q = fma(r, y, q);  r = fma(-b, q, a);  q = fma(r, y, q)
// Two Newton-style corrections with exact remainders (Markstein's scheme) give RN(a / b)
// whenever the first product a y is a normal number or +0; every other input (quotients in
// the subnormal range, -0, infinities, NaNs) is redone with the IEEE division, once per block
// of steps and only if some lane needed it.  tests/test_gpu_parity.py::test_division_by_box_denominator checks this predicate
// EXHAUSTIVELY -- all 2^32 float32 inputs for every radius the kernels accept -- against the
// hardware's correctly rounded division.  6 vector instructions instead of 14.
// ---------------------------------------------------------------------------
struct BoxDenom { float b, y; };
// Branch-free form: the quotient by the reciprocal scheme, and the lanes for which it is proven exact
// AND-ed into `okmask` (a wave lane mask in scalar registers).  The caller checks the mask once per block
// of steps and redoes the block's divisions with box_divide_ieee() if any active lane dropped out.
__device__ __forceinline__ float box_divide(float a, const BoxDenom dn, unsigned long long& okmask) {
    float q = a * dn.y;
    // class mask: -normal (bit 3), +0 (bit 6), +normal (bit 8)
    okmask &= __builtin_amdgcn_ballot_w64(__builtin_amdgcn_classf(q, 0x148));
    float r = __builtin_fmaf(-dn.b, q, a);
    q = __builtin_fmaf(r, dn.y, q);
    r = __builtin_fmaf(-dn.b, q, a);
    q = __builtin_fmaf(r, dn.y, q);
    return q;
}
__device__ __forceinline__ float box_divide_ieee(float a, const BoxDenom dn) { return a / dn.b; }
// (test hook form: one quotient, exact for every input)
__device__ __forceinline__ float box_divide_checked(float a, const BoxDenom dn) {
    unsigned long long ok = ~0ull;
    const float q = box_divide(a, dn, ok);
    return ((ok >> (threadIdx.x & 63)) & 1ull) ? q : box_divide_ieee(a, dn);
}

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
// (KS = 20: the whole delay line of r = 10 -- default.yaml's last time-axis radius -- in registers, no LDS part; blocks of 20 steps)
#ifndef BOXR_WAVES_20
#define BOXR_WAVES_20 2                  // (3: the data image's kernel spills 44 registers at 168 -- 44 ms instead of 11.7)
#endif
__host__ __device__ constexpr int boxr_pf(int ks) { return ks >= 64 ? 16 : (ks == 32 ? BOXR_PF_32 : (ks == 20 ? 20 : BOXR_PF_SMALL)); }
__host__ __device__ constexpr int boxr_waves(int ks) { return ks <= 16 ? BOXR_WAVES_16 : (ks == 20 ? BOXR_WAVES_20 : (ks <= 32 ? BOXR_WAVES_32 : 1)); }
#ifndef BOXR_WAVES_F32
#define BOXR_WAVES_F32 2
#endif
#ifndef BOXF_PF_SMALL
#define BOXF_PF_SMALL 16
#endif
__host__ __device__ constexpr int boxr_pf_f(int ks) { return ks <= 16 ? BOXF_PF_SMALL : boxr_pf(ks); }
#ifndef BOXR_WAVES_F16
#define BOXR_WAVES_F16 2
#endif
__host__ __device__ constexpr int boxr_waves_f(int ks) { return ks < 32 ? BOXR_WAVES_F16 : (ks == 32 ? BOXR_WAVES_F32 : 1); }
__host__ __device__ constexpr int boxr_lcm(int a, int b) {
    int x = a, y = b;
    while (y) { int t = x % y; x = y; y = t; }
    return a / x * b;
}

// IMG 0: weight image (0/1 input, int32 cascade), IMG 1: data image (flagged samples zeroed, float64
// cascade).  PACKED: flags as TF4 words [n / 4][C] (the 2-D path); else one byte per sample [n][C]
// (the spectrum path: lines = channels of the median spectra, columns = windows of the batch).
template <int KS, bool HASL, int IMG, bool PACKED, int PF>
__device__ __forceinline__ void boxt_body(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
                                          float* __restrict__ dstImg, int n, int C, int r, float denom, size_t sws, size_t dws,
                                          const size_t win) {
    extern __shared__ float cf_ring[];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;                                  // no workgroup barrier in this kernel
    constexpr int BT = 64;
    const int R2x = 2 * r;
    const int d = R2x - KS;                              // LDS slots per stage (host: even, >= 2 when HASL, else 0)
    // buffer addressing (host: window below 2^31 bytes): lane offset c * 4 + scalar row offset,
    // no per-load 64-bit address arithmetic
    const unsigned rowb = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(srcData + win * sws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(srcFlags + win * sws), 0, PACKED ? (int)((unsigned)(n / 4) * rowb) : (int)((unsigned)n * (unsigned)C), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstImg + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const int coff = c * 4;
    float* ring = cf_ring + threadIdx.x;                 // element (slot, p) at ((slot*4)+p)*BT
    if (HASL)
        for (int k = 0; k < 4 * d; k++) ring[(size_t)k * BT] = 0.0f;
    const int total = n + 4 * r + 3;
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
    unsigned prew[PACKED ? PF / 4 : PF];                 // packed: one word per four positions; else one byte each
    auto issue = [&](int t0) {
        if (PACKED) {
#pragma unroll
            for (int q = 0; q < PF / 4; q++) {
                const int t = t0 + 4 * q;                // (the scalar row offset is not range-checked: clamp it)
                const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, coff, t < n ? (int)((unsigned)(t >> 2) * rowb) : 0, 0);
                prew[q] = (t < n) ? w : 0x01010101u;     // beyond the line end: flagged (weight 0, data ignored)
            }
        } else {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int t = t0 + u;
                const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(frs, c, t < n ? (int)((unsigned)t * (unsigned)C) : 0, 0);
                prew[PACKED ? 0 : u] = (t < n) ? w : 1u;
            }
        }
        if (IMG == 1) {
#pragma unroll
            for (int u = 0; u < PF; u++)
                pre[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, coff, t0 + u < n ? (int)((unsigned)(t0 + u) * rowb) : 0, 0));
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
                    const bool fl = PACKED ? ((prew[u >> 2] >> (8 * (u & 3))) & 0xFFu) != 0 : prew[PACKED ? 0 : u] != 0;
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

// grid (ceil(C / 64), W), block 64, dynamic LDS 4 * d * 64 floats: one image per launch (2-D path)
template <int KS, bool HASL, int IMG>
__global__ void __launch_bounds__(64, boxr_waves(KS))
k_boxt(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
       float* __restrict__ dstImg, int n, int C, int r, float denom, size_t sws, size_t dws) {
    boxt_body<KS, HASL, IMG, true, boxr_pf(KS)>(srcData, srcFlags, dstImg, n, C, r, denom, sws, dws, blockIdx.y);
}

// Spectrum path: byte flags, both images in one launch (blockIdx.y = image): the lines are as long as the
// window has channels but there are only as many as the batch has windows, so what counts is the time per
// step of a lone wave -- no LDS round trip in the recurrence.  grid (ceil(C / 64), 2), block 64
// a lone wave hides HBM latency only through its own loads in flight: prefetch as deep as the registers allow
__host__ __device__ constexpr int boxt_spec_pf(int ks) { return ks <= 16 ? 64 : (ks == 32 ? 32 : 16); }
template <int KS, bool HASL>
__global__ void __launch_bounds__(64, 1)
k_boxt_spec(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
            float* __restrict__ dstW, float* __restrict__ dstO, int n, int C, int r, float denom) {
    if (blockIdx.y == 0) boxt_body<KS, HASL, 0, false, boxt_spec_pf(KS)>(srcData, srcFlags, dstW, n, C, r, denom, 0, 0, 0);
    else boxt_body<KS, HASL, 1, false, boxt_spec_pf(KS)>(srcData, srcFlags, dstO, n, C, r, denom, 0, 0, 0);
}

// ---- frequency-axis stage fused with the masked division ---------------------------
// Input images stored transposed: line c is row c of a [C][ld] array (the time-axis
// stage's TF output; the data image img_gap elements after the weight image), staged PF
// positions at a time through an LDS tile (row segments in, one value per lane and step out).  ONE WAVE filters BOTH images of 32 lines --
// lanes 0-31 the weight image, lanes 32-63 the weight * data image -- so nothing is
// shared between waves and the kernel has no workgroup barrier (LDS operations of one
// wave execute in order; a wave-level fence keeps the compiler from moving them).  Each
// step leaves the last stage's output in the tile slot its input came from, and after
// every PF steps each half of the wave finishes PF / 2 positions per line:
//     w = W / d^4, o = O / d^4, bg = (w == 0) ? NaN : o / w      (flagging.py:419, 506-513)
//     MODE 1: dstO = |data - bg|                                  (rejection loop, :563-566)
//     MODE 2: dstO = bg, dstW = data - bg, nanflag[line] = 1 on a NaN   (:576-578, :962)
// The filtered images are never written.  Buffer addressing throughout (host: images and
// outputs below 2^31 bytes per window).
// grid (ceil(C / 32), W), block 64, dynamic LDS 4 * d * 64 + 2 * PF * 34 floats
// tile row stride (floats): conflict-free column writes and row reads -- the staging writes go to rows s_pos (0 .. PF-1),
// columns LPI j + s_line0: 34 spreads 16 positions x 2 lines over the 32 banks, 33 does it for 32 positions x 1 line
__host__ __device__ constexpr int boxf_ts(int pf) { return pf >= 32 ? 33 : 34; }
#ifndef BOXF_DOUBLE_STAGE
#define BOXF_DOUBLE_STAGE 1
#endif
// blocks per staging load of the fused stage: whole 128-byte lines (2 x 16 positions) for the small radii
__host__ __device__ constexpr int boxf_nsub(int ks) { return (BOXF_DOUBLE_STAGE && ks <= 16 && boxr_pf_f(ks) == 16) ? 2 : 1; }
#define BOXF_TS 34
// cache policy of the fused stage's streaming accesses (amplitudes in, results out: touched once): 0 = default;
// 2 = non-temporal, so that they do not push the half-read input lines (64 of 128 bytes per block) out of L2
#ifndef BOXF_STREAM_AUX
#define BOXF_STREAM_AUX 0
#endif
template <int KS, bool HASL, int MODE, int WPS = boxr_waves_f(KS)>   // WPS: waves per SIMD the registers are budgeted for
__global__ void __launch_bounds__(64, WPS)
k_boxf(const float* __restrict__ srcW, unsigned img_gap,
       float* __restrict__ dstW, float* __restrict__ dstO, const float* __restrict__ data,
       int n, int C, int ld, int r, BoxDenom denom, size_t sws_img, size_t dws, size_t ws_data,
       uint8_t* __restrict__ nanflag) {
    extern __shared__ float cf_ring[];
    constexpr int PF = boxr_pf_f(KS);
    // Staging granularity: NSUB blocks of PF positions per load.  With PF = 16 a staging instruction takes 64 of a
    // line's 128 bytes and the other half is fetched again a block later -- 60 % of the time from HBM (traffic 1.30 x
    // algorithmic, profiles/r03_fetch_calibration.txt); NSUB = 2 stages whole 128-byte lines every second block.
    constexpr int NSUB = boxf_nsub(KS);
    constexpr int PL = PF * NSUB;                              // positions per staging load
    constexpr int UNR = boxr_lcm(boxr_lcm(KS, PF), PL);
    constexpr int BT = 64;
    constexpr int LPI = 32 / PL;                               // lines covered by one staging load instruction (per image)
    constexpr int TS = boxf_ts(PL);
    const int lane = threadIdx.x;
    const int half = lane >> 5;                                // 0: weight image, 1: data image
    const int hl = lane & 31;
    const int c0 = blockIdx.x * 32;
    const int c = c0 + hl;
    const bool colok = c < C;
    const size_t win = blockIdx.y;
    const int R2x = 2 * r;
    const int d = R2x - KS;
    float* ring = cf_ring + lane;                              // element (slot, p) at ((slot*4)+p)*BT
    float* tiles = cf_ring + (size_t)4 * d * BT;               // [2 images][PL][TS]
    float* tile = tiles + (size_t)half * PL * TS;
    if (HASL)
        for (int k = 0; k < 4 * d; k++) ring[(size_t)k * BT] = 0.0f;

    const unsigned rowb = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(data + win * ws_data), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstO + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dstW + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);

    // staging: element e = j * 32 + hl of this half's [32 lines][PF positions] patch: line = e / PF,
    // position = e % PF -> PF consecutive lanes read PF * 4 contiguous bytes of one row.
    // Lines beyond C and positions beyond n are masked.
    const int s_pos = hl % PL;
    const int s_line0 = hl / PL;                               // + LPI j
    // (the data image of a window starts img_gap elements after its weight image: one descriptor
    //  spans both, the data-image half of the wave adds the gap to its lane offset)
    const unsigned ldb = (unsigned)ld * 4u;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(srcW + win * sws_img), 0, (int)((img_gap + (unsigned)C * (unsigned)ld) * 4u), 0x00020000);
    const int s_off = (int)((unsigned)s_line0 * ldb) + s_pos * 4 + (half ? (int)(img_gap * 4u) : 0);
    float pre[PL];
    auto issue = [&](int t0) {
        const int sbase = (int)((unsigned)c0 * ldb) + t0 * 4;
        if (t0 + PL <= n && c0 + 32 <= C) {
#pragma unroll
            for (int j = 0; j < PL; j++)
                pre[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, s_off, sbase + (int)((unsigned)(LPI * j) * ldb), 0));
        } else {
            // lines beyond C / positions beyond n are masked (and their addresses kept inside the window)
            const bool tok = t0 + s_pos < n;
#pragma unroll
            for (int j = 0; j < PL; j++) {
                const bool ok = tok && (c0 + LPI * j + s_line0 < C);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, ok ? s_off + sbase + (int)((unsigned)(LPI * j) * ldb) : 0, 0, 0));
                pre[j] = ok ? v : 0.0f;
            }
        }
    };
    // LDS traffic of this wave only: order it against the compiler, not against other waves
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // this lane finishes positions u in [PF/2 half, PF/2 half + PF/2) of a block; their
    // data samples are requested at the top of the block.  Everything that depends on the
    // half sits in ONE lane offset (hoff) / one LDS base (eb): the per-position parts are
    // scalar row offsets and immediates.
    const int hrow = (PF / 2) * half;                            // first position of this half within a block
    const int hoff = (colok ? c : 0) * 4 + (int)((unsigned)hrow * rowb);
    const float* eb = tiles + (size_t)hrow * TS + hl;            // weight tile; the data tile is PL * TS further
    float dpre[PF / 2];
    auto issue_data = [&](int m0, bool fast) {
        const int i0 = m0 - 3 - 4 * r;                           // position of u = 0 (scalar)
        if (fast) {
#pragma unroll
            for (int k = 0; k < PF / 2; k++)
                dpre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, hoff, (int)((unsigned)(i0 + k) * rowb), BOXF_STREAM_AUX));
        } else {
#pragma unroll
            for (int k = 0; k < PF / 2; k++) {
                const int i = i0 + hrow + k;
                const bool ok = i >= 0 && i < n;
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? (colok ? c : 0) * 4 + (int)((unsigned)i * rowb) : 0, 0, 0));
                dpre[k] = ok ? v : 0.0f;
            }
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
            if (m0 < total) {
                const int ro = (b % NSUB) * PF;                  // this block's rows of the tile (static after unrolling)
                if (b % NSUB == 0) {
                    wave_sync();                                 // previous tile (and hand-over) fully consumed
#pragma unroll
                    for (int j = 0; j < PL; j++) tile[s_pos * TS + LPI * j + s_line0] = pre[j];
                    wave_sync();                                 // tile of positions [m0, m0 + PL) in LDS
                    issue(m0 + PL);                              // next tile's loads stay in flight during the arithmetic
                }
                const bool fast = m0 >= 4 * r + 3 && m0 + PF <= n;
                issue_data(m0, fast);
                // (the sample of step u + 1 is read before step u's LDS writes, so its latency hides
                //  behind a whole step of arithmetic)
                float xin = tile[ro * TS + hl];
                if (fast) {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        const float xnext = (u + 1 < PF) ? tile[(ro + u + 1) * TS + hl] : 0.0f;
                        float* cell = ring + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const float* ncell = ring + (size_t)lslot * BT4;
                        tile[(ro + u) * TS + hl] = boxline_step<KS, float, double, HASL, true>(L, (b * PF + u) % KS, m0 + u, xin, n, R2x, cell, ncell, BT);
                        xin = xnext;
                        if (BOXR_SCHED_EVERY > 0 && u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PF; u++) {
                        const float xnext = (u + 1 < PF) ? tile[(ro + u + 1) * TS + hl] : 0.0f;
                        float* cell = ring + (size_t)lslot * BT4;
                        if (HASL) lslot = (lslot + 1 == d) ? 0 : lslot + 1;
                        const float* ncell = ring + (size_t)lslot * BT4;
                        tile[(ro + u) * TS + hl] = boxline_step<KS, float, double, HASL, false>(L, (b * PF + u) % KS, m0 + u, xin, n, R2x, cell, ncell, BT);
                        xin = xnext;
                        if (BOXR_SCHED_EVERY > 0 && u % (BOXR_SCHED_EVERY > 0 ? BOXR_SCHED_EVERY : 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (m0 + PF - 1 - 3 - 4 * r >= 0) {              // an output position in this block (uniform)
                    wave_sync();                                 // both halves' outputs in the tiles
                    const int i0 = m0 - 3 - 4 * r;
                    unsigned long long okmask = ~0ull, active = 0;
                    auto finish = [&](auto ieee_tag) {
                        constexpr bool IEEE = decltype(ieee_tag)::value;
#pragma unroll
                        for (int k = 0; k < PF / 2; k++) {
                            const int i = i0 + hrow + k;
                            if ((fast || (i >= 0 && i < n)) && colok) {
                                if (!IEEE) active |= __builtin_amdgcn_ballot_w64(true);
                                // deferred flagging.py:419
                                const float wv = IEEE ? box_divide_ieee(eb[(ro + k) * TS], denom) : box_divide(eb[(ro + k) * TS], denom, okmask);
                                const float ov = IEEE ? box_divide_ieee(eb[(PL + ro + k) * TS], denom) : box_divide(eb[(PL + ro + k) * TS], denom, okmask);
                                const float bg = (wv == 0.0f) ? NAN : ov / wv;
                                // fast blocks: i0 + k >= 0, the half's rows ride in the lane offset
                                const int vo = fast ? hoff : (c * 4 + (int)((unsigned)i * rowb));
                                const int so = fast ? (int)((unsigned)(i0 + k) * rowb) : 0;
                                if (MODE == 1) {
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fabsf(dpre[k] - bg)), ors, vo, so, BOXF_STREAM_AUX);
                                } else {
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bg), ors, vo, so, BOXF_STREAM_AUX);
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dpre[k] - bg), wrs, vo, so, BOXF_STREAM_AUX);
                                    line_nan |= isnan(bg);
                                }
                            }
                            if (BOXR_SCHED_EVERY > 0) __builtin_amdgcn_sched_barrier(0);
                        }
                    };
                    finish(std::false_type{});
                    // some quotient fell outside the reciprocal scheme's proven range: redo the block (rare)
                    if ((okmask & active) != active) finish(std::true_type{});
                }
            }
        }
    }
    if (MODE == 2 && line_nan && colok) nanflag[win * (size_t)C + c] = 1;
}
