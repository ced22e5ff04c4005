// kernels_sumthreshold.hpp -- fused multi-window SumThreshold column kernels
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once

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
// K7p  The dynamic cascade as a STAGE PIPELINE across the waves of a workgroup (any window list of up to
// eight windows: final_st_very_broad's 32, 48, 64, 128).
// k_colst_dyn keeps every thread's prefix rings in global memory -- 2176 bytes per thread for those four
// windows, 1.1 GB across the resident threads, re-read every w positions: the kernel is bound by that HBM
// traffic (0.11 TB/s of algorithmic bytes).  Here a workgroup of nw waves owns 64 lines (lane = column):
// wave j runs ONLY window j, and the per-position (pos, neg) bits travel from stage to stage through a
// circular LDS byte buffer.
// NO PREFIX RING (round 3): S_k = cum[k + w] - cum[k] needs the prefix sum of w positions ago.  cum[k] is
// a deterministic function of the clamped values c_0 .. c_{k-1} added in order, so a SECOND accumulator that
// runs w positions behind the first and is fed the same clamped values reproduces cum[k] bit for bit:
// S_k = lead - lag.  The lagging accumulator re-reads x_k from memory (a row this workgroup streamed w
// positions ago: L2 / Infinity Cache) and re-derives the clamp from the flag byte of position k as it
// stands just before this stage emits position k -- the stages before are final there and nothing else
// has touched it, so it is the very byte the stage saw when it ingested x_k.  What used to be 139 KB of
// float64 rings per workgroup (one wave per SIMD) is two registers; LDS only holds the flag bytes.
// Positions are processed in blocks of 16 with one workgroup barrier per block; stage j + 1 runs
// ceil((w_j - 1) / 16) + 1 blocks behind stage j, i.e. behind every position stage j may still flag,
// so its clamp sees exactly the flags of the stages before it -- as in K7 / flagging.py:638-674.  Every
// float64 value is produced by the same operations in the same order as k_colst_dyn (and the reference).
// grid (ceil(C / 64), G, W), block 64 * nw, dynamic LDS stp_lds_bytes(sw)
// ---------------------------------------------------------------------------
#define STP_B 16
struct StPipe { int lag[TRI_MAX_WINDOWS + 1]; int accn; };
__host__ inline StPipe stp_plan(const StWin& sw) {
    StPipe pp{};
    int lag = 0;
    for (int j = 0; j < sw.nw; j++) {
        pp.lag[j] = lag;
        lag += (sw.w[j] - 1 + STP_B - 1) / STP_B + 1;
    }
    pp.lag[sw.nw] = lag;
    pp.accn = (lag + 2) * STP_B;
    return pp;
}
__host__ inline size_t stp_lds_bytes(const StWin& sw) {
    const StPipe pp = stp_plan(sw);
    return (size_t)pp.accn * 64;
}

#ifndef STP_MINWAVES
#define STP_MINWAVES 3                  // (tuning: minimum waves per SIMD the register allocation must allow; 1: 49.5 ms, 3: 41.7 ms, 4 spills: 103 ms per 880-window pass)
#endif
__global__ void __launch_bounds__(512, STP_MINWAVES)
k_colst_pipe(const float* __restrict__ data, const double* __restrict__ med, uint8_t* __restrict__ out,
             const int64_t* __restrict__ chunk_ends, StWin sw, StPipe pp, double thr_scale, int L, int C, int G,
             size_t ws_data, size_t ws_out) {
    extern __shared__ double stp_lds[];
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // this wave's window
    const int nw = sw.nw;
    const int g = blockIdx.y;
    const size_t win = blockIdx.z;
    const int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    if (c1 <= c0) return;                                      // (uniform: before any barrier)
    const int c = blockIdx.x * 64 + lane;
    const bool colok = c < C;
    const int cc = colok ? c : C - 1;
    // flagging.py:630-633 (slicing clamps to the axis length)
    const int p0 = max(c0 - sw.maxw + 1, 0);
    const int p1 = min(c1 + sw.maxw - 1, L);
    const int Lp = p1 - p0;
    const int w = sw.w[j];
    uint8_t* acc = reinterpret_cast<uint8_t*>(stp_lds) + lane; // [position % accn][64]
    const int accn = pp.accn;
    for (int s = threadIdx.x; s < accn * 64; s += blockDim.x) (acc - lane)[s] = 0;
    // flagging.py:622-628
    const float mad = (float)med[(win * (size_t)C + cc) * G + g];
    const float thr0 = isnan(mad) ? INFINITY : (float)((double)mad * thr_scale);
    const double limit = (double)thr0 / sw.tf[j];
    const double scale = sw.scale[j];
    const float* x = data + win * ws_data + cc;
    uint8_t* o = out + win * ws_out + c;
    const size_t Cs = (size_t)C;
    double cumlast = 0.0;                                      // cum[i + 1] after ingesting position i
    double cumlag = 0.0;                                       // cum[e], e = i + 1 - w: the same sum, w positions behind
    int sincep = 1 << 30, sincen = 1 << 30;
    int apos = 0;                                              // acc index of ingest position i
    int aemit = (accn - (w - 1) % accn) % accn;                // acc index of emit position e = i + 1 - w
    const int lag = pp.lag[j], lagF = pp.lag[nw];
    const int NBs = (Lp + sw.maxw + STP_B - 1) / STP_B;        // blocks until every stage has emitted every position
    const int NIT = NBs + lagF + 1;
    float xn[STP_B], xen[STP_B];                               // next block's ingest / emit samples
    auto fetch = [&](int b) {
        if (b >= 0 && b * STP_B + STP_B <= Lp) {               // whole block inside the line: no per-element test
            const float* xb = x + (size_t)(p0 + b * STP_B) * Cs;
#pragma unroll
            for (int u = 0; u < STP_B; u++) xn[u] = xb[(size_t)u * Cs];
        } else {
#pragma unroll
            for (int u = 0; u < STP_B; u++) {
                const int i = b * STP_B + u;
                xn[u] = (b >= 0 && i < Lp) ? x[(size_t)(p0 + i) * Cs] : 0.0f;
            }
        }
        const int e0 = b * STP_B + 1 - w;                      // the samples leaving the window: w - 1 rows back
        if (b >= 0 && e0 >= 0 && e0 + STP_B <= Lp) {
            const float* xb = x + (size_t)(p0 + e0) * Cs;
#pragma unroll
            for (int u = 0; u < STP_B; u++) xen[u] = xb[(size_t)u * Cs];
        } else {
#pragma unroll
            for (int u = 0; u < STP_B; u++) {
                const int e = e0 + u;
                xen[u] = (b >= 0 && e >= 0 && e < Lp) ? x[(size_t)(p0 + e) * Cs] : 0.0f;
            }
        }
    };
    // One block of 16 steps with every LDS read first and every LDS write last (windows of at least a block:
    // nothing the block reads was written in it -- a stage never sees its own flags).  ING / EMIT: every step
    // of the block ingests / emits (uniform per block), so the 16 steps are straight-line code.
    auto run_block = [&](const float (&xc)[STP_B], const float (&xe)[STP_B], auto ing_tag, auto emit_tag) {
        constexpr bool ING = decltype(ing_tag)::value, EMIT = decltype(emit_tag)::value;
        uint8_t ain[STP_B], aem[STP_B];
        int em[STP_B];
#pragma unroll
        for (int u = 0; u < STP_B; u++) {
            em[u] = aemit;
            if (ING) ain[u] = acc[(size_t)(apos + u) * 64];    // (accn is a multiple of 16: no wrap inside the block)
            if (EMIT) aem[u] = acc[(size_t)aemit * 64];
            aemit = aemit + 1 == accn ? 0 : aemit + 1;
        }
        apos = apos + STP_B == accn ? 0 : apos + STP_B;
        // Flags of earlier windows are sparse: when no line of the wave carries one anywhere in the block (neither at
        // the ingest nor at the emit positions) the clamps are the identity and the 16 steps are adds and compares only.
        unsigned any = 0;
#pragma unroll
        for (int u = 0; u < STP_B; u++) any |= (ING ? (unsigned)ain[u] : 0u) | (EMIT ? (unsigned)aem[u] : 0u);
        unsigned fresh = 0;                                     // bits this block adds
        auto steps = [&](auto clamp_tag) {
            constexpr bool CLAMP = decltype(clamp_tag)::value;
#pragma unroll
            for (int u = 0; u < STP_B; u++) {
                bool hp = false, hn = false;
                if (ING) {
                    double clamped = (double)xc[u];
                    if (CLAMP) {
                        const uint8_t a = ain[u];
                        const bool cp = (a & 1) && clamped > limit;
                        const bool cn = !cp && (a & 2) && clamped < -limit;
                        clamped = cp ? limit : (cn ? -limit : clamped);
                    }
                    const double cumnew = cumlast + clamped;
                    cumlast = cumnew;
                    if (EMIT) {
                        const double S = cumnew - cumlag;      // cum[i + 1] - cum[e]
                        hp = S * scale > limit;
                        hn = S * (-scale) > limit;
                        // c_e again, from the byte the stage saw when it ingested x_e (aem[u] before this stage's own bits)
                        double ce = (double)xe[u];
                        if (CLAMP) {
                            const uint8_t ae = aem[u];
                            const bool ep = (ae & 1) && ce > limit;
                            const bool en = !ep && (ae & 2) && ce < -limit;
                            ce = ep ? limit : (en ? -limit : ce);
                        }
                        cumlag = cumlag + ce;                  // cum[e + 1]
                    }
                }
                if (EMIT) {
                    // (counters start at 2^30 and a line has far fewer than 2^30 positions: no saturation needed here)
                    sincep = hp ? 0 : sincep + 1;
                    sincen = hn ? 0 : sincen + 1;
                    const unsigned add = (sincep < w ? 1u : 0u) | (sincen < w ? 2u : 0u);
                    fresh |= add;
                    aem[u] |= (uint8_t)add;
                }
            }
        };
        if (__builtin_amdgcn_ballot_w64(any != 0) == 0) steps(std::false_type{});
        else steps(std::true_type{});
        if (EMIT && __builtin_amdgcn_ballot_w64(fresh != 0) != 0) {      // (nothing new anywhere in the wave: bytes unchanged)
#pragma unroll
            for (int u = 0; u < STP_B; u++) acc[(size_t)em[u] * 64] = aem[u];
        }
    };
    fetch(-lag);                                               // (this wave's block 0 comes `lag` iterations in)
    __syncthreads();
    for (int it = 0; it < NIT; it++) {
        const int b = it - lag;                                // this stage's block
        float xc[STP_B], xe[STP_B];
#pragma unroll
        for (int u = 0; u < STP_B; u++) { xc[u] = xn[u]; xe[u] = xen[u]; }
        fetch(b + 1);
        const int i0 = b * STP_B, e0 = i0 + 1 - w;             // first ingest / emit position of the block
        const bool all_in = i0 + STP_B <= Lp, none_in = i0 >= Lp;
        const bool all_em = e0 >= 0 && e0 + STP_B <= Lp, none_em = e0 + STP_B <= 0;
        if (b >= 0 && b < NBs && w >= STP_B && all_in && all_em) {
            run_block(xc, xe, std::true_type{}, std::true_type{});
        } else if (b >= 0 && b < NBs && w >= STP_B && all_in && none_em) {
            run_block(xc, xe, std::true_type{}, std::false_type{});
        } else if (b >= 0 && b < NBs && w >= STP_B && none_in && all_em) {
            run_block(xc, xe, std::false_type{}, std::true_type{});
        } else if (b >= 0 && b < NBs) {
#pragma unroll
            for (int u = 0; u < STP_B; u++) {
                const int i = b * STP_B + u;                   // ingest position
                const int e = i + 1 - w;                       // emit position
                if (e < Lp) {
                    bool hp = false, hn = false;
                    if (i < Lp) {
                        const uint8_t a = acc[(size_t)apos * 64];
                        double clamped = (double)xc[u];
                        if ((a & 1) && clamped > limit) clamped = limit;
                        else if ((a & 2) && clamped < -limit) clamped = -limit;
                        const double cumnew = cumlast + clamped;
                        cumlast = cumnew;
                        if (e >= 0) {
                            const double S = cumnew - cumlag;
                            hp = S * scale > limit;
                            hn = S * (-scale) > limit;
                            const uint8_t ae = acc[(size_t)aemit * 64];     // (before this stage's own bits go in below)
                            double ce = (double)xe[u];
                            if ((ae & 1) && ce > limit) ce = limit;
                            else if ((ae & 2) && ce < -limit) ce = -limit;
                            cumlag = cumlag + ce;
                        }
                    }
                    if (e >= 0) {
                        sincep = hp ? 0 : min(sincep + 1, 1 << 30);
                        sincen = hn ? 0 : min(sincen + 1, 1 << 30);
                        const uint8_t add = (sincep < w ? 1 : 0) | (sincen < w ? 2 : 0);
                        if (add) acc[(size_t)aemit * 64] |= add;
                    }
                }
                apos = apos + 1 == accn ? 0 : apos + 1;
                aemit = aemit + 1 == accn ? 0 : aemit + 1;
            }
        }
        // positions final after the last stage: written out (and their slots recycled) by wave 0
        const int bf = it - lagF;
        if (j == 0 && bf >= 0) {
            const int f0 = bf * STP_B;
            if (f0 + STP_B <= Lp && p0 + f0 >= c0 && p0 + f0 + STP_B <= c1) {   // whole block inside the chunk
                const int as0 = f0 % accn;                     // (block-aligned: no wrap)
#pragma unroll
                for (int u = 0; u < STP_B; u++) {
                    const uint8_t a = acc[(size_t)(as0 + u) * 64];
                    acc[(size_t)(as0 + u) * 64] = 0;
                    if (colok) o[(size_t)(p0 + f0 + u) * Cs] = a ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < STP_B; u++) {
                    const int ef = f0 + u;
                    if (ef < Lp) {
                        const int as = ef % accn;
                        const uint8_t a = acc[(size_t)as * 64];
                        acc[(size_t)as * 64] = 0;
                        const int pos = p0 + ef;
                        if (pos >= c0 && pos < c1 && colok) o[(size_t)pos * Cs] = a ? 1 : 0;
                    }
                }
            }
        }
        __syncthreads();
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

template <int W0, int W1, int W2, int W3>
__global__ void __launch_bounds__(256, 2)
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
// K7c  Lane-mask SumThreshold cascade for windows {1,2,4,8}: the arithmetic of
// K7b (same float64 operations, same order), but every per-sample flag lives
// as a 64-bit wave lane mask in scalar registers instead of a 0/1 value per
// lane:
//   * fP[q], fN[q]: positive / negative flags of line position q (mod 16) for
//     the 64 lines of the wave.  A hit of stage j at ingest position i is
//     OR-ed straight into its w positions i-w+1..i (s_or_b64), so the
//     dilation needs no hit history and no vector instruction;
//   * the clamp of stage j >= 1 is  m = (fP & (x > thr)) | (fN & (x < -thr)),
//     cl = m ? copysign(thr, x) : x  -- two compares, one bit-field insert and
//     two selects;
//   * the last stage does not need the sign of its hits: it tests |S| > T
//     and ORs into a single ring fA = fP | fN, merged when a position reaches
//     the last stage (positive / negative flags only matter to later clamps);
//   * samples are converted to float64 once and kept in an 8-deep ring.
// The vector ALU is left with the float64 adds / compares and the selects
// (about 35 instructions per sample instead of 58 in K7b); the
// mask algebra (about 33 instructions) runs on the scalar unit, and all
// addressing is scalar: buffer loads / stores with a wave-uniform descriptor,
// a constant lane offset and a scalar row offset.
// Interior blocks of 16 ticks are branch-free and form their own loop; the
// samples of the NEXT block are requested in one burst at the top of a block,
// so every load has 16..31 ticks to land.
// grid (ceil(C/BLK), G, W), block BLK;  needs L * C * 4 < 2^31
// ---------------------------------------------------------------------------
#ifndef ST_MAXBLK
#define ST_MAXBLK 256
#endif
// PANEL (round 4): data and out are column panels [C / 64][L][64] (k_transpose<T, true>) -- the rows a wave visits are
// contiguous (256 bytes of samples, 64 bytes of flags each), its whole walk one linear stream instead of 256-byte / 64-byte
// pieces a row pitch apart (profiles/r03_stream_mix.txt: 5.3 against 4.3 TB/s for this byte mix).  Host: C % 64 == 0.
template <int W0, int W1, int W2, int W3, bool PANEL = false>
__global__ void __launch_bounds__(ST_MAXBLK, ST_MAXBLK >= 1024 ? 1 : 2)
k_colst_mask(const float* __restrict__ data, const double* __restrict__ med,
             uint8_t* __restrict__ out, const int64_t* __restrict__ chunk_ends,
             StFusedArgs fa, double thr_scale, int L, int C, int G, size_t ws_data,
             size_t ws_out) {
    constexpr int W[4] = {W0, W1, W2, W3};
    constexpr int D[4] = {0, W0, W0 + W1, W0 + W1 + W2};   // ingest delay of stage j
    constexpr int DOUT = D[3] + W3 - 1;                     // final flags lag the head by this
    constexpr int MAXW = W3;
    constexpr int UN = 16;                                   // ticks per unrolled block
    static_assert(W0 <= W1 && W1 <= W2 && W2 <= W3 && W3 <= 8, "windows must be sorted, <= 8");
    static_assert((W0 & (W0 - 1)) == 0 && (W1 & (W1 - 1)) == 0 && (W2 & (W2 - 1)) == 0 &&
                  (W3 & (W3 - 1)) == 0, "power-of-two windows");
    static_assert(W0 + W1 + W2 <= 7, "sample ring is 8 deep");
    static_assert(DOUT < UN, "flag ring is 16 deep");

    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int g = blockIdx.y;
    const size_t win = blockIdx.z;
    const int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    if (c1 <= c0) return;
    const size_t Cs = (size_t)C;

    float mad = (float)med[(win * Cs + c) * G + g];
    float thr0 = isnan(mad) ? INFINITY : (float)((double)mad * thr_scale);
    const int p0 = max(c0 - MAXW + 1, 0);
    const int p1 = min(c1 + MAXW - 1, L);
    const int Lp = p1 - p0;
    const int o0 = c0 - p0, o1 = c1 - p0;   // output interior in padded coordinates
    // Buffer addressing: a wave-uniform descriptor of the padded line block, a
    // constant 32-bit lane offset and a scalar row offset -- no vector
    // address arithmetic.
    const unsigned rowb = PANEL ? 256u : (unsigned)C * 4u;       // bytes from one row of samples to the next
    const unsigned orow = PANEL ? 64u : (unsigned)C;             // ... of flags
    const size_t panel = (size_t)__builtin_amdgcn_readfirstlane(c >> 6);
    const size_t first = PANEL ? (panel * (size_t)L + (size_t)p0) * 64 : (size_t)p0 * Cs;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(data + win * ws_data + first), 0, (int)((unsigned)Lp * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(out + win * ws_out + first), 0, (int)((unsigned)Lp * orow), 0x00020000);
    const int xoff = PANEL ? (c & 63) * 4 : c * 4, ooff = PANEL ? (c & 63) : c;
    auto ldo = [&](int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, soff, 0));
    };
    auto ld = [&](int row) { return ldo((int)((unsigned)row * rowb)); };

    // thr = f64(thr0) / rho^log2(w) (flagging.py:643); T = thr * w (exact)
    double thr[4], T[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        thr[j] = (double)thr0 / fa.tf[j];
        T[j] = thr[j] * (double)W[j];
    }
    double cum[4] = {0.0, 0.0, 0.0, 0.0};
    double r0[W0], r1[W1], r2[W2], r3[W3];
#pragma unroll
    for (int k = 0; k < W0; k++) r0[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W1; k++) r1[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W2; k++) r2[k] = 0.0;
#pragma unroll
    for (int k = 0; k < W3; k++) r3[k] = 0.0;
    uint64_t fP[UN], fN[UN], fA[UN];
#pragma unroll
    for (int k = 0; k < UN; k++) { fP[k] = 0; fN[k] = 0; fA[k] = 0; }
    double xd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) xd[k] = 0.0;

    const int nticks = Lp + DOUT;
    const int last = Lp - 1;
    float nxt[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) nxt[u] = ld(min(u, last));

    // interior blocks step two scalar byte offsets row by row; the empty asm
    // keeps each step one s_add instead of 16 precomputed multiples
    int xso = 0, oso = 0;
    auto block = [&](auto fastc, const int base) {
        constexpr bool fast = decltype(fastc)::value;
        float cur[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            cur[u] = nxt[u];
            if (fast) {
                nxt[u] = ldo(xso);
                xso += (int)rowb;
                asm("" : "+s"(xso));
            } else {
                // edge blocks: rows past the end re-read the last row, never ingested
                nxt[u] = ld(min(base + UN + u, last));
            }
        }
        if (fast) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int PH = 0; PH < UN; PH++) {
            const int n = base + PH;
            xd[PH & 7] = (double)cur[PH];
#pragma unroll
            for (int j = 3; j >= 0; j--) {
                const int w = W[j];
                const int i = n - D[j];        // ingest position
                if (fast || (i >= 0 && i < Lp)) {
                    const int q = (PH - D[j]) & (UN - 1);
                    const double x = xd[q & 7];
                    double cl = x;
#ifndef ST_ABLATE_CLAMP      // timing-only builds (scripts/st_floor.sh): results are wrong on purpose
                    if (j > 0) {
                        const uint64_t cp = fP[q] & __builtin_amdgcn_ballot_w64(x > thr[j]);
                        const uint64_t cn = fN[q] & __builtin_amdgcn_ballot_w64(x < -thr[j]);
                        const bool m = __builtin_amdgcn_inverse_ballot_w64(cp | cn);
                        cl = m ? __builtin_copysign(thr[j], x) : x;
                    }
#endif
                    const double cs = cum[j] + cl;
                    cum[j] = cs;
                    const int slot = (PH - D[j] + 1) & (w - 1);
                    double old;
                    if (j == 0) { old = r0[slot & (W0 - 1)]; r0[slot & (W0 - 1)] = cs; }
                    else if (j == 1) { old = r1[slot & (W1 - 1)]; r1[slot & (W1 - 1)] = cs; }
                    else if (j == 2) { old = r2[slot & (W2 - 1)]; r2[slot & (W2 - 1)] = cs; }
                    else { old = r3[slot & (W3 - 1)]; r3[slot & (W3 - 1)] = cs; }
                    const double S = cs - old;
                    const bool full = fast || i + 1 - w >= 0;   // window complete
#ifdef ST_ABLATE_HITS
                    if (j < 3) {
                        asm volatile("" :: "v"(S));      // keep the rolling sum alive (methodology rule 17)
                        if (j == 0) { fP[q] = 0; fN[q] = 0; }
                    } else
#endif
                    if (j < 3) {
                        uint64_t hp = __builtin_amdgcn_ballot_w64(S > T[j]);
                        uint64_t hn = __builtin_amdgcn_ballot_w64(S < -T[j]);
                        if (!full) { hp = 0; hn = 0; }
                        // the first stage opens position q: plain assignment, so that
                        // no mask of a recycled slot stays live around the loop
                        if (j == 0) { fP[q] = hp; fN[q] = hn; }
#pragma unroll
                        for (int k = (j == 0 ? 1 : 0); k < w; k++) {
                            fP[(q - k) & (UN - 1)] |= hp;
                            fN[(q - k) & (UN - 1)] |= hn;
                        }
                    } else {
                        // position q leaves the signed rings; later only the OR matters
                        uint64_t h = __builtin_amdgcn_ballot_w64(__builtin_fabs(S) > T[j]);
                        if (!full) h = 0;
                        fA[q] = fP[q] | fN[q] | h;
                        fP[q] = 0;
                        fN[q] = 0;
#pragma unroll
                        for (int k = 1; k < w; k++) fA[(q - k) & (UN - 1)] |= h;
                    }
                }
            }
            const int ef = n - DOUT;
            const int fs = (PH - DOUT) & (UN - 1);
            if (fast) {
                const bool f = __builtin_amdgcn_inverse_ballot_w64(fA[fs]);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(f ? 1 : 0), ors, ooff, oso, 0);
                oso += (int)orow;
                asm("" : "+s"(oso));
            } else if (ef >= o0 && ef < o1) {
                const bool f = __builtin_amdgcn_inverse_ballot_w64(fA[fs]);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(f ? 1 : 0), ors, ooff, (int)((unsigned)ef * orow), 0);
            }
            fA[fs] = 0;
            // keep every tick's store (and the scalar chains) in its own scheduling region
            if (fast) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // head (edge-checked) blocks, then the branch-free interior as its own
    // counted loop (its register assignment is not tied to the edge code),
    // then the edge-checked tail
    auto is_fast = [&](int base) {
        return base >= UN && base + 2 * UN - 1 < Lp && base - DOUT >= o0 && base + UN - 1 - DOUT < o1;
    };
    int base = 0;
    for (; base < nticks && !is_fast(base); base += UN) block(std::false_type{}, base);
    int nfast = 0;
    for (int b = base; b < nticks && is_fast(b); b += UN) nfast++;
    xso = (int)((unsigned)(base + UN) * rowb);
    oso = (int)((unsigned)(base - DOUT) * orow);
    for (int b = 0; b < nfast; b++) block(std::true_type{}, 0);   // interior blocks do not use base
    base += nfast * UN;
    for (; base < nticks; base += UN) block(std::false_type{}, base);
}
