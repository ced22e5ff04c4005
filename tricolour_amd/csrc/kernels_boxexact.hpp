// kernels_boxexact.hpp -- K4x: the frequency-axis box cascade for ANY radius, one line pair resident in LDS
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
//
// Every other filter kernel evaluates the reference's running sums one lane per line, in the reference's
// order: bit-exact by construction, but each line then needs its 4 x 2r-deep delay lines on chip, which
// stops at r = 110 (registers + LDS of a CU); beyond that the in-place multi-pass kernel moved 6 x the
// algorithmic bytes (final_st_very_broad: r = 277 / 221 / 166, a third of that call).
//
// This kernel maps LANES TO POSITIONS of one line instead and relies on a checked property:
//
//   A float64 sum of float32 terms is EXACT -- no rounding at all, hence independent of the order of the
//   additions -- when every term is an integer multiple of a quantum q and every partial sum stays below
//   2^53 q.  With non-negative terms, every partial sum of the terms of a window is bounded by the window's
//   total.  The reference's pass (flagging.py:404-416: s += x[i + 2r]; out[i] = float32(s); s -= x[i]) only
//   ever holds such partial sums, so under the condition it produces out[i] = RN32(exact window sum) -- and so
//   does any other summation order.
//
// A workgroup holds one line of both images (weight, weight * data) in LDS, zero-padded as the reference pads it
// (flagging.py:381-395: n + 4r positions, data at offset 4r).  2 NTI threads: NTI (128 or 256) per image, thread l
// owns the chunk of L consecutive positions [l L, (l + 1) L).  One pass = forward box sum of width 2r + 1 (Appendix B.1 of
// all four passes are that, with zeros beyond the end):
//   * the window sum at the chunk's first position comes from the chunk totals T_c of the previous pass's
//     output (a = (2r + 1) / L whole chunks) plus the m = (2r + 1) - a L positions after them, read from LDS
//     (or a + 1 whole chunks minus the L - m positions too many, whichever is fewer);
//   * then the recurrence S_{i+1} = S_i + (x_{i+2r+1} - x_i) along the chunk: x_i from registers (the thread's
//     own outputs of the previous pass), x_{i+2r+1} from LDS, one LDS read and one LDS write per position;
//   * the condition is CHECKED per thread and pass: B = sum of the chunk totals its windows touch bounds every
//     partial sum (terms are non-negative); m = the smallest non-zero term among those chunks gives the quantum
//     q = ulp(m) = 2^(e_m - 23) (every larger float32 is a multiple of it, zeros are multiples of anything).
//     Exact if B < 2^53 q.  Negative, infinite or NaN terms fail the check by construction.
//   * if ANY thread of the workgroup fails, the pass is redone for that line by ONE thread walking the LDS line
//     in the reference's own order (s += lead; store; s -= prev) -- bit-exact by construction, ~50 us, rare:
//     it takes a dynamic range of > 2^28 between a window's total and its smallest term.
// Results are therefore bit-identical to the sequential kernels for every input; only the speed depends on the
// data.  Time per line is independent of r (LDS-resident line: no delay lines at all).
//
// After the fourth pass: w = W / d^4, o = O / d^4, bg = (w == 0) ? NaN : o / w (flagging.py:419, 506-513), MODE 1:
// out = |data - bg| (rejection loop, :563-566), MODE 2: bg and data - bg + the line's NaN mark (:576-578, :962).
// Lines are rows of the time stage's TF images and the outputs are rows too (written in place over the inputs by
// the launcher; a transpose takes them to the FT layout the rest of the iteration expects).
// Long chunks (NTI = 128) spend fewer instructions per line: the per-thread cost of gathering a window's chunk
// totals and minima falls with a = (2r + 1) / L, the recurrence costs the same per position either way.
// grid (C lines, W windows), block 2 NTI, dynamic LDS boxx_lds_bytes(NTI, L, r).
// Host: n % 4 == 0, n + 4r <= NTI L, L <= 2r + 1 (hence r >= 8), (2r + 1) / L <= BOXX_AMAX, 16-byte aligned rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define BOXX_AMAX 96
__host__ __device__ constexpr int boxx_tn(int NTI) { return NTI + BOXX_AMAX + 2; }
__host__ __device__ constexpr int boxx_pb(int NTI, int L, int r) { return (NTI * L + 2 * r + 2 + 3) / 4 * 4; }
__host__ __device__ constexpr size_t boxx_lds_bytes(int NTI, int L, int r) {
    return (size_t)2 * boxx_pb(NTI, L, r) * 4 + (size_t)2 * boxx_tn(NTI) * 8 + (size_t)2 * boxx_tn(NTI) * 4;
}

// statistics (optional, tests / profiles): [0] line passes run, [1] line passes redone sequentially
#ifndef BOXX_MINWAVES
#define BOXX_MINWAVES 4                  // NTI = 256: 128 registers, two workgroups (16 waves) per compute unit
#endif
#ifndef BOXX_G
#define BOXX_G 8
#endif
#ifndef BOXX_MINWAVES128
#define BOXX_MINWAVES128 3               // NTI = 128: 168 registers, three workgroups (12 waves) per compute unit
#endif
template <int NTI, int L, int MODE, bool RECIP>
__global__ void __launch_bounds__(2 * NTI, NTI == 256 ? BOXX_MINWAVES : BOXX_MINWAVES128)
k_boxx(const float* srcW, unsigned img_gap, const float* __restrict__ data, const uint8_t* __restrict__ mask,
       float* outA, float* outB, int n, int ld, int r, BoxDenom denom, size_t sws_img,      // (outA / outB may alias the source rows: no __restrict__)
       size_t ws_data, size_t ws_mask, size_t ws_outA, size_t ws_outB, uint8_t* __restrict__ nanflag,
       unsigned long long* __restrict__ stats) {
    extern __shared__ float cf_ring[];
    const int tid = threadIdx.x;
    constexpr int TN = boxx_tn(NTI);
    const int img = tid / NTI, l = tid % NTI;
    const int line = blockIdx.x;
    const size_t win = blockIdx.y;
    const int C = gridDim.x;
    const int PB = boxx_pb(NTI, L, r);
    const int P = n + 4 * r;
    const int R2 = 2 * r;
    float* X = cf_ring + (size_t)img * PB;
    double* Ts = reinterpret_cast<double*>(cf_ring + 2 * (size_t)PB) + (size_t)img * TN;
    unsigned* Ms = reinterpret_cast<unsigned*>(reinterpret_cast<double*>(cf_ring + 2 * (size_t)PB) + 2 * (size_t)TN) + (size_t)img * TN;

    // ---- stage in: zero pads, the line at offset 4r (flagging.py:392-395) ----
    for (int k = l; k < PB; k += NTI) {
        if (k < 4 * r || k >= P) X[k] = 0.0f;
    }
    for (int k = NTI + l; k < TN; k += NTI) { Ts[k] = 0.0; Ms[k] = 0xFFFFFFFFu; }
    {
        const float* src = srcW + win * sws_img + (size_t)(img ? img_gap : 0u) + (size_t)line * ld;
        for (int q = l; q < n / 4; q += NTI) {
            const float4 v = reinterpret_cast<const float4*>(src)[q];
            float* px = X + 4 * r + 4 * q;                      // (4r + 4q: 16-byte aligned)
            *reinterpret_cast<float4*>(px) = v;
        }
    }
    __syncthreads();

    const int d = R2 + 1;
    const int a = d / L, m = d - a * L;                        // whole chunks in a window, length of the partial one
    const int i0 = l * L;
    // the thread's chunk of the current image state: as float64 on the short chunks (saves a conversion per position
    // and pass), as float32 on the long ones (half the registers)
    using ST = typename std::conditional<(L > 32), float, double>::type;
    ST od[L];
    bool bad;                                                  // a term that is negative, infinite or NaN
    // own chunk from LDS -> registers, totals, smallest non-zero term (as bits - 1: zero wraps to the maximum)
    auto rescan = [&]() {
        double T = 0.0;
        unsigned mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
        for (int k = 0; k < L; k++) {
            const float v = X[i0 + k];
            const unsigned b = __float_as_uint(v);
            mn = min(mn, b - 1u);
            mx = max(mx, b);
            od[k] = (ST)v;
            T += (double)v;
        }
        bad = mx >= 0x7F800000u;
        Ts[l] = T; Ms[l] = mn;
    };
    rescan();
    __syncthreads();

    unsigned npass_seq = 0;
#pragma unroll 1
    for (int pass = 0; pass < 4; pass++) {
        // ---- window sum at the chunk's first position, bound and quantum of everything this thread will add ----
        // (groups of independent LDS reads: a serial read-add loop would expose the LDS latency every time;
        //  entries past the window read a neutral slot: the end of the tables / the zero pad of the line)
        double acc = 0.0;
        unsigned mw = 0xFFFFFFFFu;
        for (int c0 = 0; c0 < a + 2; c0 += 8) {
            double tv[8];
            unsigned mv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + u;
                tv[u] = Ts[c < a ? l + c : TN - 1];
                mv[u] = Ms[c < a + 2 ? l + c : TN - 1];
            }
            acc += ((tv[0] + tv[1]) + (tv[2] + tv[3])) + ((tv[4] + tv[5]) + (tv[6] + tv[7]));
            mw = min(mw, min(min(min(mv[0], mv[1]), min(mv[2], mv[3])), min(min(mv[4], mv[5]), min(mv[6], mv[7]))));
        }
        {
            // the m positions after the a whole chunks -- or one more whole chunk less its last L - m positions
            const bool less = L - m < m;
            const int cnt = less ? L - m : m;
            const int e0 = i0 + a * L + (less ? m : 0);
            double part = 0.0;
            for (int j0 = 0; j0 < cnt; j0 += 4) {
                float xv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xv[u] = X[j0 + u < cnt ? e0 + j0 + u : PB - 1];
                part += ((double)xv[0] + (double)xv[1]) + ((double)xv[2] + (double)xv[3]);
            }
            acc = less ? (acc + Ts[l + a]) - part : acc + part;
        }
        const double bound = acc + Ts[l + a] + Ts[l + a + 1];
        // exact if bound < 2^53 * 2^(e_m - 23): with biased exponent fields EB (float64, 1023) and Em (float32, 127, at
        // least 1: subnormals share the quantum 2^-149): (EB - 1023) + 1 <= 30 + (Em - 127), one more bit of slack for
        // the rounding of `bound` itself.  EB = 2047 (infinite / NaN totals) can never pass.
        const int EB = (int)((__double2hiint(bound) >> 20) & 0x7FF);
        const int Em = max((int)((mw + 1u) >> 23), 1);
        // Only outputs that reach the result vote.  The cascade's last pass is used at positions [0, n) only, the pass
        // before at [0, n + 2r), ... (each pass looks 2r ahead); the reference walks a line in ascending order, so
        // whatever rounding its sums pick up beyond that point never reaches an output that counts -- and the tail of
        // the padded line is exactly where sums of ~d^3 large terms meet the single raw samples at the line's end.
        // Two chunks of margin per pass keep every total / minimum a voting thread reads inside the region the
        // previous pass's voters vouch for.
        const bool votes = i0 < n + (R2 + 2 * L) * (3 - pass);
        const bool fail = votes && (bad || (mw != 0xFFFFFFFFu && EB - Em > 924));

        // ---- the pass, fast form: recurrence along the chunk ----
        // (in groups of BOXX_G positions, each with its own LDS reads: the scheduler would otherwise convert every
        //  far sample to float64 up front and run out of registers on the long chunks)
        double T = 0.0;
        unsigned mn = 0xFFFFFFFFu;
#pragma unroll
        for (int g0 = 0; g0 < L; g0 += BOXX_G) {
            float far[BOXX_G];
#pragma unroll
            for (int u = 0; u < BOXX_G; u++)
                if (g0 + u < L - 1) far[u] = X[i0 + g0 + u + 1 + R2];
#pragma unroll
            for (int u = 0; u < BOXX_G; u++) {
                const int k = g0 + u;
                if (k < L) {
                    const float ov = (float)acc;               // flagging.py:410: the pass stores float32
                    if (k < L - 1) acc += (double)far[u] - (double)od[k];   // (exact: any order)
                    od[k] = (ST)ov;                            // (a failed pass reloads od[] from the line: rescan())
                    T += (double)ov;
                    mn = min(mn, __float_as_uint(ov) - 1u);
                }
            }
            if (L > 32) __builtin_amdgcn_sched_barrier(0);
        }
        // every far read of the workgroup is done before anybody overwrites the line; the same barrier tells
        // whether some thread's exactness condition failed
        const int anyfail = __syncthreads_or(fail ? 1 : 0);
        if (!anyfail) {
#pragma unroll
            for (int k = 0; k < L; k++) X[i0 + k] = (float)od[k];   // (exact: od[k] is a float32 value)
            Ts[l] = T; Ms[l] = mn;
            bad = false;                                       // (finite, non-negative: float32 of an exact sum of such terms)
        } else {
            // the reference's own order, one thread per image, in place (flagging.py:398-416 with zeros added
            // where the reference skips them: s + 0.0 == s)
            // (in batches of 16 positions, loads first and stores last: a load-add-store loop would pay the LDS
            //  latency at every position -- 0.6 ms per pass; a batch never stores what it still has to load: 2r >= 16)
            if (l == 0) {
                constexpr int SB = 16;
                double s = 0.0;
                const int pro = R2 < P ? R2 : P;               // prologue: the first 2r positions enter the sum
                int k = 0;
                for (; k + SB <= pro; k += SB) {
                    float v[SB];
#pragma unroll
                    for (int u = 0; u < SB; u++) v[u] = X[k + u];
#pragma unroll
                    for (int u = 0; u < SB; u++) s += (double)v[u];
                }
                for (; k < pro; k++) s += (double)X[k];
                const int lead_end = P - R2 > 0 ? P - R2 : 0;  // positions that still have a sample 2r ahead
                int i = 0;
                for (; i + SB <= lead_end; i += SB) {
                    float lead[SB], prev[SB], o[SB];
#pragma unroll
                    for (int u = 0; u < SB; u++) { lead[u] = X[i + R2 + u]; prev[u] = X[i + u]; }
#pragma unroll
                    for (int u = 0; u < SB; u++) {
                        s += (double)lead[u];
                        o[u] = (float)s;
                        s -= (double)prev[u];
                    }
#pragma unroll
                    for (int u = 0; u < SB; u++) X[i + u] = o[u];
                }
                for (; i < lead_end; i++) {
                    s += (double)X[i + R2];
                    const float prev = X[i];
                    X[i] = (float)s;
                    s -= (double)prev;
                }
                for (; i + SB <= P; i += SB) {                 // the tail: nothing left to add (flagging.py:412-416)
                    float prev[SB], o[SB];
#pragma unroll
                    for (int u = 0; u < SB; u++) prev[u] = X[i + u];
#pragma unroll
                    for (int u = 0; u < SB; u++) { o[u] = (float)s; s -= (double)prev[u]; }
#pragma unroll
                    for (int u = 0; u < SB; u++) X[i + u] = o[u];
                }
                for (; i < P; i++) {
                    const float prev = X[i];
                    X[i] = (float)s;
                    s -= (double)prev;
                }
                npass_seq++;
            }
            __syncthreads();
            rescan();
        }
        __syncthreads();
    }
    if (stats && l == 0) {
        atomicAdd(&stats[0], 4ull);
        if (npass_seq) atomicAdd(&stats[1], (unsigned long long)npass_seq);
    }

    // ---- finish: positions [0, n) of both images hold the cascade's sums (four passes shift left by 4r) ----
    const float* Xw = cf_ring;
    const float* Xo = cf_ring + PB;
    const float* drow = data + win * ws_data + (size_t)line * n;
    const uint8_t* mrow = mask ? mask + win * ws_mask + (size_t)line * n : nullptr;
    float* arow = outA + win * ws_outA + (size_t)line * ld;
    float* brow = MODE == 2 ? outB + win * ws_outB + (size_t)line * ld : nullptr;
    bool line_nan = false;
    for (int q = tid; q < n / 4; q += 2 * NTI) {
        const float4 w4 = *reinterpret_cast<const float4*>(Xw + 4 * q);
        const float4 o4 = *reinterpret_cast<const float4*>(Xo + 4 * q);
        float4 d4 = reinterpret_cast<const float4*>(drow)[q];
        if (mrow) {
            const uchar4 f4 = reinterpret_cast<const uchar4*>(mrow)[q];
            d4.x = f4.x ? 0.0f : d4.x; d4.y = f4.y ? 0.0f : d4.y; d4.z = f4.z ? 0.0f : d4.z; d4.w = f4.w ? 0.0f : d4.w;
        }
        const float wa[4] = {w4.x, w4.y, w4.z, w4.w}, oa[4] = {o4.x, o4.y, o4.z, o4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
        float wq[4], oq[4], ra[4], rb[4];
        if (RECIP) {
            unsigned long long okm = ~0ull;
#pragma unroll
            for (int k = 0; k < 4; k++) { wq[k] = box_divide(wa[k], denom, okm); oq[k] = box_divide(oa[k], denom, okm); }
            if (okm != ~0ull) {
                TRI_KEEP_BRANCH();
#pragma unroll
                for (int k = 0; k < 4; k++) { wq[k] = box_divide_ieee(wa[k], denom); oq[k] = box_divide_ieee(oa[k], denom); }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { wq[k] = box_divide_ieee(wa[k], denom); oq[k] = box_divide_ieee(oa[k], denom); }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float bg = (wq[k] == 0.0f) ? NAN : oq[k] / wq[k];
            if (MODE == 1) {
                ra[k] = fabsf(da[k] - bg);
            } else {
                ra[k] = bg;
                rb[k] = da[k] - bg;
                line_nan |= isnan(bg);
            }
        }
        reinterpret_cast<float4*>(arow)[q] = make_float4(ra[0], ra[1], ra[2], ra[3]);
        if (MODE == 2) reinterpret_cast<float4*>(brow)[q] = make_float4(rb[0], rb[1], rb[2], rb[3]);
    }
    if (MODE == 2 && line_nan) nanflag[win * (size_t)C + line] = 1;
}
