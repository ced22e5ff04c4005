// Box-Gaussian cascade, stage-pipelined across the four waves of a workgroup.
//
// The register-ring kernels (kernels_boxline.hpp) give every wave all four
// running sums of its 64 lines; with one wave per SIMD that is ~40 dependent-ish
// instructions per position and the rings fill the whole register file.  Here a
// workgroup of four waves (one per SIMD of a CU) shares 64 lines: wave s runs
// ONLY stage s + 1, for every position, and hands its output to the next wave
// through LDS.  Each stage's output stream lives in a circular LDS buffer that
// is at once the FIFO to the next stage and that stage's delay line (the sample
// that entered 2r positions ago is read back from the same buffer), so a
// position costs a wave two LDS reads, one LDS write and the five arithmetic
// instructions of one running sum:
//     s += in[t];  out[t] = (V)s;  s -= in[t - 2r]       (float64 / int32 sums)
// The arithmetic of every stage is the sequential recurrence of K4b, in the same
// order -- results are bit-identical (flagging.py:389-419 per line).
//
// Schedule: positions are processed in blocks of B.  In iteration j every thread
// helps staging block j from global memory into buffer 0 (prefetched P blocks
// ahead in registers), wave s runs its stage over block j - s - 1, and every
// thread helps storing block j - 5 of the last stage's output; one workgroup
// barrier per iteration.  A circular buffer holds Lc >= 2r + 2B positions (a
// multiple of B) plus a mirror of its first B positions after the end, so that
// neither the B new inputs nor the B trailing ones of a block ever wrap: all
// LDS addresses inside a block are static offsets from two registers.
// Requires 2r >= B (a block never reads a trailing sample written in the block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// LDS bytes of one workgroup: four stage buffers + the double-buffered output block
__host__ __device__ constexpr int boxp_lc(int r, int B) { return (2 * r + 2 * B + B - 1) / B * B; }
__host__ __device__ constexpr size_t boxp_lds_bytes(int r, int B) {
    return ((size_t)4 * (boxp_lc(r, B) + B) * 64 + (size_t)2 * B * 64) * 4;
}

// IMG 0: weight image (0/1 input, int32 sums); IMG 1: data image (flagged samples zeroed, float64 sums).
// Byte flags [n][C], data [n][C], output [n][C] (one "window": the spectrum path).
//
// Global memory goes through buffer descriptors and every iteration issues the SAME vector-memory
// instructions whatever the position (out-of-range accesses get an offset beyond the descriptor: loads
// return 0, stores are dropped).  With no branch around them the compiler can count exactly how many
// operations were issued after a prefetch and waits for that one only -- a branch would make it drain
// the whole queue, i.e. expose the full memory latency in every iteration.
// Host: C % (B / 4) == 0, n * C * 4 < 2^31, 2r >= B.
template <int IMG, int B, int P>
__device__ __forceinline__ void boxp_body(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
                                          float* __restrict__ dst, const int n, const int C, const int r, const float denom) {
    extern __shared__ float cf_ring[];
    using V = typename std::conditional<IMG == 0, int, float>::type;
    using A = typename std::conditional<IMG == 0, int, double>::type;
    constexpr int K = B / 4;                                   // consecutive columns per thread when staging / storing
    static_assert(K == 2 || K == 4, "block length 8 or 16");
    constexpr unsigned OOB = 0x7ffffff0u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R2 = 2 * r;
    const int Lc = boxp_lc(r, B), LcP = Lc + B;
    V* bufs = reinterpret_cast<V*>(cf_ring);
    float* outb = cf_ring + (size_t)4 * LcP * 64;
    for (int k = tid; k < 4 * LcP * 64; k += 256) bufs[k] = 0;
    const int NB = (n + 4 * r + B - 1) / B;                    // every stage runs over t in [0, n + 4r)

    // staging / storing: thread -> row srow of a block, columns scol .. scol + K - 1
    const int srow = tid / (64 / K);
    const int scol = (tid % (64 / K)) * K;
    const int gcol = blockIdx.x * 64 + scol;
    const bool colok = gcol < C;                               // (C % K == 0: all K columns or none)
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)srcFlags, 0, (int)((unsigned)n * (unsigned)C), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)((unsigned)n * (unsigned)C * 4u), 0x00020000);

    unsigned pref[P];                                          // K flag bytes
    float prex[IMG == 1 ? P : 1][K];
    auto issue = [&](int blk, int q) {
        const int t = blk * B + srow;
        const unsigned e = (colok && t < n) ? (unsigned)t * (unsigned)C + (unsigned)gcol : OOB;   // element index
        if (K == 4) pref[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, (int)e, 0, 0);
        else pref[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(frs, (int)e, 0, 0);
        if (IMG == 1) {
            // plain vector loads from a clamped address (hipcc 7.2 narrows __builtin_amdgcn_raw_buffer_load_b64 /
            // _b128 results that are bit-cast to float to a ONE-dword load)
            const float* a = srcData + ((colok && t < n) ? (size_t)e : (size_t)0);
            if (K == 4) {
                const float4 v = *reinterpret_cast<const float4*>(a);
                prex[q][0] = v.x; prex[q][1] = v.y; prex[q][K - 2] = v.z; prex[q][K - 1] = v.w;
            } else {
                const float2 v = *reinterpret_cast<const float2*>(a);
                prex[q][0] = v.x; prex[q][1] = v.y;
            }
        }
    };
    auto staged = [&](int blk, int q, int k) -> V {
        const int t = blk * B + srow;
        const bool fl = t >= n || !colok || ((pref[q] >> (8 * k)) & 0xFFu) != 0;    // beyond the line end: flagged
        if (IMG == 0) return fl ? (V)0 : (V)1;
        return fl ? (V)0 : (V)prex[IMG == 1 ? q : 0][k];
    };
#pragma unroll
    for (int q = 0; q < P; q++) issue(q, q);

    int lpin = 0;                                              // staging position in buffer 0
    int pin = 0, pold = Lc - R2;                               // this wave's block position / trailing position
    A acc = 0;
    V* bin = bufs + (size_t)wave * LcP * 64 + lane;
    V* bout = bufs + (size_t)(wave + 1) * LcP * 64 + lane;     // (wave 3 writes outb instead)
    // a stage's output is masked where the NEXT stage does not take it (see boxline_step):
    // stage 2 takes out_1[t] for t < n + 2r, stage 4 takes out_3[t] for t >= 2r
    const int keep_lo = wave == 2 ? R2 : 0;
    const int keep_hi = wave == 0 ? n + R2 : 0x7fffffff;
    __syncthreads();

    // NB + 5 iterations, rounded up to whole trips of P (the extra ones find nothing to do)
    for (int j0 = 0; j0 < NB + 5; j0 += P) {
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int j = j0 + q;
            if (j < NB) {                                      // stage block j (LDS only)
                V* p = bufs + ((size_t)lpin + srow) * 64 + scol;
                V sv[K];
#pragma unroll
                for (int k = 0; k < K; k++) sv[k] = staged(j, q, k);
#pragma unroll
                for (int k = 0; k < K; k++) p[k] = sv[k];
                if (lpin == 0) {
#pragma unroll
                    for (int k = 0; k < K; k++) p[(size_t)Lc * 64 + k] = sv[k];
                }
                lpin = (lpin + B == Lc) ? 0 : lpin + B;
            }
            issue(j + P, q);                                   // fetch block j + P (always issued)
            const int b = j - wave - 1;
            if (b >= 0 && b < NB) {
                const V* pi = bin + (size_t)pin * 64;
                const V* po = bin + (size_t)pold * 64;
                V xin[B], xold[B];
#pragma unroll
                for (int u = 0; u < B; u++) { xin[u] = pi[u * 64]; xold[u] = po[u * 64]; }
                const int t0 = b * B;
                if (wave == 3) {
                    float* ob = outb + (size_t)(b & 1) * B * 64 + lane;
#pragma unroll
                    for (int u = 0; u < B; u++) {
                        acc += (A)xin[u];
                        ob[u * 64] = (float)acc;
                        acc -= (A)xold[u];
                    }
                } else {
                    V* po2 = bout + (size_t)pin * 64;
                    const bool whole = t0 >= keep_lo && t0 + B <= keep_hi;
                    V o[B];
#pragma unroll
                    for (int u = 0; u < B; u++) {
                        acc += (A)xin[u];
                        o[u] = (V)acc;
                        acc -= (A)xold[u];
                    }
                    if (whole) {                                // interior blocks: no masks (both arms store: a real branch)
#pragma unroll
                        for (int u = 0; u < B; u++) po2[u * 64] = o[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < B; u++) {
                            o[u] = (t0 + u >= keep_lo && t0 + u < keep_hi) ? o[u] : (V)0;
                            po2[u * 64] = o[u];
                        }
                    }
                    if (pin == 0) {
#pragma unroll
                        for (int u = 0; u < B; u++) po2[(Lc + u) * 64] = o[u];
                    }
                }
                pin = (pin + B == Lc) ? 0 : pin + B;
                pold = (pold + B >= Lc) ? pold + B - Lc : pold + B;
            }
            {                                                  // store block j - 5 (always issued)
                const int bs = j - 5;
                const int i = bs * B + srow - 4 * r;
                const bool ok = bs >= 0 && bs < NB && i >= 0 && i < n && colok;
                const float* ob = outb + ((size_t)(bs & 1) * B + srow) * 64 + scol;
                const unsigned eb = ok ? ((unsigned)i * (unsigned)C + (unsigned)gcol) * 4u : OOB;
                if (K == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(ob);
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 y;
                    y[0] = __builtin_bit_cast(unsigned, v.x / denom); y[1] = __builtin_bit_cast(unsigned, v.y / denom);
                    y[2] = __builtin_bit_cast(unsigned, v.z / denom); y[3] = __builtin_bit_cast(unsigned, v.w / denom);
                    __builtin_amdgcn_raw_buffer_store_b128(y, ors, (int)eb, 0, 0);
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(ob);
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    u2 y;
                    y[0] = __builtin_bit_cast(unsigned, v.x / denom); y[1] = __builtin_bit_cast(unsigned, v.y / denom);
                    __builtin_amdgcn_raw_buffer_store_b64(y, ors, (int)eb, 0, 0);
                }
            }
            __syncthreads();
        }
    }
}

// Spectrum path: grid (ceil(C / 64), 2 images), block 256, dynamic LDS boxp_lds_bytes(r, B)
template <int B, int P>
__global__ void __launch_bounds__(256, 1)
k_boxp_spec(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
            float* __restrict__ dstW, float* __restrict__ dstO, int n, int C, int r, float denom) {
    if (blockIdx.y == 0) boxp_body<0, B, P>(srcData, srcFlags, dstW, n, C, r, denom);
    else boxp_body<1, B, P>(srcData, srcFlags, dstO, n, C, r, denom);
}

// ---------------------------------------------------------------------------
// K4q  The stage pipeline for the 2-D time-axis stage, delay lines in REGISTERS.
//
// With one stage per wave a delay line is 2r values per lane -- it fits the register file where the
// four delay lines of K4r (kernels_boxline.hpp) do not.  The newest KS = 16 * floor(2r / 16) positions of
// delay are registers addressed statically (the iteration loop is unrolled over KS / 16 blocks;
// "R[slot] = in" is a register rename, not a move); the remaining d = 2r - KS < 16 positions come for
// free from the inter-stage FIFO: a stage reads its input block twice from LDS, once as it is (the sample
// entering the sum) and once d positions back (the sample entering the register delay line), so what
// leaves the registers entered 2r positions ago.  The FIFOs hold three blocks (+ a mirror of the first,
// so that no 16-position read wraps); there is no LDS ring.
// Per position and wave: 2 LDS reads, 5 arithmetic instructions, 1 LDS write, and two workgroups fit a
// compute unit, so the barrier per block and the dependent float64 adds of one wave overlap with the
// other workgroup's work.
//
// Same schedule as K4p: iteration j stages block j (data rows + TF4 flag words, fetched P blocks ahead),
// wave s runs stage s + 1 over block j - s - 1, everybody stores block j - 5.  Thread (wave w, lane l)
// stages / stores positions 4w .. 4w + 3 of column l: one flag word covers them.
// grid (ceil(C / 64), W), block 256; host: n % 4 == 0, window below 2^31 bytes, 16 <= 2r, 2r - KS < 16.
//
// Round 3: the block length B is a template parameter.  What bounds these kernels is OCCUPANCY (one barrier per block,
// dependent float64 chains: only other resident waves fill the gaps), so wherever the delay line leaves room the kernel is
// compiled for 128 registers and runs blocks of EIGHT positions -- 36 KB of FIFOs, four workgroups = four waves per SIMD,
// KS = 8 * floor(2r / 8) <= 80, 2r - KS < 8, a thread stages / stores two positions (half a TF4 flag word): 15.3-17.3 ms per
// 1008-window launch pair against 19.3-20.8 with blocks of 16.  k_boxq_deep (KS = 80 / 96 with 8 <= 2r - KS < 16): blocks of 8
// with FIFOs of FOUR blocks + mirror and 168 registers, three waves per SIMD.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr size_t boxq_lds_bytes(int B, int NF = 3) { return (size_t)(4 * (NF + 1) * B * 64 + 2 * B * 64) * 4; }
#define BOXQ_LDS_BYTES boxq_lds_bytes(16)   // 72 KB: two workgroups per compute unit (B = 8: 36 KB, four)
__host__ __device__ constexpr int boxq_prefetch(int ks) { return ks == 48 || ks == 96 ? 6 : (ks == 80 ? 5 : 4); }
#ifndef BOXQ_ABLATE
#define BOXQ_ABLATE 0                    // timing-only builds: 1 = no finishing, 2 = no stage arithmetic, 4 = no staging / loads
#endif
__host__ __device__ constexpr int boxq_lcm(int a, int b) {
    int x = a, y = b;
    while (y) { int t = x % y; x = y; y = t; }
    return a / x * b;
}

// blocks of 8 positions: prefetch depth for KS / 8 register blocks (P divides or is divided by it: short unroll; 128 registers)
__host__ __device__ constexpr int boxq_prefetch8(int nblk) {
    return nblk == 6 ? 3 : (nblk == 8 ? 4 : (nblk == 9 ? 3 : (nblk == 10 ? 2 : (nblk == 12 ? 4 : nblk))));
}

// NF: blocks a stage FIFO holds (+ a mirror of the first).  3 serves a FIFO delay d < B; 4 serves d < 2 B (the block
// being written, the one being read and the two before it): blocks of 8 positions for 2r - KS < 16.
template <int IMG, int KS, int P, int B = 16, int NF = 3>
__device__ __forceinline__ void boxq_body(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags,
                                          float* __restrict__ dst, const int n, const int C, const int r, const BoxDenom denom,
                                          const size_t sws, const size_t dws, const size_t win) {
    using V = typename std::conditional<IMG == 0, int, float>::type;
    using A = typename std::conditional<IMG == 0, int, double>::type;
    constexpr int PPT = B / 4;                                 // positions per thread when staging / storing (4 or 2)
    constexpr int NBLK = KS / B;
    constexpr int U = boxq_lcm(NBLK, P);
    constexpr unsigned OOB = 0x7ffffff0u;
    static_assert(KS % B == 0 && KS >= B, "register part: whole blocks");
    static_assert(B == 8 || B == 16, "block length");
    extern __shared__ float cf_ring[];                         // dynamic LDS: BOXQ_LDS_BYTES
    typedef V FifoT[(NF + 1) * B][64];                         // one stage's input stream: NF blocks + mirror of the first
    typedef float OutT[B][64];
    FifoT* fifo = reinterpret_cast<FifoT*>(cf_ring);           // [4]
    OutT* outb = reinterpret_cast<OutT*>(cf_ring + 4 * (NF + 1) * B * 64);   // [2] output blocks of stage 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R2 = 2 * r;
    const int d = R2 - KS;                                     // host: 0 <= d < B (NF = 3), < 2 B (NF = 4)
    const int c = blockIdx.x * 64 + lane;
    const bool colok = c < C;
    const int NB = (n + 4 * r + B - 1) / B;                    // every stage runs over t in [0, n + 4r)
    const unsigned rowb = (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)(srcFlags + win * sws), 0, (int)((unsigned)(n / 4) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(srcData + win * sws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const unsigned coff = (unsigned)c * 4u;
    for (int k = tid; k < 4 * (NF + 1) * B * 64; k += 256) reinterpret_cast<V*>(cf_ring)[k] = 0;

    unsigned pref[P];                                          // TF4 word: flags of positions 4w .. 4w + 3
    float prex[IMG == 1 ? P : 1][PPT];
    auto issue = [&](int blk, int q) {
        const int t = blk * B + PPT * wave;                    // multiple of 4 (2); n % 4 == 0: all its positions in or out
        const bool ok = colok && t < n;
        pref[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, (int)(ok ? (unsigned)(t >> 2) * rowb + coff : OOB), 0, 0);
        if (IMG == 1) {
#pragma unroll
            for (int k = 0; k < PPT; k++)
                prex[q][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, (int)(ok ? (unsigned)(t + k) * rowb + coff : OOB), 0, 0));
        }
    };
#pragma unroll
    for (int q = 0; q < P; q++) issue(q, q);

    V R[KS];
#pragma unroll
    for (int k = 0; k < KS; k++) R[k] = 0;
    A acc = 0;
    int lslot = 0;                                             // FIFO block slot the staging writes next
    int slot = 0, s0 = NF * B - d;                             // this wave's input block slot / first row of the delayed window
    if (s0 >= NF * B) s0 -= NF * B;
    // a stage's output is masked where the NEXT stage does not take it (see boxline_step):
    // stage 2 takes out_1[t] for t < n + 2r, stage 4 takes out_3[t] for t >= 2r
    const int keep_lo = wave == 2 ? R2 : 0;
    const int keep_hi = wave == 0 ? n + R2 : 0x7fffffff;
    __syncthreads();

    for (int j0 = 0; j0 < NB + 5; j0 += U) {
#pragma unroll
        for (int qq = 0; qq < U; qq++) {
            const int j = j0 + qq;
            const int q = qq % P;                              // prefetch slot (static)
            const int sb = (qq % NBLK) * B;                    // register delay-line slots of this iteration (static)
#if !(BOXQ_ABLATE & 4)
            {                                                  // stage block j into the stage-1 FIFO (LDS only)
                const int t = j * B + PPT * wave;
                const bool tin = colok && t < n;               // beyond the line end: flagged
                const unsigned fw = PPT == 4 ? pref[q] : pref[q] >> (16 * (wave & 1));   // (B = 8: the word's upper half for odd waves)
                V sv[PPT];
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    const bool fl = !tin || ((fw >> (8 * k)) & 0xFFu) != 0;
                    if (IMG == 0) sv[k] = fl ? (V)0 : (V)1;
                    else sv[k] = fl ? (V)0 : (V)prex[IMG == 1 ? q : 0][k];
                }
                V* p = &fifo[0][lslot * B + PPT * wave][lane];
#pragma unroll
                for (int k = 0; k < PPT; k++) p[k * 64] = sv[k];
                if (lslot == 0) {
#pragma unroll
                    for (int k = 0; k < PPT; k++) p[(NF * B + k) * 64] = sv[k];
                }
                lslot = lslot == NF - 1 ? 0 : lslot + 1;
            }
            issue(j + P, q);                                   // fetch block j + P (always issued)
#endif
            const int b = j - wave - 1;
#if !(BOXQ_ABLATE & 2)
            if (b >= 0 && b < NB) {
                const int t0 = b * B;
                const V* pi = &fifo[wave][slot * B][lane];
                const V* pd = &fifo[wave][s0][lane];
                V xin[B], xdel[B], o[B];
#pragma unroll
                for (int u = 0; u < B; u++) { xin[u] = pi[u * 64]; xdel[u] = pd[u * 64]; }
#pragma unroll
                for (int u = 0; u < B; u++) {
                    const V old = R[sb + u];                   // entered the registers KS positions ago, the sum 2r positions ago
                    R[sb + u] = xdel[u];
                    acc += (A)xin[u];
                    o[u] = (V)acc;
                    acc -= (A)old;
                }
                if (wave == 3) {
                    float* ob = &outb[b & 1][0][lane];
#pragma unroll
                    for (int u = 0; u < B; u++) ob[u * 64] = (float)o[u];
                } else {
                    const bool whole = t0 >= keep_lo && t0 + B <= keep_hi;
                    V* po = &fifo[wave + 1][slot * B][lane];
                    if (whole) {                                // interior blocks: no masks (both arms store: a real branch)
#pragma unroll
                        for (int u = 0; u < B; u++) po[u * 64] = o[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < B; u++) {
                            o[u] = (t0 + u >= keep_lo && t0 + u < keep_hi) ? o[u] : (V)0;
                            po[u * 64] = o[u];
                        }
                    }
                    if (slot == 0) {
#pragma unroll
                        for (int u = 0; u < B; u++) po[(NF * B + u) * 64] = o[u];
                    }
                }
                slot = slot == NF - 1 ? 0 : slot + 1;
                s0 = s0 + B >= NF * B ? s0 + B - NF * B : s0 + B;
            }
#endif
#if !(BOXQ_ABLATE & 1)
            {                                                  // store block j - 5 (always issued)
                const int bs = j - 5;
                const int i0 = bs * B + PPT * wave - 4 * r;
                const float* ob = &outb[bs & 1][PPT * wave][lane];
                // division by the launch constant through its reciprocal (kernels_boxline.hpp: exact wherever the
                // class test passes, checked for all 2^32 inputs per radius; anything else is redone by IEEE division)
                float a[PPT], y[PPT];
                unsigned long long okm = ~0ull;
#pragma unroll
                for (int k = 0; k < PPT; k++) { a[k] = ob[k * 64]; y[k] = box_divide(a[k], denom, okm); }
                if (okm != ~0ull) {
                    TRI_KEEP_BRANCH();
#pragma unroll
                    for (int k = 0; k < PPT; k++) y[k] = box_divide_ieee(a[k], denom);
                }
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    const int i = i0 + k;
                    const bool ok = bs >= 0 && bs < NB && i >= 0 && i < n && colok;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[k]), ors, (int)(ok ? (unsigned)i * rowb + coff : OOB), 0, 0);
                }
            }
#endif
            __syncthreads();
        }
    }
}

// (HIP: the second __launch_bounds__ argument counts WAVES PER SIMD.)  B = 8: 36 KB of LDS and at most 128 registers:
// four workgroups = four waves per SIMD, for delay lines of up to 80 registers; host: 2r - KS < 8.
template <int KS, int IMG, int B = 16>
__global__ void __launch_bounds__(256, B == 8 ? 4 : 2)
k_boxq(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags, float* __restrict__ dstImg,
       int n, int C, int r, BoxDenom denom, size_t sws, size_t dws) {
    boxq_body<IMG, KS, B == 8 ? boxq_prefetch8(KS / 8) : boxq_prefetch(KS), B>(srcData, srcFlags, dstImg, n, C, r, denom, sws, dws, blockIdx.y);
}
// Delay lines of 96 registers (r = 52 ... 55, default.yaml's largest time-axis radius 54): blocks of 8 with FIFOs one block
// deeper (8 <= 2r - KS < 16; 44 KB of LDS) and 168 registers: THREE workgroups = three waves per SIMD instead of the
// two that blocks of 16 allow.  host: 8 <= 2r - KS < 16
template <int KS, int IMG>
__global__ void __launch_bounds__(256, 3)
k_boxq_deep(const float* __restrict__ srcData, const uint8_t* __restrict__ srcFlags, float* __restrict__ dstImg,
            int n, int C, int r, BoxDenom denom, size_t sws, size_t dws) {
    boxq_body<IMG, KS, boxq_prefetch8(KS / 8), 8, 4>(srcData, srcFlags, dstImg, n, C, r, denom, sws, dws, blockIdx.y);
}

// ---------------------------------------------------------------------------
// K4qf  The same pipeline for the frequency-axis stage, fused with the masked division (what k_boxf
// does with register + LDS rings at one wave per SIMD).  One workgroup of EIGHT waves filters both
// images of 64 lines: waves 0-3 are the four stages of the weight image, waves 4-7 those of the
// weight * data image; every block of 16 positions both fourth stages leave their sums in LDS and all
// 512 threads finish two (position, line) pairs each:
//     w = W / d^4, o = O / d^4, bg = (w == 0) ? NaN : o / w        (flagging.py:419, 506-513)
//     MODE 1: dstO = |data - bg|                                    (rejection loop, :563-566)
//     MODE 2: dstO = bg, dstW = data - bg, nanflag[line] = 1 on a NaN   (:576-578, :962)
// Input images stored transposed (the time-axis stage's TF output: line c is row c of a [C][ld] array,
// the data image img_gap elements after the weight image): a block is 64 row segments of 16 floats per
// image, one float4 per thread, transposed on its way into the stage-1 FIFO.  Outputs and `data` are
// position-major [n][C] (coalesced along the lines).  The filtered images are never written.
// grid (ceil(C / 64), W), block 512, dynamic LDS BOXQF_LDS_BYTES (one workgroup = two waves per SIMD).
// Host: n % 4 == 0, ld % 4 == 0, img_gap % 4 == 0, 16-byte aligned images, windows below 2^31 bytes.
// ---------------------------------------------------------------------------
// LDS bytes for blocks of B positions: 144 KB at B = 16 (one workgroup per CU), 72 KB at B = 8 (two)
// Row pitch of the LDS tiles in floats.  The stage arithmetic reads / writes whole rows (lane = line: any pitch is
// conflict-free); the transposing stage-in writes 4 (2) consecutive ROWS per thread with 4 threads per line, so with
// a pitch of 64 the four threads of a line hit one bank.  A pitch = 2 (mod 8) spreads them: bank = 8 q + line (mod 32).
#ifndef BOXQF_LW
#define BOXQF_LW 64
#endif
// Round 4 -- WHOLE 64-BYTE ROW PIECES with blocks of 8 (BOXQF_DBL): the stage-in used to take 32 bytes of every line per block
// and found the other half of the 64-byte sector evicted more often than not (PMC: 13.3 bytes read per sample for 12).  Now every
// SECOND iteration stages a double block -- one float4 per thread, 64 bytes per line -- into stage-1 FIFOs that hold one block
// more (four + mirror: 4 KB per workgroup, 76 KB in all: still two workgroups per compute unit).  Even register-block counts only
// (KS = 16, 32, 48, 64, 80: every radius of the shipped strategies); the others keep the single-block staging.
#ifndef BOXQF_DBL
#define BOXQF_DBL 1
#endif
#ifndef BOXQF_PAIR
#define BOXQF_PAIR 1                     // FIFO rows stored in pairs (8-byte LDS accesses) where the register budget allows
#endif
#ifndef BOXQF_ABLATE_BARRIER
#define BOXQF_ABLATE_BARRIER 0
#endif
__host__ __device__ constexpr bool boxqf_dbl(int KS, int B) { return BOXQF_DBL && B == 8 && (KS / B) % 2 == 0; }
// double blocks in flight per thread (2 x that many blocks ahead; 2 x it divides or is divided by the register-block count)
__host__ __device__ constexpr int boxqf_prefetch2(int nblk) { return nblk == 6 ? 3 : (nblk == 10 || nblk == 2 ? 1 : 2); }
// floats: stage-1 FIFOs [2 images][NF1 + 1 blocks], stage-2..4 FIFOs [2][3][3 + 1 blocks], output blocks [2][2]
__host__ __device__ constexpr size_t boxqf_lds_floats(int B, bool dbl) { return (size_t)(2 * ((dbl ? 5 : 4) + 3 * 4) * B * BOXQF_LW + 2 * 2 * B * BOXQF_LW); }
__host__ __device__ constexpr size_t boxqf_lds_bytes(int B, bool dbl = false) { return boxqf_lds_floats(B, dbl) * 4; }
#define BOXQF_LDS_BYTES boxqf_lds_bytes(16)
// B = 8: half the FIFO memory and, for delay lines of up to 80 registers, half the register budget: TWO workgroups
// (four waves per SIMD) share a compute unit -- twice the barriers per position against twice the latency hiding.
// Occupancy, not instruction count, is what this kernel's time follows: at two waves per SIMD (B = 16) dropping 22
// vector instructions per sample changed nothing (22.1 ms), at four waves (B = 8) the same cut took 19.0 -> 17.3 ms.
#ifndef BOXQF_P10
#define BOXQF_P10 2                      // KS = 80 at B = 8: two blocks ahead is what 128 registers leave (5: 13 dwords of scratch, 23 ms instead of 17.8)
#endif
__host__ __device__ constexpr int boxqf_prefetch8(int nblk) {
    return nblk == 6 ? 3 : (nblk == 8 ? 4 : (nblk == 9 ? 3 : (nblk == 10 ? BOXQF_P10 : (nblk == 12 ? 4 : nblk))));
}
#ifndef BOXQF_MINW8
#define BOXQF_MINW8 4                    // HIP: the second __launch_bounds__ argument counts WAVES PER SIMD -- two 8-wave workgroups per CU need 4
                                         // (with 2 the KS = 64 kernel took 130 registers: ONE workgroup per CU, 25 ms; bounded to 128 it runs in 17.7)
#endif
template <int KS, int MODE, int B = 16>
__global__ void __launch_bounds__(512, B == 8 ? BOXQF_MINW8 : 1)
k_boxqf(const float* __restrict__ srcW, unsigned img_gap, float* __restrict__ dstW, float* __restrict__ dstO,
        const float* __restrict__ data, int n, int C, int ld, int r, BoxDenom denom, size_t sws_img, size_t dws,
        size_t ws_data, uint8_t* __restrict__ nanflag) {
    constexpr int PP = B / 4;                                  // positions per thread when staging (4 or 2)
    constexpr int FN = B / 8;                                  // (position, line) pairs per thread when finishing (2 or 1)
    constexpr int NBLK = KS / B;
    constexpr bool DBL = boxqf_dbl(KS, B);                     // stage a double block every second iteration (whole 64-byte row pieces)
    constexpr int P2 = boxqf_prefetch2(NBLK);                  // ... double blocks in flight
    constexpr int P = DBL ? 2 * P2 : (B == 8 ? boxqf_prefetch8(NBLK) : boxq_prefetch(KS));   // (B = 8: KS = 32, 40, 48, 56, 64, 72, 80, 96 -> 4, 5, 3, 7, 4, 3, BOXQF_P10, 4 blocks ahead: 128 registers)
    static_assert(KS % B == 0, "register part: whole blocks");
    static_assert(B == 8 || B == 16, "block length");
    constexpr int U = boxq_lcm(NBLK, P);
    static_assert(!DBL || U % 2 == 0, "double-block staging: an even number of iterations per trip");
    constexpr unsigned OOB = 0x7ffffff0u;
    extern __shared__ float cf_ring[];
    constexpr int LW = BOXQF_LW;
    // one stage's input stream: NF blocks + a mirror of the first (no block read ever wraps); NF = 3, the stage-1 streams of the
    // double-block staging 4.  Rows of LW floats: [stage-1 of image 0][of image 1][stages 2..4 of image 0][of image 1][output blocks]
    constexpr int NF1 = DBL ? 4 : 3;
    constexpr int F1 = (NF1 + 1) * B * LW, FN3 = 4 * B * LW;
    typedef float OutT[B][LW];
    auto fifo_of = [&](int image, int stage) -> float* {
        return stage == 0 ? cf_ring + image * F1 : cf_ring + 2 * F1 + (image * 3 + stage - 1) * FN3;
    };
    OutT* outb = reinterpret_cast<OutT*>(cf_ring + 2 * F1 + 6 * FN3);   // [2 images][2 blocks]
    // Round 4 -- FIFO rows are stored in PAIRS: positions 2p and 2p + 1 of a line sit next to each other (element (row, line) at
    // ((row >> 1) * LW + line) * 2 + (row & 1)), so a stage takes / hands on two positions with one 8-byte LDS access
    // (ds_read_b64 runs at twice ds_read_b32's bytes per clock, ds_write_b64 at 4/3 of ds_write_b32's; MI355X_MICROARCH.md, LDS) and
    // the transposing stage-in writes 8 bytes per thread without a bank conflict (it was four-way).  Every row offset a
    // block starts from is even: blocks of 8 / 16, FIFO delay d = 2r - KS with 2r and KS even.
    // Only where the registers allow it (KS <= 40 at the 128-register bound of blocks of 8): the 8-byte accesses want aligned
    // register pairs, and with 48 ... 80 delay registers that spills 23 - 41 dwords (r = 25 ... 43: 17.4 -> 42 ... 53 ms);
    // at KS = 32 it buys 3 % (16.45 -> 15.99 ms per 1008-window launch).  The other instantiations keep one row per position.
    constexpr bool PAIR = BOXQF_PAIR && B == 8 && KS <= 40;
    auto ld2 = [&](const float* fifo, int row) -> float2 {               // rows row, row + 1 (row even) of this lane's line
        if (PAIR) return *reinterpret_cast<const float2*>(fifo + ((size_t)(row >> 1) * LW + (threadIdx.x & 63)) * 2);
        const float* p = fifo + (size_t)row * LW + (threadIdx.x & 63);
        return make_float2(p[0], p[LW]);
    };
    auto st2 = [&](float* fifo, int row, int line, float a, float b2) {
        if (PAIR) { *reinterpret_cast<float2*>(fifo + ((size_t)(row >> 1) * LW + line) * 2) = make_float2(a, b2); return; }
        float* p = fifo + (size_t)row * LW + line;
        p[0] = a;
        p[LW] = b2;
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = wave >> 2, st = wave & 3;                  // this wave's image and stage
    const size_t win = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int R2 = 2 * r;
    const int d = R2 - KS;                                     // host: 0 <= d < 16
    const int NB = (n + 4 * r + B - 1) / B;
    for (int k = tid; k < 2 * F1 + 6 * FN3; k += 512) cf_ring[k] = 0.0f;

    // staging: thread -> image (tid >> 8), line (tid & 255) / 4, positions PP (tid & 3) .. + PP - 1 of a block
    const int s_line = (tid & 255) >> 2, s_q = tid & 3;
    const bool s_lok = c0 + s_line < C;
    const float* s_base = srcW + win * sws_img + (size_t)(img ? img_gap : 0u) + (size_t)(s_lok ? c0 + s_line : 0) * ld;
    float pre[DBL ? P2 : P][DBL ? 4 : PP];
    auto issue = [&](int blk, int q) {
        if (DBL) {                                             // double block blk: positions 16 blk + 4 s_q .. + 3
            const int p = blk * 16 + 4 * s_q;
            const float4 v = *reinterpret_cast<const float4*>(s_base + (p < n ? p : 0));
            pre[q][0] = v.x; pre[q][1] = v.y; pre[q][DBL ? 2 : 0] = v.z; pre[q][DBL ? 3 : 0] = v.w;
            return;
        }
        const int p = blk * B + PP * s_q;                      // n % 4 == 0: the positions are in or out together
        if (PP == 4) {
            const float4 v = *reinterpret_cast<const float4*>(s_base + (p < n ? p : 0));
            pre[q][0] = v.x; pre[q][1] = v.y; pre[q][PP - 2] = v.z; pre[q][PP - 1] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2*>(s_base + (p < n ? p : 0));
            pre[q][0] = v.x; pre[q][1] = v.y;
        }
    };
    // finishing: thread -> line tid & 63, position tid >> 6 (and that + 8 with blocks of 16)
    const int f_u = tid >> 6;
    const int c = c0 + lane;
    const bool colok = c < C;
    const unsigned rowb = (unsigned)C * 4u;
    const unsigned coff = (unsigned)c * 4u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(data + win * ws_data), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(dstO + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dstW + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);
    float dpre[P][FN];
    auto issue_data = [&](int blk, int q) {                    // data samples of output block blk
#pragma unroll
        for (int h = 0; h < FN; h++) {
            const int i = blk * B + f_u + 8 * h - 4 * r;
            const bool ok = blk >= 0 && i >= 0 && i < n && colok;
            dpre[q][h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(ok ? (unsigned)i * rowb + coff : OOB), 0, 0));
        }
    };
#pragma unroll
    for (int q = 0; q < P; q++) { if (!DBL || q < P2) issue(q, q); issue_data(q - 5, q); }

    float R[KS];
#pragma unroll
    for (int k = 0; k < KS; k++) R[k] = 0.0f;
    double acc = 0.0;
    int lslot = 0;
    // Every wave runs its stage in EVERY iteration, also over the st + 1 blocks before its first one (all-zero
    // input: sums and delay line stay zero) and past its last (results never stored): no branch around the
    // arithmetic, so it can be scheduled together with the finishing work.  Block j - st - 1 sits in FIFO slot
    // (j - st - 1) mod 3; the delayed window starts d rows earlier.
    const int nf = st == 0 ? NF1 : 3;                          // blocks this wave's input FIFO holds
    int slot = (nf * 8 - st - 1) % nf, s0 = (nf * B * 8 - (st + 1) * B - d) % (nf * B);
    int oslot = (3 * 8 - st - 1) % 3;                          // the same block in the NEXT stage's FIFO (always three blocks)
    float* const fin = fifo_of(img, st);
    float* const fout = fifo_of(img, st < 3 ? st + 1 : 3);
    const int keep_lo = st == 2 ? R2 : 0;
    const int keep_hi = st == 0 ? n + R2 : 0x7fffffff;
    bool line_nan = false;
    __syncthreads();

    for (int j0 = 0; j0 < NB + 5; j0 += U) {
#pragma unroll
        for (int qq = 0; qq < U; qq++) {
            const int j = j0 + qq;
            const int q = qq % P;
            const int sb = (qq % NBLK) * B;
            if (DBL) {
                if (qq % 2 == 0) {                             // double block j / 2 (transposed) into this image's stage-1 FIFO: rows lslot B + 4 s_q ..
                    const int q2 = (qq / 2) % P2;
                    const int p = j * B + 4 * s_q;
                    const bool ok = s_lok && p < n;            // beyond the line end / the last line: zero input
                    float* pf = fifo_of(img, 0);
                    const int r0 = lslot * B + 4 * s_q;
                    const float v0 = ok ? pre[q2][0] : 0.0f, v1 = ok ? pre[q2][1] : 0.0f, v2 = ok ? pre[q2][DBL ? 2 : 0] : 0.0f, v3 = ok ? pre[q2][DBL ? 3 : 0] : 0.0f;
                    st2(pf, r0, s_line, v0, v1);
                    st2(pf, r0 + 2, s_line, v2, v3);
                    if (lslot == 0 && s_q < 2) {
                        st2(pf, NF1 * B + r0, s_line, v0, v1);
                        st2(pf, NF1 * B + r0 + 2, s_line, v2, v3);
                    }
                    lslot = lslot == 0 ? 2 : 0;
                    issue(j / 2 + P2, q2);                     // (always issued: clamped address)
                }
            } else {                                           // stage block j (transposed) into this image's stage-1 FIFO
                const int p = j * B + PP * s_q;
                const bool ok = s_lok && p < n;                // beyond the line end / the last line: zero input
                float* pf = fifo_of(img, 0);
                const int r0 = lslot * B + PP * s_q;
#pragma unroll
                for (int k = 0; k < PP; k += 2) {
                    const float va = ok ? pre[DBL ? 0 : q][k] : 0.0f, vb = ok ? pre[DBL ? 0 : q][k + 1] : 0.0f;
                    st2(pf, r0 + k, s_line, va, vb);
                    if (lslot == 0) st2(pf, 3 * B + r0 + k, s_line, va, vb);
                }
                lslot = lslot == 2 ? 0 : lslot + 1;
            }
            if (!DBL) issue(j + P, q);                         // (always issued: clamped address)
            const int b = j - st - 1;
            {
                const int t0 = b * B;
                float xin[B], xdel[B], o[B];
#pragma unroll
                for (int u = 0; u < B; u += 2) {
                    const float2 a = ld2(fin, slot * B + u), d2 = ld2(fin, s0 + u);
                    xin[u] = a.x; xin[u + 1] = a.y; xdel[u] = d2.x; xdel[u + 1] = d2.y;
                }
#pragma unroll
                for (int u = 0; u < B; u++) {
                    const float old = R[sb + u];
                    R[sb + u] = xdel[u];
                    acc += (double)xin[u];
                    o[u] = (float)acc;
                    acc -= (double)old;
                }
                if (st == 3) {
                    float* ob = &outb[img * 2 + (b & 1)][0][lane];
#pragma unroll
                    for (int u = 0; u < B; u++) ob[u * LW] = o[u];
                } else {
                    const bool whole = t0 >= keep_lo && t0 + B <= keep_hi && b >= 0;
                    if (!whole) {
#pragma unroll
                        for (int u = 0; u < B; u++) o[u] = (t0 + u >= keep_lo && t0 + u < keep_hi) ? o[u] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < B; u += 2) st2(fout, oslot * B + u, lane, o[u], o[u + 1]);
                    if (oslot == 0) {
#pragma unroll
                        for (int u = 0; u < B; u += 2) st2(fout, 3 * B + u, lane, o[u], o[u + 1]);
                    }
                }
                slot = slot == nf - 1 ? 0 : slot + 1;
                oslot = oslot == 2 ? 0 : oslot + 1;
                s0 = s0 + B >= nf * B ? s0 + B - nf * B : s0 + B;
            }
            {                                                  // finish block j - 5 (stores always issued)
                const int bs = j - 5;
                float wq[FN], oq[FN], a0[FN], a1[FN];
                unsigned long long okm = ~0ull;
#pragma unroll
                for (int h = 0; h < FN; h++) {
                    a0[h] = outb[bs & 1][f_u + 8 * h][lane];
                    a1[h] = outb[2 + (bs & 1)][f_u + 8 * h][lane];
                    wq[h] = box_divide(a0[h], denom, okm);     // deferred flagging.py:419
                    oq[h] = box_divide(a1[h], denom, okm);
                }
                if (okm != ~0ull) {
                    TRI_KEEP_BRANCH();
#pragma unroll
                    for (int h = 0; h < FN; h++) { wq[h] = box_divide_ieee(a0[h], denom); oq[h] = box_divide_ieee(a1[h], denom); }
                }
#pragma unroll
                for (int h = 0; h < FN; h++) {
                    const int i = bs * B + f_u + 8 * h - 4 * r;
                    const bool ok = bs >= 0 && bs < NB && i >= 0 && i < n && colok;
                    const unsigned off = ok ? (unsigned)i * rowb + coff : OOB;
                    const float bg = (wq[h] == 0.0f) ? NAN : oq[h] / wq[h];
                    const float dv = dpre[q][h];
                    if (MODE == 1) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fabsf(dv - bg)), ors, (int)off, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bg), ors, (int)off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dv - bg), wrs, (int)off, 0, 0);
                        line_nan |= ok && isnan(bg);
                    }
                }
            }
            issue_data(j + P - 5, q);                          // data samples of the block finished P iterations from now
#if BOXQF_ABLATE_BARRIER                 // timing-only build (results are wrong on purpose): what the workgroup barrier costs
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#else
            __syncthreads();
#endif
        }
    }
    if (MODE == 2 && line_nan) nanflag[win * (size_t)C + c] = 1;
}
