// kernels_boxweight.hpp -- K4w: the WEIGHT image of the time-axis box-Gaussian stage in integer arithmetic,
// all four running sums in one thread, delay lines bit- / byte- / halfword-packed in registers.
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once
#include <type_traits>

// ---------------------------------------------------------------------------
// K4w  masked_gaussian_filter's weight image along time (flagging.py:500-505 builds weight = !flag, :362-419 filters
// it).  Its input is 0 / 1, so every stage value is an INTEGER: stage k's running sum is at most (2r + 1)^k, float32
// holds the first three exactly while (2r + 1)^3 <= 2^24 and the last stage's sum (< 2^31 for r <= 107) rounds to
// float32 exactly like the reference's float64 -> float32 store (this is what k_boxt / k_boxq already rely on for
// their int32 weight pipelines, kernels_boxline.hpp).  Integer sums are exact, hence order-free -- and SMALL:
//   stage 1 input   0 / 1            one BIT per delay slot        2r / 32 registers
//   stage 2 input   <= 2r + 1 <= 255 one BYTE per slot             2r / 4
//   stage 3 input   <= (2r + 1)^2 < 65536: one HALFWORD per slot   2r / 2
//   stage 4 input   <= (2r + 1)^3    one register per slot         2r
// 1.78 x 2r registers for all four delay lines of a line instead of 4 x 2r -- so ONE thread runs the whole cascade of
// its line even at r = 54 (193 registers), where the float kernels need a stage pipeline across four waves with
// LDS FIFOs and a workgroup barrier every 8 positions (k_boxq: 7.0 - 8.5 ms per 1008-window launch although it only
// moves 5 bytes per sample).  No LDS, no barrier, ~25 full-rate 32-bit instructions per sample.
//
// The loop is unrolled over blocks of 2r steps: a slot's register, bit offset and byte lane are compile-time
// constants ("R[slot] = x" is a rename or one v_lshl_add), and a TF4 flag word (four consecutive times of one channel)
// starts a group of four steps at a static position (two block bodies when 2r = 2 mod 4).  A packed slot is replaced by ADDING (new - old) << offset: the
// field holds `old`, becomes `new`, and no carry or borrow leaves it.
// Cascade (the single-sweep form of K4b / boxq_body, all stages at the same t -- integer adds are short enough that the
// chain s1 -> s2 -> s3 -> s4 of one step never waits):
//     out_k[t] = sum_{j = t - 2r .. t} in_k[j],  in_1[t] = weight[t] (0 for t >= n),  in_2 = out_1 (boxq_body masks it for
//     t >= n + 2r, where it is 0 anyway in exact arithmetic),  in_3 = out_2,  in_4[t] = out_3[t] for t >= 2r else 0,
//     result[i] = float32(out_4[i + 4r]) / float32(2r + 1)**4,  0 <= i < n
// grid (ceil(C / 64), W), block 64; host: n % 4 == 0, n * C * 4 < 2^31, 8 <= 2r <= 214, 2r even.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int boxw_unroll(int r2) { return r2 % 4 == 0 ? r2 : 2 * r2; }
__host__ __device__ constexpr int boxw_regs(int r2) { return (r2 + 31) / 32 + (r2 + 3) / 4 + (r2 + 1) / 2 + r2 + 56; }   // (measured: packed lines + 50 ... 56)
__host__ __device__ constexpr int boxw_waves(int r2) { return boxw_regs(r2) <= 128 ? 4 : (boxw_regs(r2) <= 168 ? 3 : 2); }
#ifndef BOXW_SCHED_EVERY
#define BOXW_SCHED_EVERY 2               // a scheduling barrier after every so many steps (bounds live ranges)
#endif
#ifndef BOXW_PREFETCH
#define BOXW_PREFETCH 4                  // flag words in flight per lane (one word = four steps)
#endif

// quotient by the launch constant for a = +0 or a >= 1 (an integer-valued float32): a * y is then +0 or a normal number,
// the class for which box_divide() is proven exact (kernels_boxline.hpp; checked exhaustively per radius) -- no class test
__device__ __forceinline__ float boxw_divide(float a, const BoxDenom dn) {
    float q = a * dn.y;
    float r = __builtin_fmaf(-dn.b, q, a);
    q = __builtin_fmaf(r, dn.y, q);
    r = __builtin_fmaf(-dn.b, q, a);
    q = __builtin_fmaf(r, dn.y, q);
    return q;
}

// acc + (x << SH) as ONE opaque instruction: written as plain C++ the compiler reassociates the four (eight, 32) inserts of a
// packed word into a tree it evaluates at the end of the block -- every delta then stays live for a whole block (+70
// registers at 2r = 64).  SH is a compile-time constant once the block is unrolled.
#define BOXW_LSHL_ADD(acc, x, SH) asm("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "n"(SH))

template <int R2>
__global__ void __launch_bounds__(64, boxw_waves(R2))
k_boxw(const uint8_t* __restrict__ srcFlags, float* __restrict__ dstW, int n, int C, BoxDenom denom, size_t sws, size_t dws) {
    constexpr int N1 = (R2 + 31) / 32, N2 = (R2 + 3) / 4, N3 = (R2 + 1) / 2;
    constexpr int P = BOXW_PREFETCH;
    constexpr unsigned OOB = 0x7ffffff0u;
    static_assert(R2 % 2 == 0 && R2 >= 8 && R2 <= 214, "2r even, stage sums within a byte / a halfword / int32");
    const int lane = threadIdx.x;
    const size_t win = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const bool colok = c < C;
    const unsigned rowb = (unsigned)C * 4u;
    const unsigned coff = colok ? (unsigned)c * 4u : OOB;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)(srcFlags + win * sws), 0, (int)((unsigned)(n / 4) * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(dstW + win * dws), 0, (int)((unsigned)n * rowb), 0x00020000);

    unsigned D1[N1], D2[N2], D3[N3], D4[R2];
#pragma unroll
    for (int k = 0; k < N1; k++) D1[k] = 0;
#pragma unroll
    for (int k = 0; k < N2; k++) D2[k] = 0;
#pragma unroll
    for (int k = 0; k < N3; k++) D3[k] = 0;
#pragma unroll
    for (int k = 0; k < R2; k++) D4[k] = 0;
    unsigned s1 = 0, s2 = 0, s3 = 0, s4 = 0;

    // flag words of the coming groups: pf[0] is the next group's; rows beyond the image return 0 (and are masked)
    unsigned pf[P];
    const int ngrp = n / 4;
#pragma unroll
    for (int q = 0; q < P; q++)
        pf[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, (int)((colok && q < ngrp) ? (unsigned)q * rowb + coff : OOB), 0, 0);

    const int nsteps = n + 2 * R2;                             // t in [0, n + 4r)
    unsigned wb = 0;                                           // weight bytes of the current group of four steps
    // One block = 2r steps (every slot once).  PH: t0 mod 4 -- with 2r = 2 (mod 4) every second block starts in the middle
    // of a flag word (its first two steps take the upper half of the word the block before loaded), hence two bodies.
    auto block = [&](auto phase, const int t0, const unsigned head) {
        constexpr int PH = decltype(phase)::value;
#pragma unroll
        for (int u = 0; u < R2; u++) {
            const int t = t0 + u;
            const int slot = u;                                // (static after unrolling)
            if ((u + PH) % 4 == 0) {
                // four consecutive times of this channel: weight byte = 1 where the flag byte is zero
                const unsigned fw = pf[0];
#pragma unroll
                for (int q = 0; q + 1 < P; q++) pf[q] = pf[q + 1];
                const int g = (t >> 2) + P;
                pf[P - 1] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(frs, (int)(g < ngrp ? (unsigned)g * rowb + coff : OOB), 0, 0);
                const unsigned nz = ((((fw & 0x7f7f7f7fu) + 0x7f7f7f7fu) | fw) >> 7) & 0x01010101u;
                wb = t < n ? (nz ^ 0x01010101u) : 0u;          // beyond the line end: flagged (n % 4 == 0: whole words)
            }
            const unsigned in1 = (wb >> (8 * ((u + PH) % 4))) & 1u;
            // stage 1: one bit per slot
            const unsigned old1 = (D1[slot >> 5] >> (slot & 31)) & 1u;
            const unsigned o1 = s1 + in1;
            const unsigned d1 = in1 - old1;
            s1 = o1 - old1;
            BOXW_LSHL_ADD(D1[slot >> 5], d1, slot & 31);
            // stage 2: one byte per slot
            const unsigned old2 = (D2[slot >> 2] >> (8 * (slot & 3))) & 0xffu;
            const unsigned o2 = s2 + o1;
            s2 = o2 - old2;
            const unsigned d2 = o1 - old2;
            BOXW_LSHL_ADD(D2[slot >> 2], d2, 8 * (slot & 3));
            // stage 3: one halfword per slot
            const unsigned old3 = (D3[slot >> 1] >> (16 * (slot & 1))) & 0xffffu;
            const unsigned o3 = s3 + o2;
            s3 = o3 - old3;
            const unsigned d3 = o2 - old3;
            BOXW_LSHL_ADD(D3[slot >> 1], d3, 16 * (slot & 1));
            // stage 4: one register per slot (a rename); it takes out_3[t] from t = 2r on (head = 0 in the first block)
            const unsigned in4 = o3 & head;
            const unsigned old4 = D4[slot];
            D4[slot] = in4;
            const unsigned o4 = s4 + in4;
            s4 = o4 - old4;
            // out_4[t] -> result row t - 4r: lane offset in the vector register (out of range when the row or the column is),
            // row offset in a scalar register
            const int i = t - 2 * R2;
            const bool rowok = (unsigned)i < (unsigned)n;
            const float y = boxw_divide((float)o4, denom);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ors, (int)(rowok ? coff : OOB), (int)(rowok ? (unsigned)i * rowb : 0u), 0);
            // (the ILP scheduler would otherwise interleave many steps and stretch the live ranges of their temporaries:
            //  at 2r >= 80 that costs 30 - 40 spilled registers)
            if (BOXW_SCHED_EVERY > 0 && (u + 1) % BOXW_SCHED_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t0 = 0; t0 < nsteps; t0 += boxw_unroll(R2)) {
        block(std::integral_constant<int, 0>{}, t0, t0 == 0 ? 0u : ~0u);
        if constexpr (R2 % 4 != 0) block(std::integral_constant<int, 2>{}, t0 + R2, ~0u);
    }
}
