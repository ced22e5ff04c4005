// kernels_reject_tile.hpp -- K3t: block median + rejection of the background loop in one pass over the residual, TILE-PARALLEL
// Part of the single translation unit tricolour_amd.hip (see there for the overview; kernels_reject.hpp for the idea).
#pragma once

// ---------------------------------------------------------------------------
// K3t  The one-pass median + rejection of kernels_reject.hpp (K3r) cut so that the pass over |data - background| has the
// shape of k_reject4_t -- one workgroup per 64 x 64-word tile, hundreds of thousands of them -- instead of one workgroup
// per (window, chunk) block walking 28 tiles one after the other (K3r streams at 2 TB/s that way: every batch of loads is
// waited for by the wave that issued it, with five waves per SIMD to hide it).  Per background iteration:
//   k_mr_predict  (G, W)            per block: key range + predicted median bin from 64 runs of 256 consecutive samples
//                                   (4 % of the block) -> candidate window, decision bracket [thrA, thrB]; clears the block's
//                                   histogram and counters (global memory)
//   k_mr_pass     (tiles, W)        per tile: LDS histogram of the unflagged keys (flushed to the block's histogram with one
//                                   atomic per occupied bin), window keys and undecided samples (index, value) compacted per
//                                   WAVE in LDS (ballot + mbcnt: no atomic, no branch per sample) and appended to the block's
//                                   lists with one reservation per wave and tile; decided flags to the FT image and, transposed,
//                                   to the TF4 image
//   k_mr_finish   (G, W)            per block: median's bin from the histogram, exact select over the window keys, bracket
//                                   verification, the undecided samples against the exact threshold -- the pass decided them by a
//                                   PROVISIONAL threshold (the middle of the predicted bin), so only those between the two are rewritten
//   k_mr_pass / k_mr_finish again   ROUND 2 for the blocks whose median fell outside the window or the decision bracket: the
//                                   histogram is complete, so the bin of the median is now KNOWN -- the window becomes that bin
//                                   +- 1 and the same two kernels run once more over those blocks' tiles only (all other
//                                   workgroups leave at once); a block redone by ONE workgroup costs 1.8 ms of tail per launch
//   k_median_reject (G, W)          whatever still fails (list overflow, nothing to predict from): redone from the input flags by
//                                   one workgroup per block (K3r's fallback)
// Block status: 1 = round 1, 2 = round 2, 3 = done, 0 = redo by k_median_reject.
// Tiles are chunk-aligned (a tile never straddles two blocks): grid.y enumerates (chunk, 64-row group of the chunk).
// 7 B / sample in k_mr_pass, ~0.3 in the other kernels, against 12.2 for k_median2 + k_reject4_t.
// ---------------------------------------------------------------------------
#ifndef MRT_CWIN
#define MRT_CWIN 4u                      // half-width (bins) of the candidate window around the predicted bin
#endif
#ifndef MRT_DWIN
#define MRT_DWIN 2u                      // half-width of the decision bracket (<= MRT_CWIN)
#endif
#ifndef MRT_WCAND
#define MRT_WCAND 384                    // window keys a WAVE of a tile can hold in LDS (4096 samples: ~4 % expected)
#endif
#ifndef MRT_WUND
#define MRT_WUND 192                     // undecided samples (index, value) per wave and tile (~1.5 % expected)
#endif
#define MRT_PARW 16                      // words of a block's parameter record
#define MRT_ANY_ROUND2 0x10000u           // in pad[0] of a window's FIRST record: some block of the window goes into a second round
struct MrtPar {                          // (global memory, one per block; written by k_mr_predict, counters by k_mr_pass)
    unsigned lo, S, wlo, whi;            // histogram map, candidate window (bins)
    double thrA, thrB;                   // decision bracket
    unsigned ncand, nund, below1, status;   // list fill, largest key below the window + 1, 1 = predicted / 0 = redo the block
    unsigned pad[2];                     // [0] rounds beyond the first, [1] why the block is redone (statistics)
    double thrP;                         // provisional threshold (predicted median x scale), thrA <= thrP <= thrB
};
static_assert(sizeof(MrtPar) == MRT_PARW * 4, "parameter record");

// scratch of one window (words): [G records][G histograms of 2048][G candidate lists of ccap][G undecided lists of 2 ucap]
__host__ __device__ inline size_t mrt_scratch_words(int G, size_t ccap, size_t ucap) {
    return (size_t)G * (MRT_PARW + SEL_BINS + ccap + 2 * ucap);
}
struct MrtLayout {
    unsigned* base; int G; size_t ccap, ucap;
    __device__ MrtPar* par(int g) const { return reinterpret_cast<MrtPar*>(base + (size_t)g * MRT_PARW); }
    __device__ unsigned* hist(int g) const { return base + (size_t)G * MRT_PARW + (size_t)g * SEL_BINS; }
    __device__ unsigned* cand(int g) const { return base + (size_t)G * (MRT_PARW + SEL_BINS) + (size_t)g * ccap; }
    __device__ unsigned* und(int g) const { return base + (size_t)G * (MRT_PARW + SEL_BINS + ccap) + (size_t)g * 2 * ucap; }
};

// ---- prediction: one workgroup per block --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mr_predict(const float* __restrict__ resid, const uint8_t* __restrict__ flags_in, const int64_t* __restrict__ chunk_ends,
             double scale, int C4, int G, size_t ws_resid, size_t ws_flags, unsigned* __restrict__ gscratch, size_t scratch_ws,
             size_t ccap, size_t ucap, int force_fallback) {
    __shared__ unsigned hist[SEL_BINS];
    __shared__ unsigned sh[4];
    __shared__ unsigned sh_lo, sh_hi, sh_bin;
    const int g = blockIdx.x;
    const size_t win = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MrtLayout lay{gscratch + win * scratch_ws, G, ccap, ucap};
    MrtPar* par = lay.par(g);
    unsigned* ghist = lay.hist(g);
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) ghist[u * 256 + tid] = 0;
    const int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    const int64_t len = (int64_t)(c1 - c0) * C4 * 4;
    auto give_up = [&]() {
        if (tid == 0) { MrtPar p{}; p.status = 0; p.wlo = 1; p.whi = 0; p.thrA = 0.0; p.thrB = 0.0; p.pad[1] = 1; *par = p; }   // (pad[1]: why the block is redone)
    };
    if (force_fallback || len < 65536) { give_up(); return; }
    const float* dblk = resid + win * ws_resid + (size_t)c0 * C4 * 4;
    const uint8_t* fblk = flags_in + win * ws_flags + (size_t)c0 * C4 * 4;
    if (tid == 0) { sh_lo = 0xFFFFFFFFu; sh_hi = 0; }
    __syncthreads();
    const int64_t rstep = len / 64;
    unsigned keys[64];                                                  // the runs are read ONCE: 64 samples per thread (flagged: all ones)
    {
        unsigned kmin = 0xFFFFFFFFu, kmax = 0;
#pragma unroll
        for (int rr = 0; rr < 64; rr++) {
            const int64_t i = rr * rstep + tid;
            const unsigned k = __float_as_uint(dblk[i]) & 0x7FFFFFFFu;
            keys[rr] = fblk[i] ? 0xFFFFFFFFu : k;
        }
#pragma unroll
        for (int rr = 0; rr < 64; rr++) {
            if (keys[rr] != 0xFFFFFFFFu) { kmin = min(kmin, keys[rr]); kmax = max(kmax, keys[rr]); }
        }
        kmin = ~wave_max_u32(~kmin);
        kmax = wave_max_u32(kmax);
        if (lane == 0) { atomicMin(&sh_lo, kmin); atomicMax(&sh_hi, kmax); }
    }
    __syncthreads();
    unsigned lo = sh_lo;
    const unsigned hi = sh_hi;
    if (hi < lo) { give_up(); return; }
    // (the bins are linear in the KEY, i.e. logarithmic in the value: one exactly-zero sample would stretch them over the whole
    //  exponent range -- 1/8 octave each, 40 % of the samples inside the window.  The histogram covers the top 16 octaves of the
    //  sample; everything below counts into bin 0, where a median is not expected -- and the checks catch it if it is there.)
    if (hi > (16u << 23) && lo < hi - (16u << 23)) lo = hi - (16u << 23);
    int S = 0;
    {
        const unsigned span = hi - lo;
        while (S < 31 && (span >> S) >= 2046u) S++;
    }
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = 0;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 64; rr++) {
        const unsigned k = keys[rr];
        if (k != 0xFFFFFFFFu) {
            const unsigned b = k < lo ? 0u : min(((k - lo) >> S) + 1u, 2047u);
            atomicAdd(&hist[b], 1u);
        }
    }
    __syncthreads();
    // bin holding rank ns / 2 of the sample
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
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    unsigned woff = 0, ns = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
        unsigned t2 = sh[w2];
        if (w2 < wave) woff += t2;
        ns += t2;
    }
    const unsigned kk = ns >> 1, exc = woff + inc - sacc;
    if (ns > 0 && kk >= exc && kk < exc + sacc) {
        unsigned c = exc;
        int j = 0;
#pragma unroll
        for (int q = 0; q < 7; q++)
            if (j == q && kk >= c + v[q]) { c += v[q]; j = q + 1; }
        sh_bin = 8u * tid + j;
    }
    __syncthreads();
    if (ns < 4096) { give_up(); return; }
    if (tid == 0) {
        const unsigned bp = sh_bin;
        MrtPar p{};
        p.lo = lo; p.S = (unsigned)S;
        // window widths are RELATIVE (the prediction is good to ~1 % of the value whatever the bin width): MRT_CWIN / MRT_DWIN bins
        // of 1/64 octave (S = 17), scaled to the bins this block has
        const unsigned cw = S >= 17 ? max(MRT_CWIN >> (S - 17), 2u) : min(MRT_CWIN << (17 - S), 200u);
        const unsigned dw = S >= 17 ? max(MRT_DWIN >> (S - 17), 1u) : min(MRT_DWIN << (17 - S), 100u);
        p.wlo = bp > cw + 1u ? bp - cw : 1u;
        p.whi = bp + cw < 2046u ? bp + cw : 2046u;
        const unsigned dlo = bp > dw + 1u ? bp - dw : 1u;
        const unsigned dhi = bp + dw < 2046u ? bp + dw : 2046u;
        const unsigned long long kB64 = (unsigned long long)lo + ((unsigned long long)dhi << S) - 1ull;
        const unsigned kA = lo + ((dlo - 1u) << S), kB = kB64 > 0x7F7FFFFFull ? 0x7F7FFFFFu : (unsigned)kB64;
        p.thrA = (double)__uint_as_float(kA) * scale;
        p.thrB = (double)__uint_as_float(kB) * scale;
        // the middle of the predicted bin: what the pass decides the bracket's samples by until the exact median is known
        const unsigned long long kP64 = (unsigned long long)lo + ((unsigned long long)(bp > 0u ? bp - 1u : 0u) << S) + ((1ull << S) >> 1);
        const unsigned kP = min(max((unsigned)min(kP64, 0x7F7FFFFFull), kA), kB);
        p.thrP = (double)__uint_as_float(kP) * scale;
        p.status = 1;
        *par = p;
    }
}

// ---- the pass: one workgroup per chunk-aligned tile of 64 rows x 64 words -----------------------------------------------
// grid (ceil(C4 / 64), sum over chunks of ceil(rows / 64), W), block 256
#ifndef MRT_PASS_WAVES
#define MRT_PASS_WAVES 4                 // waves per SIMD the pass is compiled for (HIP: second __launch_bounds__ argument)
#endif
__global__ void __launch_bounds__(256, MRT_PASS_WAVES)
k_mr_pass(const float* __restrict__ resid, const uint8_t* __restrict__ flags_in, uint8_t* __restrict__ flags_out,
          uint8_t* __restrict__ flags_t4, const int64_t* __restrict__ chunk_ends, int L, int C4, int G,
          size_t ws_resid, size_t ws_flags, unsigned* __restrict__ gscratch, size_t scratch_ws, size_t ccap, size_t ucap, unsigned round) {
    __shared__ unsigned hist[SEL_BINS + 64];
    __shared__ unsigned tile[64][65];
    __shared__ unsigned lcand[4][MRT_WCAND];
    __shared__ uint2 lund[4][MRT_WUND];
    const size_t win = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = lane, ty = wave;
    // second round: a window none of whose blocks asked for it is left before anything else is looked up (0.36 ms per launch
    // for 282 k workgroups that each searched their chunk and read their block's status just to find it done)
    if (round == 2 && !(reinterpret_cast<const MrtPar*>(gscratch + win * scratch_ws)->pad[0] & MRT_ANY_ROUND2)) return;
    // the tile's chunk and first row: grid.y counts 64-row groups chunk by chunk
    int g = 0, l0 = 0, c1 = 0;
    {
        int y = blockIdx.y;
        for (g = 0; g < G; g++) {
            const int a = (int)chunk_ends[g], b = (int)chunk_ends[g + 1];
            const int nt = (b - a + 63) / 64;
            if (y < nt) { l0 = a + 64 * y; c1 = b; break; }
            y -= nt;
        }
        if (g >= G) return;                                            // (grid sized by the host: not reached)
    }
    const MrtLayout lay{gscratch + win * scratch_ws, G, ccap, ucap};
    MrtPar* par = lay.par(g);
    // (the status word may be cleared by another tile of the block at any time: ONE thread reads it for the workgroup)
    __shared__ unsigned sh_status;
    if (tid == 0) sh_status = __hip_atomic_load(&par->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_status != round) return;                                    // (uniform) not this round's block
    const unsigned lo = par->lo, S = par->S, wlo = par->wlo, wspan = par->whi - par->wlo;
    const double thrA = par->thrA, thrB = par->thrB, thrP = par->thrP;
    const int w0 = blockIdx.x * 64, w = w0 + tx;
    const float4* r4 = reinterpret_cast<const float4*>(resid + win * ws_resid);
    const unsigned* fin = reinterpret_cast<const unsigned*>(flags_in + win * ws_flags);
    unsigned* fout = reinterpret_cast<unsigned*>(flags_out + win * ws_flags);
    unsigned* ft = reinterpret_cast<unsigned*>(flags_t4 + win * ws_flags);
    constexpr unsigned OOBR = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t fors = __builtin_amdgcn_make_buffer_rsrc((void*)fout, 0, (int)((unsigned)L * (unsigned)C4 * 4u), 0x00020000);
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = 0;
    // all 16 words of this thread (rows ty, ty + 4, ...): loads first
    float4 rv[16];
    unsigned f[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int l = l0 + ty + 4 * q;
        const size_t i = (l < c1 && w < C4) ? (size_t)l * C4 + w : (size_t)l0 * C4;
        rv[q] = r4[i];
        f[q] = fin[i];
    }
    __syncthreads();
    unsigned mb1 = 0;
    unsigned ccnt = 0, ucnt = 0;                                       // this wave's list fill (uniform)
    bool over = false;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int l = l0 + ty + 4 * q;
        const bool valid = l < c1 && w < C4;
        const unsigned i = (unsigned)l * (unsigned)C4 + (unsigned)w;
        const float xv[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
        unsigned fn = f[q];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const unsigned k = __float_as_uint(xv[k4]) & 0x7FFFFFFFu;
            const bool unfl = valid && ((f[q] >> (8 * k4)) & 0xFFu) == 0u;
            const unsigned b = k < lo ? 0u : min(((k - lo) >> S) + 1u, 2047u);
            atomicAdd(&hist[unfl ? b : SEL_BINS + (unsigned)lane], 1u);
            mb1 = max(mb1, (unfl && b < wlo) ? k + 1u : 0u);
            // window keys: compacted into this wave's LDS list
            const bool inwin = unfl && (b - wlo <= wspan);
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(inwin);
            if (cm) {                                                   // (uniform)
                const unsigned pos = ccnt + __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                if (inwin && pos < MRT_WCAND) lcand[wave][pos] = k;
                ccnt += (unsigned)__builtin_popcountll(cm);
                if (ccnt > MRT_WCAND) {                                 // (uniform, rare) this wave's LDS list is full: straight to the block's list
                    const bool sp = inwin && pos >= MRT_WCAND;
                    const unsigned long long sm = __builtin_amdgcn_ballot_w64(sp);
                    unsigned b2 = 0;
                    if (lane == 0 && sm) b2 = atomicAdd(&par->ncand, (unsigned)__builtin_popcountll(sm));
                    b2 = (unsigned)__builtin_amdgcn_readfirstlane((int)b2);
                    const size_t at = (size_t)b2 + __builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                    if (sp && at < ccap) lay.cand(g)[at] = k;
                }
            }
            const double dx = (double)xv[k4];
            // decided: x > thrB flagged, x <= thrA not; between them (the undecided ~1 %, listed for k_mr_finish) the PROVISIONAL threshold
            // decides for now -- the finish then only rewrites the few samples between it and the exact one
            // (a NaN compares false every time: never flagged, never listed)
            const bool gtB = dx > thrB, gtA = dx > thrA, gtP = dx > thrP;
            fn |= gtP ? (1u << (8 * k4)) : 0u;
            const bool und = unfl && gtA && !gtB;
            const unsigned long long um = __builtin_amdgcn_ballot_w64(und);
            if (um) {
                const unsigned pos = ucnt + __builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
                const uint2 ent = make_uint2(i * 4u + (unsigned)k4, __float_as_uint(xv[k4]));
                if (und && pos < MRT_WUND) lund[wave][pos] = ent;
                ucnt += (unsigned)__builtin_popcountll(um);
                if (ucnt > MRT_WUND) {                                  // (uniform, rare: a tile full of samples near the threshold)
                    const bool sp = und && pos >= MRT_WUND;
                    const unsigned long long sm = __builtin_amdgcn_ballot_w64(sp);
                    unsigned b2 = 0;
                    if (lane == 0 && sm) b2 = atomicAdd(&par->nund, (unsigned)__builtin_popcountll(sm));
                    b2 = (unsigned)__builtin_amdgcn_readfirstlane((int)b2);
                    const size_t at = (size_t)b2 + __builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
                    if (sp && at < ucap) reinterpret_cast<uint2*>(lay.und(g))[at] = ent;
                }
            }
        }
        __builtin_amdgcn_raw_buffer_store_b32(fn, fors, (int)(valid ? i * 4u : OOBR), 0, 0);
        tile[ty + 4 * q][tx] = fn;
    }
    over = false;                                                       // (a full wave list spills to the block's list: nothing fails here)
    ccnt = min(ccnt, (unsigned)MRT_WCAND);                              // what is left in LDS
    ucnt = min(ucnt, (unsigned)MRT_WUND);
    mb1 = wave_max_u32(mb1);
    // this wave's lists to the block's: one reservation each
    unsigned cbase = 0, ubase = 0;
    if (lane == 0) {
        if (mb1) atomicMax(&par->below1, mb1);
        if (over) { atomicExch(&par->status, 0u); par->pad[1] = 2 + (ccnt > MRT_WCAND ? 0 : 8); }   // a wave's list overflowed: the block is redone
        if (ccnt && !over) cbase = atomicAdd(&par->ncand, ccnt);
        if (ucnt && !over) ubase = atomicAdd(&par->nund, ucnt);
    }
    cbase = (unsigned)__builtin_amdgcn_readfirstlane((int)cbase);
    ubase = (unsigned)__builtin_amdgcn_readfirstlane((int)ubase);
    __syncthreads();                                                    // tile[] and hist[] complete (and this wave's lists)
    // FT words transposed -> TF4
    for (int j = ty; j < 64; j += 4) {
        const int wq = w0 + j, l = l0 + tx;
        if (l < c1 && wq < C4) ft[(size_t)wq * L + l] = tile[tx][j];
    }
    // histogram: one atomic per occupied bin (round 2: the block's histogram is already complete)
    if (round == 1) {
        unsigned* ghist = lay.hist(g);
#pragma unroll
        for (int u = 0; u < SEL_BINS / 256; u++) {
            const unsigned h = hist[u * 256 + tid];
            if (h) atomicAdd(&ghist[u * 256 + tid], h);
        }
    }
    if (!over) {
        unsigned* gc = lay.cand(g);
        uint2* gu = reinterpret_cast<uint2*>(lay.und(g));
        for (unsigned j = lane; j < ccnt; j += 64)
            if ((size_t)cbase + j < ccap) gc[cbase + j] = lcand[wave][j];
        for (unsigned j = lane; j < ucnt; j += 64)
            if ((size_t)ubase + j < ucap) gu[ubase + j] = lund[wave][j];
    }
}

// ---- finish: one workgroup per block ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mr_finish(const float* __restrict__ resid, const uint8_t* __restrict__ flags_in, uint8_t* __restrict__ flags_out,
            uint8_t* __restrict__ flags_t4, double* __restrict__ med, const int64_t* __restrict__ chunk_ends, double scale, int L, int C4, int G,
            size_t ws_resid, size_t ws_flags, unsigned* __restrict__ gscratch, size_t scratch_ws, size_t ccap, size_t ucap, unsigned round) {
    __shared__ unsigned hist[SEL_BINS];
    __shared__ unsigned sh[9];
    __shared__ unsigned sh_bin, sh_exc, sh_excw;
    const int g = blockIdx.x;
    const size_t win = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MrtLayout lay{gscratch + win * scratch_ws, G, ccap, ucap};
    MrtPar* par = lay.par(g);
    if (par->status != round) return;                                  // (uniform: nothing writes it while this kernel runs but this workgroup)
    const unsigned wlo = par->wlo, whi = par->whi, nc = par->ncand, nu = par->nund, below1 = par->below1;
    const double thrA = par->thrA, thrB = par->thrB, thrP = par->thrP;
    const unsigned* ghist = lay.hist(g);
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = ghist[u * 256 + tid];
    __syncthreads();
    // bin holding rank total / 2, keys below it, keys below bin wlo (k_median2's locate)
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
    if (lane == 63) sh[wave] = inc;
    if (tid == 0) sh_bin = 0xFFFFFFFFu;
    __syncthreads();
    unsigned woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
        unsigned t2 = sh[w2];
        if (w2 < wave) woff += t2;
        total += t2;
    }
    const unsigned kk = total >> 1, exc = woff + inc - sacc;
    if ((unsigned)tid == (wlo >> 3)) {
        unsigned c = exc;
#pragma unroll
        for (int q = 0; q < 7; q++)
            if ((unsigned)q < (wlo & 7u)) c += v[q];
        sh_excw = c;
    }
    if (total > 0 && kk >= exc && kk < exc + sacc) {
        unsigned c = exc;
        int j = 0;
#pragma unroll
        for (int q = 0; q < 7; q++)
            if (j == q && kk >= c + v[q]) { c += v[q]; j = q + 1; }
        sh_bin = 8u * tid + j;
        sh_exc = c;
    }
    __syncthreads();
    const unsigned bsel = sh_bin, excw = sh_excw;
    auto fail = [&](unsigned why) { if (tid == 0) { par->status = 0; par->pad[1] = why; } };
    // Round 2 set-up: the histogram says which bin holds the median -- window and decision bracket become that bin +- 1
    auto second_round = [&]() {
        if (tid == 0) {
            const unsigned lo = par->lo, S = par->S;
            const unsigned nlo = bsel > 2u ? bsel - 1u : 1u, nhi = bsel + 1u < 2046u ? bsel + 1u : 2046u;
            const unsigned long long kB64 = (unsigned long long)lo + ((unsigned long long)nhi << S) - 1ull;
            const unsigned kA = lo + ((nlo - 1u) << S), kB = kB64 > 0x7F7FFFFFull ? 0x7F7FFFFFu : (unsigned)kB64;
            par->wlo = nlo; par->whi = nhi;
            par->thrA = (double)__uint_as_float(kA) * scale;
            par->thrB = (double)__uint_as_float(kB) * scale;
            const unsigned long long kP64 = (unsigned long long)lo + ((unsigned long long)(bsel - 1u) << S) + ((1ull << S) >> 1);
            par->thrP = (double)__uint_as_float(min(max((unsigned)min(kP64, 0x7F7FFFFFull), kA), kB)) * scale;
            par->ncand = 0; par->nund = 0; par->below1 = 0;
            atomicAdd(&par->pad[0], 1u);                                // (statistics: rounds beyond the first)
            atomicOr(&lay.par(0)->pad[0], MRT_ANY_ROUND2);              // this window has work for the second round's launches
            atomicAdd(&g_medrej_stats[3], 1ull);
            par->status = 2;
        }
    };
    if (!(total > 0 && nc <= ccap && nu <= ucap && bsel >= 1 && bsel <= 2046)) { fail(total == 0 ? 3u : (bsel < 1 || bsel > 2046 ? 4u : 5u)); return; }
    if (!(bsel >= wlo && bsel <= whi)) {                               // the median is outside the window
        if (round == 1) second_round(); else fail(6u);
        return;
    }
    __syncthreads();   // sh[] is reused by select3 below
    const unsigned* gcand = lay.cand(g);
    auto enumerate_gc = [&](auto&& visit) {
        const uint4* g4 = reinterpret_cast<const uint4*>(gcand);
        const unsigned n4 = nc >> 2;
        unsigned i = tid;
        for (; i + 256 * 7 < n4; i += 256 * 8) {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = g4[i + 256 * u];
#pragma unroll
            for (int u = 0; u < 8; u++) { visit(q[u].x); visit(q[u].y); visit(q[u].z); visit(q[u].w); }
        }
        for (; i < n4; i += 256) { const uint4 q = g4[i]; visit(q.x); visit(q.y); visit(q.z); visit(q.w); }
        if ((unsigned)tid < (nc & 3u)) visit(gcand[(n4 << 2) + tid]);
    };
    Sel3State st = select3(hist, sh, enumerate_gc, (long long)((total >> 1) - excw));
    st.n = total;
    double m;
    if (st.n == 0) m = __longlong_as_double(0x7FF8000000000000LL);
    else if (st.n & 1u) m = (double)__uint_as_float(st.hi);
    else {
        const unsigned lo2 = st.lo_found ? st.lo : below1 - 1;
        const float sm = __uint_as_float(lo2) + __uint_as_float(st.hi);
        m = (double)sm / 2.0;
    }
    const double thr = m * scale;
    if (tid == 0) med[win * (size_t)G + g] = m;
    if (!(thr >= thrA && thr <= thrB)) {
        // the exact median lies outside the DECISION bracket: the pass's decisions cannot be trusted
        if (round == 1) second_round(); else fail(7u);
        return;
    }
    // the undecided samples against the exact threshold
    const uint2* gu = reinterpret_cast<const uint2*>(lay.und(g));
    uint8_t* fo8 = flags_out + win * ws_flags;
    uint8_t* ft8 = flags_t4 + win * ws_flags;
    const unsigned T = (unsigned)C4 * 4u;
    for (unsigned j = tid; j < nu; j += 256) {
        const uint2 e = gu[j];
        // (listed samples were unflagged on input and carry the provisional decision x > thrP: only a different exact decision is written)
        const double x = (double)__uint_as_float(e.y);
        const bool d = x > thr;
        if (d != (x > thrP)) {
            const unsigned l = e.x / T, t = e.x - l * T;
            const uint8_t v = d ? 1 : 0;
            fo8[e.x] = v;
            ft8[((size_t)(t >> 2) * L + l) * 4 + (t & 3u)] = v;
        }
    }
    if (tid == 0) par->status = 3;                                      // done
}
