// kernels_reject.hpp -- K3r: block median and rejection of the background loop in ONE pass over the residual
// Part of the single translation unit tricolour_amd.hip (see there for the overview).
#pragma once

// ---------------------------------------------------------------------------
// K3r  _get_background2d's rejection step (flagging.py:553-574): per (window, frequency chunk) block
//     threshold = median(|data - background| over the block's unflagged samples) * 1.4826 * reject_threshold
//     flags |= |data - background| > threshold
// used to be two full passes over the |data - background| image per background iteration -- the exact block median
// (k_median2: 5.2 B / sample) and the rejection with its TF4 re-pack (k_reject4_t: 7 B / sample), 17 % of a stage-1
// call.  The rejection cannot start before the median is known -- but almost every sample can be DECIDED before:
// k_median2's prediction (64 runs of 256 consecutive samples give the bin of the median; the histogram pass collects
// the keys within +-8 bins and proves afterwards whether the median is among them) also brackets the threshold,
//     thrA = value(lowest key of the window) * scale  <=  threshold  <=  value(highest key) * scale = thrB,
// (float64 products of a monotone map: the bracket holds whenever the window holds the median) so during the ONE pass
//     x <= thrA  -> stays as it is            x > thrB  -> flagged            else -> its index goes to a short list
// (~2 % of the samples: those within +-4 % of the threshold), and once the exact median is selected from the window's
// keys the listed samples are compared with the exact threshold and the few that exceed it flagged by single byte
// stores.  One pass: 4 B (sample) + 1 B (flag) read, 1 B (FT flag) + 1 B (TF4 flag word for the next time-axis
// stage) written = 7 B / sample instead of 12.2.
// Exactness: the median is the same exact selection as k_median2's (select3 over the window's keys, even counts
// through the largest key below the window); the rejection compares (double)x > (double)median * scale exactly as
// k_reject4_t does -- for listed samples with the exact threshold, for the others through the bracket, which is
// verified (thrA <= threshold <= thrB) before anything is trusted.  Whenever the prediction does not hold -- too
// few unflagged samples to predict from, median outside the window, a list overflowing -- the block is redone from
// its INPUT flags by the plain three-pass select and a full rejection pass (flags are written to a second image:
// the input is never modified).  NaN samples (background NaN under flags) are never flagged by the comparison.
// One workgroup per (chunk, window): rows [chunk_ends[g], chunk_ends[g + 1]) of L rows of C4 words (4 samples each).
// grid (G, W), block 256.  Host: images 16-byte aligned, window strides % 4 == 0, L * C4 * 4 < 2^31.
// ---------------------------------------------------------------------------
#ifndef MEDREJ_ABLATE
#define MEDREJ_ABLATE 0                  // timing-only builds (results wrong on purpose): 1 no histogram atomics, 2 no list stores,
                                         // 4 no float64 decisions, 8 no TF4 transposition (tile + barriers)
#endif
#ifndef MEDREJ_DWIN
#define MEDREJ_DWIN 3u                   // half-width (bins) of the decision bracket around the predicted bin (<= MED2_WIN)
#endif
#ifndef MEDREJ_RB
#define MEDREJ_RB 4                      // rows of a tile a thread keeps in flight (16-byte sample group + flag word each)
#endif

// per-process statistics (test hook tri_medrej_stats): blocks run, blocks that fell back before the pass (nothing to predict
// from), after it (median outside the window / a list overflowing), at the bracket verification
__device__ unsigned long long g_medrej_stats[20];      // [4 + r]: blocks of the tile-parallel form (K3t) redone for reason r; [3] also its second rounds

__global__ void __launch_bounds__(256)
k_median_reject(const float* __restrict__ resid, const uint8_t* __restrict__ flags_in, uint8_t* __restrict__ flags_out,
                uint8_t* __restrict__ flags_t4, double* __restrict__ med, const int64_t* __restrict__ chunk_ends,
                double scale, int L, int C4, int G, size_t ws_resid, size_t ws_flags,
                unsigned* __restrict__ gscratch, size_t scratch_ws, unsigned cand_cap, unsigned und_cap, int force_fallback,
                const unsigned* __restrict__ redo_status = nullptr, size_t redo_ws = 0, int redo_stride = 0, int redo_word = 0) {
    // redo_status (the tile-parallel form K3t, kernels_reject_tile.hpp): this launch only REDOES the blocks whose status word
    // -- redo_status[win * redo_ws + g * redo_stride + redo_word] -- is zero, by the fallback below; the others return at once.
    __shared__ unsigned hist[SEL_BINS + 64];                            // (+ 64 dummy bins the flagged samples count into)
    __shared__ unsigned tile[64][65];
    __shared__ unsigned sh[9];
    __shared__ unsigned sh_lo, sh_hi, sh_bin, sh_exc, sh_ncand, sh_nund, sh_below1, sh_mode, sh_excw, sh_dummy, sh_over;
    const int g = blockIdx.x;
    const size_t win = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = lane, ty = wave;
    const int c0 = (int)chunk_ends[g], c1 = (int)chunk_ends[g + 1];
    const float4* r4 = reinterpret_cast<const float4*>(resid + win * ws_resid);
    const unsigned* fin = reinterpret_cast<const unsigned*>(flags_in + win * ws_flags);
    unsigned* fout = reinterpret_cast<unsigned*>(flags_out + win * ws_flags);
    unsigned* ft = reinterpret_cast<unsigned*>(flags_t4 + win * ws_flags);
    const int nrows = c1 - c0;
    const size_t oidx = win * (size_t)G + g;
    if (redo_status) {
        if (redo_status[win * redo_ws + (size_t)g * redo_stride + redo_word] != 0u) return;   // (uniform: nothing writes it during this launch)
        if (threadIdx.x == 0) atomicAdd(&g_medrej_stats[4 + (redo_status[win * redo_ws + (size_t)g * redo_stride + redo_word + 2] & 15u)], 1ull);
        force_fallback = 1;
    }
    if (nrows <= 0) {
        if (tid == 0) med[oidx] = __longlong_as_double(0x7FF8000000000000LL);
        return;
    }
    const size_t wbeg = (size_t)c0 * C4, wend = (size_t)c1 * C4;       // the block's words: [wbeg, wend)
    const int64_t len = (int64_t)(wend - wbeg) * 4;
    const float* dblk = resid + win * ws_resid + wbeg * 4;             // ... its samples as one contiguous slab
    const uint8_t* fblk = flags_in + win * ws_flags + wbeg * 4;
    unsigned* gcand = gscratch + win * scratch_ws + (size_t)g * ((size_t)cand_cap + und_cap);
    unsigned* gund = gcand + cand_cap;

    // every unflagged key of the block (input flags), in this thread's share -- the fallback's enumerator
    auto enumerate_all = [&](auto&& visit) {
        for (size_t i = wbeg + tid; i < wend; i += 256) {
            const float4 rv = r4[i];
            const unsigned f = fin[i];
            if (!(f & 0x000000FFu)) visit(__float_as_uint(rv.x) & 0x7FFFFFFFu);
            if (!(f & 0x0000FF00u)) visit(__float_as_uint(rv.y) & 0x7FFFFFFFu);
            if (!(f & 0x00FF0000u)) visit(__float_as_uint(rv.z) & 0x7FFFFFFFu);
            if (!(f & 0xFF000000u)) visit(__float_as_uint(rv.w) & 0x7FFFFFFFu);
        }
    };
    auto median_of = [&](const Sel3State& st, unsigned below1) -> double {
        if (st.n == 0) return __longlong_as_double(0x7FF8000000000000LL);
        if (st.n & 1u) return (double)__uint_as_float(st.hi);
        const unsigned lo = st.lo_found ? st.lo : below1 - 1;
        const float sm = __uint_as_float(lo) + __uint_as_float(st.hi);
        return (double)sm / 2.0;
    };
    // the rejection of one flag word's four samples against a threshold (k_reject4_t's comparison)
    auto reject_word = [&](const float4 rv, unsigned f, const double thr) -> unsigned {
        if ((double)rv.x > thr) f = (f & 0xFFFFFF00u) | 0x00000001u;
        if ((double)rv.y > thr) f = (f & 0xFFFF00FFu) | 0x00000100u;
        if ((double)rv.z > thr) f = (f & 0xFF00FFFFu) | 0x00010000u;
        if ((double)rv.w > thr) f = (f & 0x00FFFFFFu) | 0x01000000u;
        return f;
    };
    // Tile walk over the block: tiles of 64 rows x 64 words; `word(i, rv, f)` returns the word's new flags, which go to
    // the FT image and, transposed through LDS, to the TF4 image ([C4][L] words: 256-byte runs along every output row).
    // A tile is taken in batches of MEDREJ_RB rows per thread; the loads of batch b + 1 are issued before batch b is
    // processed (two register sets), so a wave always has a batch in flight while it works.
    constexpr unsigned OOBR = 0x7ffffff0u;                              // buffer offset beyond every descriptor: stores are dropped
    const __amdgpu_buffer_rsrc_t fors = __builtin_amdgcn_make_buffer_rsrc((void*)fout, 0, (int)((unsigned)L * (unsigned)C4 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)gcand, 0, (int)(cand_cap * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void*)gund, 0, (int)(und_cap * 4u), 0x00020000);
    constexpr int NBT = 64 / (4 * MEDREJ_RB);                           // batches per tile
    static_assert(NBT >= 1 && 64 % (4 * MEDREJ_RB) == 0, "batch rows");
    const int ntw = (C4 + 63) / 64, ntl = (nrows + 63) / 64;
    const int nbatch = ntl * ntw * NBT;
    auto tile_walk = [&](auto&& word) {
        float4 rvA[MEDREJ_RB], rvB[MEDREJ_RB];
        unsigned fA[MEDREJ_RB], fB[MEDREJ_RB];
        auto coords = [&](int bi, int& l0, int& w0, int& j0) {
            const int tile_i = bi / NBT;
            j0 = ty + (bi - tile_i * NBT) * 4 * MEDREJ_RB;
            const int tl = tile_i / ntw;
            l0 = c0 + tl * 64;
            w0 = (tile_i - tl * ntw) * 64;
        };
        auto load = [&](int bi, float4* rv, unsigned* f) {
            int l0, w0, j0;
            coords(bi, l0, w0, j0);
            const int w = w0 + tx;
#pragma unroll
            for (int q = 0; q < MEDREJ_RB; q++) {
                const int l = l0 + j0 + 4 * q;
                const size_t i = (l < c1 && w < C4) ? (size_t)l * C4 + w : wbeg;
                rv[q] = r4[i];
                f[q] = fin[i];
            }
        };
        auto process = [&](int bi, const float4* rv, const unsigned* f) {
            int l0, w0, j0;
            coords(bi, l0, w0, j0);
            const int w = w0 + tx;
#pragma unroll
            for (int q = 0; q < MEDREJ_RB; q++) {
                // (no branch around a word: words outside the block carry valid = false -- they count into the dummy bins,
                //  append nothing, and their stores go to an out-of-range buffer offset)
                const int l = l0 + j0 + 4 * q;
                const bool valid = l < c1 && w < C4;
                const unsigned i = (unsigned)l * (unsigned)C4 + (unsigned)w;
                const unsigned fn = word(valid, i, rv[q], f[q]);
                __builtin_amdgcn_raw_buffer_store_b32(fn, fors, (int)(valid ? i * 4u : OOBR), 0, 0);
                if (!(MEDREJ_ABLATE & 8)) tile[j0 + 4 * q][tx] = fn;
            }
            if ((bi + 1) % NBT == 0 && !(MEDREJ_ABLATE & 8)) {          // the tile is complete: its words transposed to TF4
                __syncthreads();
                for (int j = ty; j < 64; j += 4) {
                    const int wq = w0 + j, l = l0 + tx;
                    if (l < c1 && wq < C4) ft[(size_t)wq * L + l] = tile[tx][j];
                }
                __syncthreads();
            }
        };
        load(0, rvA, fA);
        for (int bi = 0; bi < nbatch; bi += 2) {
            if (bi + 1 < nbatch) load(bi + 1, rvB, fB);
            process(bi, rvA, fA);
            if (bi + 1 < nbatch) {
                if (bi + 2 < nbatch) load(bi + 2, rvA, fA);
                process(bi + 1, rvB, fB);
            }
        }
    };
    // exact median by the three-pass select over the whole block, then a full rejection pass -- from the INPUT flags
    auto fallback = [&]() {
        __syncthreads();
        const Sel3State st = select3(hist, sh, enumerate_all, -1);
        const double m = median_of(st, 1);
        const double thr = m * scale;
        if (tid == 0) med[oidx] = m;
        tile_walk([&](bool, unsigned, const float4 rv, unsigned f) -> unsigned { return reject_word(rv, f, thr); });
    };

    // ---- pass 0: key range of 64 runs of 256 consecutive samples ----
    if (tid == 0) { sh_lo = 0xFFFFFFFFu; sh_hi = 0; sh_ncand = 0; sh_nund = 0; sh_below1 = 0; sh_mode = 0; sh_dummy = 0; sh_over = 0; }
    __syncthreads();
    const bool can_predict = !force_fallback && len >= 65536;
    if (tid == 0) atomicAdd(&g_medrej_stats[0], 1ull);
    if (!can_predict) { if (tid == 0) atomicAdd(&g_medrej_stats[1], 1ull); fallback(); return; }
    const int64_t rstep = len / 64;
    {
        unsigned kmin = 0xFFFFFFFFu, kmax = 0;
#pragma unroll 8
        for (int rr = 0; rr < 64; rr++) {
            const int64_t i = rr * rstep + tid;
            if (!fblk[i]) {
                const unsigned k = __float_as_uint(dblk[i]) & 0x7FFFFFFFu;
                kmin = min(kmin, k);
                kmax = max(kmax, k);
            }
        }
        kmin = ~wave_max_u32(~kmin);
        kmax = wave_max_u32(kmax);
        if (lane == 0) { atomicMin(&sh_lo, kmin); atomicMax(&sh_hi, kmax); }
    }
    __syncthreads();
    const unsigned lo = sh_lo, hi = sh_hi;
    if (hi < lo) { if (tid == 0) atomicAdd(&g_medrej_stats[1], 1ull); fallback(); return; }   // (uniform) nothing unflagged among the samples
    int S = 0;
    {
        const unsigned span = hi - lo;
        while (S < 31 && (span >> S) >= 2046u) S++;
    }
    auto bin_of = [&](unsigned k) -> unsigned {
        if (k < lo) return 0u;
        const unsigned b = ((k - lo) >> S) + 1u;
        return b > 2047u ? 2047u : b;
    };
    // bin holding rank total / 2 of the histogram -> sh_bin, keys below it -> sh_exc; keys below bin wlo -> sh_excw (k_median2's)
    auto locate = [&](unsigned wlo) -> unsigned {
        unsigned v[8];
        uint4 q0 = reinterpret_cast<const uint4*>(hist)[2 * tid];
        uint4 q1 = reinterpret_cast<const uint4*>(hist)[2 * tid + 1];
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
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
        unsigned woff = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++) {
            unsigned t2 = sh[w2];
            if (w2 < wave) woff += t2;
            total += t2;
        }
        const unsigned kk = total >> 1;
        const unsigned exc = woff + inc - sacc;
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
        return total;
    };

    // ---- pass 0b: predict the median's bin from the same runs ----
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = 0;
    __syncthreads();
#pragma unroll 8
    for (int rr = 0; rr < 64; rr++) {
        const int64_t i = rr * rstep + tid;
        if (!fblk[i]) atomicAdd(&hist[bin_of(__float_as_uint(dblk[i]) & 0x7FFFFFFFu)], 1u);
    }
    __syncthreads();
    const unsigned ns = locate(1);
    if (ns < 4096) { if (tid == 0) atomicAdd(&g_medrej_stats[1], 1ull); fallback(); return; }   // (uniform) too few unflagged samples to predict from
    const unsigned bp = sh_bin;
    const unsigned wlo = bp > MED2_WIN + 1u ? bp - MED2_WIN : 1u;
    const unsigned whi = bp + MED2_WIN < 2046u ? bp + MED2_WIN : 2046u;
    const unsigned wspan = whi - wlo;
    // the window's key range [kA, kB] and the thresholds it brackets
    // (the DECISION bracket is narrower than the candidate window: every undecided sample costs a list entry and, if it turns out
    //  rejected, two scattered byte stores -- MEDREJ_DWIN bins either side of the predicted one keep them at ~1 % of the samples;
    //  a median outside it (but inside the window) fails the verification below and the block takes the fallback)
    const unsigned dlo = bp > MEDREJ_DWIN + 1u ? bp - MEDREJ_DWIN : 1u;
    const unsigned dhi = bp + MEDREJ_DWIN < 2046u ? bp + MEDREJ_DWIN : 2046u;
    const unsigned long long kB64 = (unsigned long long)lo + ((unsigned long long)dhi << S) - 1ull;
    const unsigned kA = lo + ((dlo - 1u) << S), kB = kB64 > 0x7F7FFFFFull ? 0x7F7FFFFFu : (unsigned)kB64;
    const double thrA = (double)__uint_as_float(kA) * scale, thrB = (double)__uint_as_float(kB) * scale;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SEL_BINS / 256; u++) hist[u * 256 + tid] = 0;
    __syncthreads();

    // ---- the pass: histogram + window keys (unflagged samples), decided flags + the undecided list (all samples) ----
    // No branch and no atomic per sample beyond the histogram's: the window keys and the undecided indices go to PER-THREAD lists
    // (entry j of thread t at [j * 256 + t]; the thread that wrote a list is the one that reads it back), flagged samples
    // count into a dummy bin.
    const unsigned capt = cand_cap / 256u, ucapt = und_cap / 512u;        // entries per thread (undecided: index + value)
    unsigned mb1 = 0, ccnt = 0, ucnt = 0;
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    tile_walk([&](bool valid, unsigned i, const float4 rv, unsigned f) -> unsigned {
        const float xv[4] = {rv.x, rv.y, rv.z, rv.w};
        unsigned fn = f;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            const unsigned k = __float_as_uint(xv[k4]) & 0x7FFFFFFFu;
            const bool unfl = valid && ((f >> (8 * k4)) & 0xFFu) == 0u;
            const unsigned b = bin_of(k);
            if (!(MEDREJ_ABLATE & 1)) atomicAdd(&hist[unfl ? b : SEL_BINS + (unsigned)lane], 1u);
            const bool inwin = unfl && (b - wlo <= wspan);
            if (!(MEDREJ_ABLATE & 2))
                __builtin_amdgcn_raw_buffer_store_b32(k, crs, (int)((inwin && ccnt < capt) ? ccnt * 1024u + (unsigned)tid * 4u : OOBR), 0, 0);
            ccnt += inwin ? 1u : 0u;
            mb1 = max(mb1, (unfl && b < wlo) ? k + 1u : 0u);
            const double dx = (double)xv[k4];
            const bool gtB = !(MEDREJ_ABLATE & 4) && dx > thrB, gtA = !(MEDREJ_ABLATE & 4) && dx > thrA;   // (a NaN compares false twice: never flagged)
            fn |= gtB ? (1u << (8 * k4)) : 0u;
            const bool und = valid && gtA && !gtB;
            if (!(MEDREJ_ABLATE & 2)) {
                u2v ev;
                ev[0] = i * 4u + (unsigned)k4;
                ev[1] = __float_as_uint(xv[k4]);
                __builtin_amdgcn_raw_buffer_store_b64(ev, urs, (int)((und && ucnt < ucapt) ? ucnt * 2048u + (unsigned)tid * 8u : OOBR), 0, 0);
            }
            ucnt += und ? 1u : 0u;
        }
        return fn;
    });
    mb1 = wave_max_u32(mb1);
    if (lane == 0 && mb1) atomicMax(&sh_below1, mb1);
    if (ccnt > capt || ucnt > ucapt) sh_over = 1;
    __threadfence_block();
    __syncthreads();
    const unsigned total = locate(wlo);
    const unsigned excw = sh_excw, below1 = sh_below1, bsel = sh_bin;
    if (!(total > 0 && bsel >= wlo && bsel <= whi && !sh_over)) {
        if (tid == 0) atomicAdd(&g_medrej_stats[2], 1ull);
        fallback();
        return;
    }
    __syncthreads();   // sh[] is reused by select3 below
    auto enumerate_gc = [&](auto&& visit) {                            // this thread's own window keys (eight loads in flight)
        unsigned j = 0;
        for (; j + 8 <= ccnt; j += 8) {
            unsigned q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = gcand[(size_t)(j + u) * 256 + tid];
#pragma unroll
            for (int u = 0; u < 8; u++) visit(q[u]);
        }
        for (; j < ccnt; j++) visit(gcand[(size_t)j * 256 + tid]);
    };
    Sel3State st = select3(hist, sh, enumerate_gc, (long long)((total >> 1) - excw));
    st.n = total;
    const double m = median_of(st, below1);
    const double thr = m * scale;
    // (an even count's lower middle element may lie below the window: then the median may, too -- the bracket must hold)
    if (!(thr >= thrA && thr <= thrB)) { if (tid == 0) atomicAdd(&g_medrej_stats[3], 1ull); fallback(); return; }
    if (tid == 0) med[oidx] = m;
    // ---- the undecided samples against the exact threshold (every thread its own list) ----
    uint8_t* fo8 = flags_out + win * ws_flags;
    uint8_t* ft8 = flags_t4 + win * ws_flags;
    const unsigned T = (unsigned)C4 * 4u;
    for (unsigned j = 0; j < ucnt; j += 4) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = j + u < ucnt ? reinterpret_cast<const uint2*>(gund)[(size_t)(j + u) * 256 + tid] : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j + u < ucnt && (double)__uint_as_float(e[u].y) > thr) {
                const unsigned l = e[u].x / T, t = e[u].x - l * T;
                fo8[e[u].x] = 1;
                ft8[((size_t)(t >> 2) * L + l) * 4 + (t & 3u)] = 1;
            }
        }
    }
}
