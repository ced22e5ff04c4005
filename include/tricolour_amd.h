/*
 * tricolour_amd.h -- C ABI of the MI355X-native SumThreshold flagger.
 *
 * This is the drop-in boundary for ONE path of ratt-ru/tricolour: the
 * per-block callable behind tricolour.dask_wrappers.sum_threshold_flagger
 * (reference tricolour/dask_wrappers.py:23-46), i.e.
 * tricolour.flagging.sum_threshold_flagger (reference
 * tricolour/flagging.py:1076-1196) and the window pack / unpack transposes
 * either side of it (reference tricolour/packing.py:243-278, 369-415).
 *
 * Plain pointers and sizes only; no torch / Python types.  All data pointers
 * are DEVICE pointers (HBM resident); `stream` is a hipStream_t (NULL = the
 * null stream).  Functions enqueue work on `stream` and return without
 * synchronising; inputs are never written; outputs and the workspace are
 * caller-allocated.  Every function returns 0 on success or a TRI_E* code, in
 * which case tri_last_error() (thread-local) describes the failure.  The
 * library is re-entrant: it keeps no mutable global state, so concurrent calls
 * from several host threads (the reference's dask ThreadPool,
 * apps/tricolour/app.py:266-271) are safe when each uses its own stream and
 * workspace.
 */
#ifndef TRICOLOUR_AMD_H
#define TRICOLOUR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRI_MAX_WINDOWS 16

enum tri_status {
    TRI_OK = 0,
    TRI_EINVAL = 1,      /* bad shape / NULL pointer -> Python ValueError       */
    TRI_EUNSUPPORTED = 2,/* parameter combination the reference itself rejects
                            or that this build does not implement               */
    TRI_EWORKSPACE = 3,  /* workspace smaller than tri_workspace_bytes(.., 1)   */
    TRI_EHIP = 4         /* a HIP runtime call failed                           */
};

enum tri_vis_dtype {
    TRI_VIS_C64 = 0,     /* interleaved (re, im) float32 -- MS DATA columns     */
    TRI_VIS_F32 = 1,     /* real float32 amplitudes (flagging.py:830-832)       */
    TRI_VIS_C128 = 2,    /* interleaved float64: tri_stokes_intensity only       */
    TRI_VIS_F64 = 3      /* real float64 AMPLITUDES: what float64 / complex128 visibilities are to
                            _average_freq (np.abs in float64, flagging.py:856), added to the float32
                            accumulator in float64 and rounded at every step (:858-859).  The caller forms
                            them (|x|, hypot(re, im)); a visibility with a NaN part must arrive as NaN
                            (the final isnan(in_data), :777-781).  tri_sum_threshold_flagger only.      */
};

/*
 * Prepared parameters of one sum_threshold_flagger call: the keyword
 * arguments of flagging.py:1076-1083 after the plain-Python preparation of
 * flagging.py:1156-1179 (window de-duplication / clipping, frequency chunk
 * ends).  Build it with tri_prepare_params().
 */
typedef struct tri_params {
    double  outlier_nsigma;
    int64_t n_windows_time;
    int64_t windows_time[TRI_MAX_WINDOWS];
    int64_t n_windows_freq;
    int64_t windows_freq[TRI_MAX_WINDOWS];
    double  background_reject;
    int64_t background_iterations;
    double  spike_width_time;
    double  spike_width_freq;
    int64_t time_extend;
    int64_t freq_extend;
    int64_t n_chunk_ends;          /* freq_chunks + 1                          */
    const int64_t *chunk_ends;     /* HOST pointer, averaged-channel units     */
    int64_t average_freq;
    double  flag_all_time_frac;
    double  flag_all_freq_frac;
    double  rho;
    int64_t num_major_iterations;
} tri_params;

/*
 * Restates flagging.py:1156-1179: windows_freq = unique(int(ceil(f32(w)) /
 * average_freq)); freq_chunk_ends = linspace(0, averaged_channels,
 * freq_chunks + 1).astype(int); windows clipped to <= ntime / <=
 * averaged_channels.  `chunk_ends_buf` (HOST, capacity `chunk_cap` >=
 * freq_chunks + 1) receives the chunk ends and is referenced by `out`.
 * A zero-sized frequency window (possible with average_freq > 1, where the
 * reference fails with a broadcasting ValueError at flagging.py:663) returns
 * TRI_EINVAL.
 */
int tri_prepare_params(int64_t ntime, int64_t nchan,
                       double outlier_nsigma,
                       const double *windows_time, int64_t n_windows_time,
                       const double *windows_freq, int64_t n_windows_freq,
                       double background_reject, int64_t background_iterations,
                       double spike_width_time, double spike_width_freq,
                       int64_t time_extend, int64_t freq_extend,
                       int64_t freq_chunks, int64_t average_freq,
                       double flag_all_time_frac, double flag_all_freq_frac,
                       double rho, int64_t num_major_iterations,
                       int64_t *chunk_ends_buf, int64_t chunk_cap,
                       tri_params *out);

/*
 * Device workspace needed to process `batch_windows` correlation products
 * (bl x corr windows) of shape (ntime, nchan) at a time.  The flagger accepts
 * any workspace >= tri_workspace_bytes(.., 1) and sizes its internal batch to
 * what it is given; larger batches fill the GPU better.
 */
size_t tri_workspace_bytes(int64_t batch_windows, int64_t ntime, int64_t nchan,
                           const tri_params *p);

/*
 * Replaces tricolour.flagging.sum_threshold_flagger (flagging.py:1076-1196)
 * for one window block.
 *   vis        (n_cp, ntime, nchan) visibilities, C order, dtype `vis_dtype`
 *              [n_cp = bl * corr after the reshape of flagging.py:1156-1158]
 *   flags      (n_cp, ntime, nchan) uint8 / bool, non-zero = flagged
 *   out_flags  (n_cp, ntime, nchan) uint8, receives 0/1: the LAST major
 *              iteration's flags (flagging.py:1181-1196), not OR-ed with the
 *              input flags
 * Bit-exact with the reference executed under its numba JIT.
 */
int tri_sum_threshold_flagger(const void *vis, int vis_dtype,
                              const uint8_t *flags, uint8_t *out_flags,
                              int64_t n_cp, int64_t ntime, int64_t nchan,
                              const tri_params *p,
                              void *workspace, size_t workspace_bytes,
                              void *stream);

/*
 * Replaces packing._numba_pack_data (packing.py:243-278): scatter MS rows
 * (row, chan, corr) into windows (bl, corr, time, chan).  `row_bl` / `row_time`
 * (int32, length `rows`) give each row's window baseline index and time index
 * (row_bl < 0: row belongs to no baseline of this block and is skipped); they
 * replace the reference's O(nbl * rows) antenna matching with a precomputed
 * row -> (bl, t) map (see tri_row_map in the Python mirror).  Cells no row
 * maps to keep their prior contents: initialise the windows with
 * tri_fill_windows (NaN+NaNj / 1, packing.py:97,117).
 */
int tri_pack_data(const void *data_c64, const uint8_t *flag,
                  const int32_t *row_bl, const int32_t *row_time,
                  int64_t rows, int64_t nchan, int64_t ncorr,
                  int64_t nbl, int64_t ntime,
                  void *vis_windows_c64, uint8_t *flag_windows, void *stream);

int tri_fill_windows(void *vis_windows_c64, uint8_t *flag_windows,
                     int64_t n_elements, void *stream);

/*
 * Replaces packing._unpack_data / _numpy_unpack_transpose
 * (packing.py:369-415): gather flag windows (bl, corr, time, chan) back to MS
 * row order (row, chan, corr); rows with row_bl < 0 are set to 0.  With
 * any_corr != 0 every correlation of a (row, chan) cell receives the OR over
 * its correlations -- the equalisation the application applies right after
 * unpacking (apps/tricolour/app.py:479-480).
 */
int tri_unpack_data(const uint8_t *flag_windows,
                    const int32_t *row_bl, const int32_t *row_time,
                    int64_t rows, int64_t nchan, int64_t ncorr,
                    int64_t nbl, int64_t ntime,
                    uint8_t *out_flags, int any_corr, void *stream);

/*
 * Replaces tricolour.stokes.polarised_intensity (stokes.py:157-209, mode 0)
 * and unpolarised_intensity (stokes.py:79-153, mode 1) on (row, chan, corr)
 * visibilities, n = row * chan samples:
 *   value_k = a_k * (s1_k * vis[c1_k] + s2_k * vis[c2_k])     in complex128
 *   mode 0: out = sqrt(sum_pol |value|^2)
 *   mode 1: out = sum_unpol |value| - sqrt(sum_pol |value|^2)
 * written as one correlation of the visibility dtype (imaginary part 0).
 * Term tables are HOST arrays: idx = (c1, c2, s1, s2) per term, alpha =
 * (re, im) per term (the (c1, c2, a, s1, s2) tuples of stokes_corr_map).
 * vis_dtype: TRI_VIS_C64 or TRI_VIS_C128.
 */
int tri_stokes_intensity(const void *vis, int vis_dtype, int64_t n, int64_t ncorr,
                         const int32_t *pol_idx, const double *pol_alpha, int64_t n_pol,
                         const int32_t *unpol_idx, const double *unpol_alpha, int64_t n_unpol,
                         int mode, void *out, void *stream);

/*
 * Flag counts behind tricolour.window_statistics._window_stats
 * (window_statistics.py:12-66): one pass over a (bl, corr, time, chan) uint8
 * flag window gives
 *   per_bl[bl]     number of set flags of baseline bl     (per-antenna,
 *                  per-baseline, per-scan and per-field sums, :27-53)
 *   per_chan[chan] number of set flags of channel chan    (per-ddid frequency
 *                  bins, :55-66)
 * Both outputs are device arrays of uint64 and are zeroed by the call.
 */
int tri_window_counts(const uint8_t *flags, int64_t nbl, int64_t ncorr,
                      int64_t ntime, int64_t nchan, uint64_t *per_bl,
                      uint64_t *per_chan, void *stream);

/*
 * Replaces tricolour.flagging.flag_nans_and_zeros (flagging.py:29-62):
 * out = (vis == 0) | isnan(vis) | (flags != 0), elementwise over n samples.
 */
int tri_flag_nans_and_zeros(const void *vis, int vis_dtype, const uint8_t *flags,
                            uint8_t *out_flags, int64_t n, void *stream);

/*
 * Device half of tricolour.flagging.apply_static_mask (flagging.py:151-172,
 * one call per mask) and flag_autos (flagging.py:90-93): out = flags, then on
 * every baseline with bl_sel[bl] != 0 either out |= chan_mask (mode 0, "or")
 * or out = chan_mask (mode 1, "override"), broadcast over corr and time.
 * flags / out: (nbl, ncorr, ntime, nchan) uint8; may alias.  The channel mask
 * and the baseline selection (uv-range test on antenna positions) are
 * computed on the host exactly as flagging.py:131-160 does.
 */
int tri_apply_baseline_channel_mask(const uint8_t *flags, uint8_t *out_flags,
                                    const uint8_t *bl_sel, const uint8_t *chan_mask,
                                    int mode, int64_t nbl, int64_t ncorr, int64_t ntime,
                                    int64_t nchan, void *stream);

/*
 * Replaces tricolour.flagging.uvcontsub_flagger (flagging.py:989-1073): per
 * correlation product and major cycle, the time-mean spectrum of the unflagged
 * visibilities is low-passed (first `taylor_degrees` Fourier bins), samples
 * whose |vis - smooth| exceeds sigma x the MAD-of-MAD of the unflagged
 * residuals are flagged; cycles below `or_original_from_cycle` REPLACE the
 * flags, later ones OR them in.  vis complex64 (n_cp, ntime, nchan); out_flags
 * receives 0/1.  The reference is plain NumPy whose FFT / residual precision
 * depends on the NumPy version (float32 under NumPy >= 2, followed here), so
 * parity is by flag-agreement rate, not bit for bit.
 */
size_t tri_uvcontsub_workspace_bytes(int64_t batch_windows, int64_t ntime, int64_t nchan);
int tri_uvcontsub_flagger(const void *vis_c64, const uint8_t *flags, uint8_t *out_flags,
                          int64_t n_cp, int64_t ntime, int64_t nchan,
                          int64_t major_cycles, int64_t or_original_from_cycle,
                          int64_t taylor_degrees, double sigma,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Thread-local description of the last failure in the calling thread. */
const char *tri_last_error(void);

/* Library / ABI version (major * 100 + minor). */
int tri_version(void);

/*
 * Measurement hooks (used by bench.py and tests only; not part of the
 * reference's interface).
 *
 * tri_bench_sumthreshold runs ONLY the fused SumThreshold column kernel
 * (clamp + float64 prefix + multi-scale thresholds + flag dilation, all
 * windows in one pass; flagging.py:582-681) `repeats` times on `stream`,
 * bracketed by HIP events recorded on that stream, and returns the mean kernel
 * time in milliseconds through `ms_per_launch`.
 *   data (n_win, n_line, n_col) float32 residuals; the sequential axis is
 *        n_line, columns are coalesced
 *   mad  (n_win, n_col) float64 medians of |data| (NaN = nothing unflagged)
 *   out  (n_win, n_line, n_col) uint8
 * `variant`: 0 = best available for these windows, 1 = generic (dynamic
 * windows, global rings), 2 = register cascade (windows 1,2,4,8 only),
 * 3 = lane-mask cascade (windows 1,2,4,8, window below 2^31 bytes; what 0
 * selects for those), 4 = stage pipeline across the waves of a workgroup with
 * the prefix rings in LDS (K7p; any list of up to eight windows whose rings
 * fit 160 KB, e.g. 32, 48, 64, 128 -- the flagger's route for such lists),
 * 5 = the lane-mask cascade on COLUMN PANELS [n_col / 64][n_line][64] (what the
 * flagger's time-axis pass runs, and what 0 selects when n_col % 64 == 0): the
 * row images are re-laid out before the timed region, the flags taken back to
 * rows after it.
 */
int tri_bench_sumthreshold(const float *data, const double *mad, uint8_t *out,
                           int64_t n_win, int64_t n_line, int64_t n_col,
                           const int64_t *windows, int64_t n_windows,
                           double outlier_nsigma, double rho, int variant,
                           int repeats, float *ms_per_launch, void *stream);

/*
 * tri_bench_boxfilter runs ONE axis stage of masked_gaussian_filter's two box
 * filters (flagging.py:362-419, 469-513) `repeats` times on `stream` in the
 * launch geometry the flagger itself uses, bracketed by HIP events on that
 * stream; mean time per stage in `ms_per_launch`.
 *   stage 0: time-axis stage.  data (n_win, n_line, n_col) float32, line axis
 *            = time, columns coalesced; flags4 the same window's flags packed
 *            four line positions per 32-bit word, (n_win, n_line / 4, n_col)
 *            words; out_w / out_o (n_win, n_line, n_col) receive the filtered
 *            weight (!flag) and weight * data images, already divided by
 *            float32(2 r + 1) ** 4.
 *   stage 1: frequency-axis stage fused with the masked division of the
 *            rejection loop (flagging.py:506-513, 563-566).  `flags4` is a float32
 *            array (n_win, 2, n_line, n_col): per window the time-filtered weight
 *            image followed by the weight * data image, lines contiguous (n_col
 *            positions each); `data` the amplitudes (n_win, n_col, n_line);
 *            out_o (n_win, n_col, n_line) receives |data - background|, out_w
 *            is scratch of the same size.
 *   stage 2: the 1-D filter of the spectrum path (flagging.py:945-947 via
 *            :516-579): n_win = 1, data (n_line, n_col) float32 with the line
 *            axis = channel and one column per window of a batch, flags4 one
 *            BYTE per sample (n_line, n_col); out_w / out_o (n_line, n_col)
 *            receive the filtered weight and weight * data images, divided by
 *            float32(2 r + 1) ** 4.
 * `variant`: 0 = the flagger's default route for this radius, 1 = LDS delay
 * lines only (K4b / K4c / multi-pass), 2 = register delay lines (K4r).
 * For stage 2: 0 = default route, 1 = register delay lines (K4r), 2 / 3 = the
 * stage pipeline across four waves (K4p) with blocks of 16 / 8 positions
 * (TRI_EUNSUPPORTED when that block length does not apply to the shape).
 * For stage 1 also: 3 = the eight-wave stage pipeline (K4qf), 4 = the exact
 * row filter (K4x: one line pair resident in LDS, lanes = positions, any
 * radius; the time covers the kernel and the transpose of its rows to out_o).
 * Stages 0 and 1: 3 = the stage pipeline (K4q / K4qf) with the block length the
 * flagger would take (8 positions while the delay line fits 128 registers,
 * else 16), 5 = the same with blocks of 16 positions throughout.
 */
int tri_bench_boxfilter(const float *data, const uint8_t *flags4, float *out_w,
                        float *out_o, int64_t n_win, int64_t n_line, int64_t n_col,
                        int64_t radius, int stage, int variant, int repeats,
                        float *ms_per_launch, void *stream);

/*
 * Test hook: how many line passes the last tri_bench_boxfilter(stage 1,
 * variant 4) call of this thread ran, and how many of them failed the exactness
 * check and were redone in the reference's sequential order
 * (kernels_boxexact.hpp; flagging.py:398-416).
 */
int tri_boxx_last_stats(uint64_t *passes, uint64_t *sequential);

/*
 * Test hook: counters of the one-pass block-median + rejection kernels of the background loop (kernels_reject.hpp,
 * kernels_reject_tile.hpp) since the last reset -- out20[0..2] = blocks the one-workgroup form (K3r) ran, fell back
 * before / after its pass; [3] = second rounds of the tile-parallel form (K3t: median outside the predicted window or
 * the decision bracket); [4 + r] = blocks K3t handed to the one-workgroup redo for reason r (1 nothing to predict
 * from, 2 / 10 a wave's window-key / undecided list overflowed, 3 empty, 4 median in an end bin, 5 a block list
 * overflowed, 6 / 7 window / bracket miss in round 2).  Synchronises the device.  No reference counterpart.
 */
int tri_medrej_stats(uint64_t *out20, int reset);

/*
 * Measurement hook: per-thread kernel log.  op 0 clears the log and switches it on; op 1 writes
 * "kernel=launches;kernel=launches;..." (demangled device kernel symbols launched by this thread since op 0)
 * into buf[cap] and switches the log off; op 2 switches it off.  bench.py names the kernels of its roofline
 * legs with it and derives the step's dominant kernel family from the launch counts.  Returns the number of
 * distinct kernels (op 1).  No reference counterpart.
 */
int tri_kernel_log(int op, char *buf, int64_t cap);

/*
 * Test hook: tri_sum_threshold_flagger that additionally taps the LAST major
 * iteration's intermediates of window 0 into caller-provided device buffers
 * (Fa = averaged channels, N = ntime * Fa):
 *   dbg_f32: [spec_resid (Fa)][background, FT layout (Fa x ntime)][residual, TF layout (ntime x Fa)]
 *   dbg_u8 : [spec_flags (Fa)][time_flags TF (N)][freq_flags TF (N)]
 */
int tri_sum_threshold_flagger_debug(const void *vis, int vis_dtype,
                                    const uint8_t *flags, uint8_t *out_flags,
                                    int64_t n_cp, int64_t ntime, int64_t nchan,
                                    const tri_params *p,
                                    void *workspace, size_t workspace_bytes,
                                    void *stream, float *dbg_f32, uint8_t *dbg_u8);

/*
 * Measurement / test hook: ONE rejection step of the background loop,
 *     flags |= resid > median_abs(resid[~flags]) * 1.4826 * reject     per (window, chunk) block
 * (flagging.py:553-574 with _median_abs :267), by the one-pass route the flagger
 * takes for blocks of at least 65536 samples (k_mr_predict / k_mr_pass /
 * k_mr_finish + the redo kernel; TRI_EUNSUPPORTED for smaller blocks).
 * resid (n_win, n_chan, n_time) float32 and flags_in (same shape, bytes) are the
 * channel-major images of the loop; flags_out (same shape) receives the new
 * flags, flags_t4 (n_win, n_time / 4, n_chan, 4) the same flags with four
 * consecutive times of a channel per 32-bit word (the time-axis filter's input),
 * med (n_win, n_chunk_ends - 1) float64 the block medians (NaN: nothing
 * unflagged).  chunk_ends is a HOST array of channel boundaries, 0 ... n_chan.
 * ms_per_step: mean time of the whole step over `repeats`, HIP events on `stream`.
 */
int tri_bench_reject(const float *resid, const uint8_t *flags_in, uint8_t *flags_out,
                     uint8_t *flags_t4, double *med, int64_t n_win, int64_t n_chan,
                     int64_t n_time, const int64_t *chunk_ends, int64_t n_chunk_ends,
                     double reject, int repeats, float *ms_per_step, void *stream);

/*
 * Test hook: exact segmented medians of |x| over the unflagged samples of a
 * (n_win, rows, row_len) float32 array; segments [seg_ends[g], seg_ends[g+1])
 * (HOST array) along each row; med is (n_win, rows, n_seg_ends - 1) float64,
 * NaN where a segment has no unflagged sample.  variant 0 = automatic,
 * 1 = wave kernel (4: its 16-byte-load form over masked groups), 2 / 3 = three-pass workgroup kernel with scalar / vector loads,
 * 5 / 6 = two-pass workgroup kernel with vector / scalar loads, 7 = two-pass
 * kernel, vector loads over segments that need not be 4-aligned (row_len % 4 == 0),
 * 8 = multi-workgroup two-pass select (one segment spanning the row),
 * 9 / 10 = the two-pass select with the predicted-bin candidate window in
 * global scratch (one read of the segment when the prediction holds;
 * 9 vector loads, 10 scalar).  1 / 4 run the wave kernels with several rows of a
 * segment per wave (the flagger's launch shape), 11 / 14 with one segment per
 * wave (TRI_MEDIAN_WAVE_OLD=1 in the flagger).
 * Restates _median_abs / _median_abs_axis0 (flagging.py:267-304).
 */
int tri_test_median(const float *data, const uint8_t *flags, double *med,
                    int64_t n_win, int64_t rows, int64_t row_len,
                    const int64_t *seg_ends, int64_t n_seg_ends, int variant,
                    void *stream);

/*
 * Test hook: the register-ring box-filter kernels divide by the launch constant
 * float32(2 r + 1) ** 4 (flagging.py:419) through its reciprocal with exact
 * remainder corrections.  Runs that division on ALL 2^32 float32 bit patterns
 * for one radius and counts the inputs whose result differs from the correctly
 * rounded IEEE quotient (NaN results compare equal); must be 0.
 */
int tri_test_box_divide(int64_t radius, uint64_t *mismatches, void *stream);

/* Device amplitude of complex64 samples, the |z| used at flagging.py:856
 * (libm hypotf semantics); exposed so the tests can pin it (golden G0). */
int tri_abs_c64(const void *z_c64, float *out, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRICOLOUR_AMD_H */
