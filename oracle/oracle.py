"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module (and only as the checker /
reported baseline).  The product package ``tricolour_amd`` never does.

The numerics live in ``tricolour_oracle.c``; this file restates the plain
Python parameter preparation of ``tricolour/flagging.py:1156-1179``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtricolour_oracle.so")

POW_SQMUL, POW_POWF = 0, 1
INTERP_F64, INTERP_F32 = 0, 1
MAX_WINDOWS = 16


class TroParams(C.Structure):
    _fields_ = [
        ("outlier_nsigma", C.c_double),
        ("n_windows_time", C.c_int64),
        ("windows_time", C.c_int64 * MAX_WINDOWS),
        ("n_windows_freq", C.c_int64),
        ("windows_freq", C.c_int64 * MAX_WINDOWS),
        ("background_reject", C.c_double),
        ("background_iterations", C.c_int64),
        ("spike_width_time", C.c_double),
        ("spike_width_freq", C.c_double),
        ("time_extend", C.c_int64),
        ("freq_extend", C.c_int64),
        ("n_chunk_ends", C.c_int64),
        ("chunk_ends", C.POINTER(C.c_int64)),
        ("average_freq", C.c_int64),
        ("flag_all_time_frac", C.c_double),
        ("flag_all_freq_frac", C.c_double),
        ("rho", C.c_double),
        ("num_major_iterations", C.c_int64),
    ]


def build(force=False):
    src = os.path.join(_HERE, "tricolour_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libtricolour_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.tro_median_abs.restype = C.c_double
        _lib.tro_box_denominator.restype = C.c_float
        _lib.tro_box_radius.restype = C.c_int64
        _lib.tro_abs_c64.restype = C.c_float
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _i64(v):
    return C.c_int64(int(v))


def set_modes(pow_mode=POW_SQMUL, interp_mode=INTERP_F64):
    lib().tro_set_modes(C.c_int(pow_mode), C.c_int(interp_mode))


def prepare(nchan, ntime, windows_time, windows_freq, freq_chunks, average_freq):
    """tricolour/flagging.py:1160-1179, plain Python + NumPy."""
    wf = np.asarray(windows_freq, dtype=np.float32)
    wf = np.ceil(wf) / np.float32(average_freq)
    wf = np.unique(wf.astype(np.int64))
    fa = (int(nchan) + int(average_freq) - 1) // int(average_freq)
    chunk_ends = np.linspace(0, fa, freq_chunks + 1).astype(np.int64)
    wt = np.array([int(w) for w in windows_time if w <= ntime], np.int64)
    wf = np.array([int(w) for w in wf if w <= fa], np.int64)
    return wt, wf, chunk_ends, fa


DEFAULTS = dict(outlier_nsigma=4.5, windows_time=[1, 2, 4, 8],
                windows_freq=[1, 2, 4, 8], background_reject=2.0,
                background_iterations=1, spike_width_time=12.5,
                spike_width_freq=10.0, time_extend=3, freq_extend=3,
                freq_chunks=10, average_freq=1, flag_all_time_frac=0.6,
                flag_all_freq_frac=0.8, rho=1.3, num_major_iterations=5)


def make_params(ntime, nchan, **kw):
    k = dict(DEFAULTS)
    unknown = set(kw) - set(k)
    if unknown:
        raise TypeError("unexpected kwargs %s" % sorted(unknown))
    k.update(kw)
    wt, wf, ce, fa = prepare(nchan, ntime, k["windows_time"], k["windows_freq"],
                             k["freq_chunks"], k["average_freq"])
    if len(wt) > MAX_WINDOWS or len(wf) > MAX_WINDOWS:
        raise ValueError("too many windows")
    if int(k["num_major_iterations"]) > 0:
        # what the reference's _sum_threshold does with such lists (flagging.py:630, 663)
        if len(wt) == 0 or len(wf) == 0:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
        if (wt <= 0).any() or (wf <= 0).any():
            raise ValueError("operands could not be broadcast together (window of size 0)")
    p = TroParams()
    p.outlier_nsigma = float(k["outlier_nsigma"])
    p.n_windows_time = len(wt)
    for i, w in enumerate(wt):
        p.windows_time[i] = int(w)
    p.n_windows_freq = len(wf)
    for i, w in enumerate(wf):
        p.windows_freq[i] = int(w)
    p.background_reject = float(k["background_reject"])
    p.background_iterations = int(k["background_iterations"])
    p.spike_width_time = float(k["spike_width_time"])
    p.spike_width_freq = float(k["spike_width_freq"])
    p.time_extend = int(k["time_extend"])
    p.freq_extend = int(k["freq_extend"])
    ce = np.ascontiguousarray(ce, np.int64)
    p.n_chunk_ends = len(ce)
    p.chunk_ends = _p(ce, C.POINTER(C.c_int64))
    p.average_freq = int(k["average_freq"])
    p.flag_all_time_frac = float(k["flag_all_time_frac"])
    p.flag_all_freq_frac = float(k["flag_all_freq_frac"])
    p.rho = float(k["rho"])
    p.num_major_iterations = int(k["num_major_iterations"])
    p._keep = ce  # keep the chunk array alive
    return p, fa


def sum_threshold_flagger(vis, flags, n_threads=1, dump=False, **kw):
    """Oracle twin of ``tricolour.flagging.sum_threshold_flagger``
    (flagging.py:1076-1196).  ``vis``: (bl,corr,time,chan) complex64 or
    float32; ``flags``: same shape, any 1-byte type.  Returns bool array; with
    ``dump=True`` also a dict of last-iteration intermediates of cp 0."""
    vis = np.ascontiguousarray(vis)
    if vis.shape != flags.shape:
        raise ValueError("shape mismatch")
    nbl, ncorr, T, F = vis.shape
    flags8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    p, fa = make_params(T, F, **kw)
    out = np.zeros(vis.shape, np.uint8)
    c64 = f32 = c128 = f64 = None
    if vis.dtype == np.complex64:
        c64 = vis
    elif vis.dtype == np.float32:
        f32 = vis
    elif vis.dtype == np.complex128:
        c128 = vis
    elif vis.dtype == np.float64:
        f64 = vis
    else:
        raise TypeError("oracle handles real / complex float32 / float64 visibilities")
    dptr = fptr = None
    d = {}
    if dump:
        d = dict(spec_resid=np.zeros(fa, np.float32),
                 background=np.zeros((T, fa), np.float32),
                 residual=np.zeros((T, fa), np.float32),
                 spec_flags=np.zeros(fa, np.uint8),
                 time_flags=np.zeros((T, fa), np.uint8),
                 freq_flags=np.zeros((T, fa), np.uint8))
        dptr = (C.c_void_p * 3)(_p(d["spec_resid"]), _p(d["background"]), _p(d["residual"]))
        fptr = (C.c_void_p * 3)(_p(d["spec_flags"]), _p(d["time_flags"]), _p(d["freq_flags"]))
    lib().tro_sum_threshold_flagger_any(_p(c64), _p(f32), _p(c128), _p(f64), _p(flags8), _p(out),
                                        _i64(nbl * ncorr), _i64(T), _i64(F),
                                        C.byref(p), C.c_int(n_threads), dptr, fptr)
    out = out.view(np.bool_)
    return (out, d) if dump else out


# ---- helper-level entry points (used by the KAT tests) -------------------

def abs_c64(z):
    z = np.ascontiguousarray(z, np.complex64)
    out = np.empty(z.shape, np.float32)
    lib().tro_abs_c64_array(_p(z), _i64(z.size), _p(out))
    return out


def average_freq(data, flags, factor):
    data = np.ascontiguousarray(data)
    n_cp, T, F = data.shape
    fa = (F + factor - 1) // factor
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    od = np.zeros((n_cp, T, fa), np.float32)
    of = np.zeros((n_cp, T, fa), np.uint8)
    c64 = data if data.dtype == np.complex64 else None
    f32 = data.astype(np.float32) if c64 is None else None
    lib().tro_average_freq(_p(c64), _p(f32), _p(f8), _i64(n_cp), _i64(T),
                           _i64(F), _i64(factor), _p(od), _p(of))
    return od, of.view(np.bool_)


def time_median(data, flags):
    data = np.ascontiguousarray(data, np.float32)
    T, F = data.shape
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    od = np.zeros((1, F), np.float32)
    of = np.zeros((1, F), np.uint8)
    lib().tro_time_median(_p(data), _p(f8), _i64(T), _i64(F), _p(od), _p(of))
    return od, of.view(np.bool_)


def median_abs(data, flags):
    data = np.ascontiguousarray(data, np.float32)
    data2 = data.reshape(data.shape[0], -1) if data.ndim > 1 else data.reshape(1, -1)
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    T, F = data2.shape
    return float(lib().tro_median_abs(_p(data2), _p(f8), _i64(T), _i64(F), _i64(0), _i64(F)))


def median_abs_axis0(data, flags):
    data = np.ascontiguousarray(data, np.float32)
    n0 = data.shape[0]
    n1 = data.size // max(n0, 1)
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    out = np.zeros((1,) + data.shape[1:], np.float32)
    lib().tro_median_abs_axis0(_p(data), _p(f8), _i64(n0), _i64(n1), _p(out))
    return out


def linearly_interpolate_nans1d(y):
    """In place, float32 only."""
    assert y.dtype == np.float32 and y.flags.c_contiguous
    lib().tro_interpolate_nans1d(_p(y), _i64(y.size))


def box_radius(sigma, passes=4):
    return int(lib().tro_box_radius(C.c_double(float(sigma)), _i64(passes)))


def box_gaussian_filter(data, sigma, passes=4):
    data = np.ascontiguousarray(data, np.float32)
    T, F = data.shape
    out = np.empty_like(data)
    r0, r1 = box_radius(sigma[0], passes), box_radius(sigma[1], passes)
    lib().tro_box_gaussian_filter(_p(data), _p(out), _i64(T), _i64(F), _i64(r0),
                                  _i64(r1), _i64(passes))
    return out


def box_gaussian_filter1d(a, r, passes):
    a = np.ascontiguousarray(a, np.float32).reshape(1, -1)
    out = np.empty_like(a)
    if r > 0:
        lib().tro_box_gaussian_filter(_p(a), _p(out), _i64(1), _i64(a.size),
                                      _i64(0), _i64(r), _i64(passes))
    else:
        out[:] = a
    return out.reshape(-1)


def masked_gaussian_filter(data, flags, sigma):
    data = np.ascontiguousarray(data, np.float32)
    T, F = data.shape
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    out = np.empty_like(data)
    w = np.empty_like(data)
    lib().tro_masked_gaussian_filter(_p(data), _p(f8), _i64(T), _i64(F),
                                     _i64(box_radius(sigma[0])),
                                     _i64(box_radius(sigma[1])), _p(out), _p(w))
    return out


def get_background2d(data, flags, iterations, spike_width, reject_threshold,
                     freq_chunk_ends):
    data = np.ascontiguousarray(data, np.float32)
    T, F = data.shape
    f8 = np.ascontiguousarray(np.asarray(flags) != 0).view(np.uint8)
    ce = np.ascontiguousarray(freq_chunk_ends, np.int64)
    out = np.empty_like(data)
    lib().tro_get_background2d(_p(data), _p(f8), _i64(T), _i64(F),
                               _i64(iterations), C.c_double(float(spike_width[0])),
                               C.c_double(float(spike_width[1])),
                               C.c_double(float(reject_threshold)), _p(ce),
                               _i64(len(ce)), _p(out))
    return out


def sum_threshold(data, flags, axis, windows, outlier_nsigma, rho, chunks=None):
    data = np.ascontiguousarray(data, np.float32)
    n0, n1 = data.shape
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    w = np.ascontiguousarray(windows, np.int64)
    if chunks is None:
        chunks = np.array([0, data.shape[axis]])
    ch = np.ascontiguousarray(chunks, np.int64)
    out = np.zeros(data.shape, np.uint8)
    lib().tro_sum_threshold(_p(data), _p(f8), _i64(n0), _i64(n1), C.c_int(axis),
                            _p(w), _i64(len(w)), C.c_double(float(outlier_nsigma)),
                            C.c_double(float(rho)), _p(ch), _i64(len(ch)), _p(out))
    return out.view(np.bool_)


def combine_flags(spec_flags, time_flags, freq_flags, time_extend):
    T, F = time_flags.shape
    s8 = np.ascontiguousarray(spec_flags != 0).view(np.uint8).reshape(-1)
    t8 = np.ascontiguousarray(time_flags != 0).view(np.uint8)
    q8 = np.ascontiguousarray(freq_flags != 0).view(np.uint8)
    out = np.zeros((T, F), np.uint8)
    lib().tro_combine_flags(_p(s8), _p(t8), _p(q8), _i64(T), _i64(F),
                            _i64(time_extend), _p(out))
    return out.view(np.bool_)


def unaverage_freq(flags, freq_extend, average_freq, flag_all_time_frac,
                   flag_all_freq_frac, orig_freq):
    T, fa = flags.shape
    f8 = np.ascontiguousarray(flags != 0).view(np.uint8)
    out = np.zeros((T, orig_freq), np.uint8)
    lib().tro_unaverage_freq(_p(f8), _i64(T), _i64(fa), _i64(orig_freq),
                             _i64(freq_extend), _i64(average_freq),
                             C.c_double(flag_all_time_frac),
                             C.c_double(flag_all_freq_frac), _p(out))
    return out.view(np.bool_)


def pack_data(time_inv, ubl, ant1, ant2, data, flag, ntime):
    """packing.py:243-278 on numpy inputs; windows initialised NaN+NaNj / 1."""
    rows, chans, corrs = data.shape
    nbl = ubl.shape[0]
    vis_w = np.full((nbl, corrs, ntime, chans), np.nan + np.nan * 1j, np.complex64)
    flag_w = np.ones((nbl, corrs, ntime, chans), np.uint8)
    lib().tro_pack_data(_p(np.ascontiguousarray(time_inv, np.int32)),
                        _p(np.ascontiguousarray(ubl, np.int32)), _i64(nbl),
                        _p(np.ascontiguousarray(ant1, np.int32)),
                        _p(np.ascontiguousarray(ant2, np.int32)),
                        _p(np.ascontiguousarray(data, np.complex64)),
                        _p(np.ascontiguousarray(flag != 0).view(np.uint8)),
                        _i64(rows), _i64(chans), _i64(corrs), _i64(ntime),
                        _p(vis_w), _p(flag_w))
    return vis_w, flag_w.view(np.bool_)


def unpack_data(time_inv, ubl, ant1, ant2, flag_windows):
    nbl, corrs, ntime, chans = flag_windows.shape
    rows = len(ant1)
    out = np.zeros((rows, chans, corrs), np.uint8)
    lib().tro_unpack_data(_p(np.ascontiguousarray(time_inv, np.int32)),
                          _p(np.ascontiguousarray(ubl, np.int32)), _i64(nbl),
                          _p(np.ascontiguousarray(ant1, np.int32)),
                          _p(np.ascontiguousarray(ant2, np.int32)), _i64(rows),
                          _i64(chans), _i64(corrs), _i64(ntime),
                          _p(np.ascontiguousarray(flag_windows != 0).view(np.uint8)),
                          _p(out))
    return out.view(np.bool_)


# ---- cheap strategy steps (numpy restatements) ------------

def flag_nans_and_zeros(vis_windows, flag_windows):
    """flagging.py:29-62."""
    if vis_windows.shape != flag_windows.shape:
        raise ValueError("vis_windows.shape != flag_windows.shape")
    flag = (vis_windows == 0) | np.isnan(vis_windows) | (flag_windows != 0)
    return flag.astype(flag_windows.dtype)


def flag_autos(flags, ubl):
    """flagging.py:65-95 (ubl wrapped in a list, :84)."""
    ubl = ubl[0]
    if flags.shape[0] != ubl.shape[0]:
        raise ValueError("flag and ubl shape mismatch")
    out = flags.copy()
    out[ubl[:, 1] == ubl[:, 2], :, :, :] = True
    return out


def apply_static_mask(flag, ubl, antspos, masks, chan_freqs, chan_widths,
                      accumulation_mode="or", uvrange=(0, np.inf)):
    """flagging.py:98-172 with the uv-range already parsed to (lo, hi)."""
    spw_chanlb = chan_freqs - chan_widths * 0.5
    spw_chanub = chan_freqs + chan_widths * 0.5
    bl_length = antspos[ubl[:, 1]] - antspos[ubl[:, 2]]
    d2 = 0.5 * np.sum(bl_length**2, axis=1)
    lo, hi = min(uvrange), max(uvrange)
    bl_sel = np.logical_and(d2 >= lo**2, d2 <= hi**2)
    out = flag.copy()
    for mask in masks:
        mc = np.logical_and(mask >= spw_chanlb[None, :], mask < spw_chanub[None, :]).sum(axis=0) > 0
        if accumulation_mode == "or":
            out[bl_sel, :, :, :] |= mc[None, None, None, :].astype(out.dtype)
        elif accumulation_mode == "override":
            out[bl_sel, :, :, :] = mc[None, None, None, :]
        else:
            raise ValueError("Invalid accumulation_mode")
    return out


def uvcontsub_flagger(vis, flags, major_cycles=5, or_original_from_cycle=1,
                      taylor_degrees=20, sigma=5, dump=False):
    """NumPy restatement of flagging.py:989-1073 (the reference routine is
    itself plain NumPy; results follow the running NumPy's FFT precision).
    dump=True: also returns the last cycle's |vis - smooth| and the per-product
    threshold sigma * mad (the tests use them to tell borderline samples)."""
    import warnings
    if vis.shape != flags.shape:
        raise ValueError("vis and flags must have the same shape")
    nbl, ncorr, ntime, nfreq = vis.shape
    v = vis.reshape(nbl * ncorr, ntime, nfreq)
    res = flags.reshape(nbl * ncorr, ntime, nfreq).astype(bool).copy()
    last_abs = np.full(v.shape, np.nan)
    last_thr = np.full(v.shape[0], np.nan)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for mi in range(major_cycles):
            start = res.copy()      # every product sees the flags of the cycle start
            for cp in range(v.shape[0]):
                if start[cp].all():
                    continue
                masked = v[cp].copy()
                masked[start[cp]] = np.nan
                avg = np.nanmean(masked, axis=0)
                avg[np.isnan(avg)] = 0.0
                spec = np.fft.fft(avg, axis=0)
                spec[taylor_degrees:] = 0
                smooth = np.fft.ifft(spec)
                absres = np.abs(v[cp] - smooth[None, :]).real
                far = absres.copy()
                far[start[cp]] = np.nan
                diff = np.abs(np.abs(far) - np.nanmedian(np.abs(far)))
                mad = np.nanmedian(np.abs(diff))
                new = absres > sigma * mad
                res[cp] = (start[cp] | new) if mi >= or_original_from_cycle else new
                last_abs[cp] = absres
                last_thr[cp] = sigma * mad
    if dump:
        return res.reshape(nbl, ncorr, ntime, nfreq), dict(absres=last_abs, thr=last_thr)
    return res.reshape(nbl, ncorr, ntime, nfreq)


# ---- flag summary statistics (numpy restatement) ----------

def window_counts(flag_window):
    """per-baseline and per-channel numbers of set flags of a (bl, corr, time,
    chan) window: the two reductions every tally of window_statistics.py:12-66
    decomposes into."""
    fw = np.asarray(flag_window) != 0
    return (fw.sum(axis=(1, 2, 3), dtype=np.uint64), fw.sum(axis=(0, 1, 2), dtype=np.uint64))


def window_stats_block(flag_window, ubls, chan_freqs, antenna_names, scan_no,
                       field_name, ddid, nchanbins=10):
    """Restates `_window_stats` (window_statistics.py:12-66) with the same
    masked reductions over the whole window as the reference; returns plain
    dicts  {"counts_per_ant": {...}, "size_per_ant": {...}, ...}."""
    fw = np.asarray(flag_window)
    ubls = np.asarray(ubls)
    chan_freqs = np.asarray(chan_freqs)
    res = {k: {} for k in ("counts_per_ant", "size_per_ant", "counts_per_bl", "size_per_bl",
                           "counts_per_field", "size_per_field", "counts_per_scan", "size_per_scan",
                           "counts_per_ddid", "bins_per_ddid", "size_per_ddid")}
    for ai, a in enumerate(antenna_names):                                   # :27-32
        sel = np.logical_or(ubls[:, 1] == ai, ubls[:, 2] == ai)
        res["counts_per_ant"][a] = int(np.sum(fw[sel], dtype=np.uint64))
        res["size_per_ant"][a] = int(fw[sel].size)
    for b in np.unique(ubls[:, 0]):                                          # :35-43
        sel = ubls[:, 0] == b
        name = "%s&%s" % (antenna_names[ubls[sel, 1][0]], antenna_names[ubls[sel, 2][0]])
        res["counts_per_bl"][name] = res["counts_per_bl"].get(name, 0) + int(np.sum(fw[sel], dtype=np.uint64))
        res["size_per_bl"][name] = res["size_per_bl"].get(name, 0) + int(fw[sel].size)
    total = int(np.sum(fw, dtype=np.uint64))                                 # :46-52
    res["counts_per_field"][field_name] = total
    res["size_per_field"][field_name] = int(fw.size)
    res["counts_per_scan"][scan_no] = total
    res["size_per_scan"][scan_no] = int(fw.size)
    edges = np.linspace(np.min(chan_freqs), np.max(chan_freqs), nchanbins)   # :55-62
    bins = np.zeros(nchanbins, dtype=np.uint32)
    for i in range(nchanbins - 1):
        sel = np.logical_and(chan_freqs >= edges[i], chan_freqs < edges[i + 1])
        bins[i] = np.sum(fw[:, :, :, sel], dtype=np.uint64)
    res["counts_per_ddid"][ddid] = bins.astype(np.uint64)
    res["bins_per_ddid"][ddid] = edges
    res["size_per_ddid"][ddid] = int(fw.size)
    return res


# ---- Stokes intensities (numpy restatement with numba's typing; 8f-4) -----

def _stokes_term(vis128, term):
    c1, c2, a, s1, s2 = term
    # int64 * complex64 promotes to complex128 under numba; complex(a) is complex128
    return complex(a) * (np.complex128(s1) * vis128[..., c1] + np.complex128(s2) * vis128[..., c2])


def polarised_intensity(vis, stokes_pol):
    """stokes.py:157-209: sqrt(sum |a (s1 v[c1] + s2 v[c2])|^2), float64 inside,
    cast to vis.dtype, shape (row, chan, 1)."""
    v = np.asarray(vis)
    v128 = v.astype(np.complex128)
    pol = np.zeros(v.shape[:2], np.float64)
    for term in stokes_pol:
        pol += np.abs(_stokes_term(v128, term)) ** 2
    return np.sqrt(pol)[..., None].astype(v.dtype)


def unpolarised_intensity(vis, stokes_unpol, stokes_pol):
    """stokes.py:79-153: sum_unpol |.| - sqrt(sum_pol |.|^2)."""
    v = np.asarray(vis)
    v128 = v.astype(np.complex128)
    pol = np.zeros(v.shape[:2], np.float64)
    for term in stokes_pol:
        pol += np.abs(_stokes_term(v128, term)) ** 2
    unpol = np.zeros(v.shape[:2], np.float64)
    for term in stokes_unpol:
        unpol += np.abs(_stokes_term(v128, term))
    return (unpol - np.sqrt(pol))[..., None].astype(v.dtype)
