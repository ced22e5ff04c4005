#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/G*.npz by running the reference
(ratt-ru/tricolour, /root/reference) UN-JITTED in the build container.

Run:  cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python3 make_golden.py [names...]

Only the resulting .npz data files travel to the GPU box; nothing of the
reference does.  Harness rules:
  * every float kwarg is wrapped in np.float64 (D3: Python floats are "weak"
    under NumPy 2 but float64 under numba);
  * visibilities are fed as float32 amplitudes (D6: NumPy's complex abs is not
    libm hypotf); complex inputs are restricted to samples with one zero
    component, whose magnitude is exact under any hypot;
  * fixtures therefore pin the oracle in (POW_POWF, INTERP_F32) mode, which
    differs from the numba-canonical mode only at D1/D2.
Intermediates of the LAST major iteration of correlation product 0 are
recorded by rebinding the reference's module-level helpers with recording
closures (they are looked up as globals at call time when un-jitted).
"""
import ctypes
import sys
import time

import numpy as np

from refshim import load_reference_flagging

FLOAT_KW = ("outlier_nsigma", "background_reject", "spike_width_time",
            "spike_width_freq", "flag_all_time_frac", "flag_all_freq_frac", "rho")


def run_reference(fl, vis, flags, **kw):
    calls = {"bg": [], "st": []}
    orig_bg, orig_st = fl._get_background2d, fl._sum_threshold

    def rec_bg(data, flags_, *a):
        out = orig_bg(data, flags_, *a)
        calls["bg"].append(out.copy())
        return out

    def rec_st(data, flags_, axis, *a):
        out = orig_st(data, flags_, axis, *a)
        calls["st"].append((data.copy(), out.copy()))
        return out

    fl._get_background2d, fl._sum_threshold = rec_bg, rec_st
    try:
        kw2 = {k: (np.float64(v) if k in FLOAT_KW else v) for k, v in kw.items()}
        out = fl.sum_threshold_flagger(vis, flags, **kw2)
    finally:
        fl._get_background2d, fl._sum_threshold = orig_bg, orig_st
    n_cp = vis.shape[0] * vis.shape[1]
    iters = kw.get("num_major_iterations", 5)
    base_bg = (iters - 1) * n_cp * 2
    base_st = (iters - 1) * n_cp * 3
    inter = dict(
        spec_resid=calls["st"][base_st][0].reshape(-1),
        spec_flags=calls["st"][base_st][1].reshape(-1),
        background=calls["bg"][base_bg + 1],
        residual=calls["st"][base_st + 1][0],
        time_flags=calls["st"][base_st + 1][1],
        freq_flags=calls["st"][base_st + 2][1],
    )
    return out, inter


def save(name, vis, flags, kw, out, inter, t0):
    kwitems = {"kw_" + k: np.asarray(v) for k, v in kw.items()}
    np.savez_compressed(name + ".npz", vis=vis, flags=flags, out=out,
                        **{"i_" + k: v for k, v in inter.items()}, **kwitems)
    print("%s: shape %s flagged %d/%d  (%.0f s)" % (name, vis.shape, out.sum(),
                                                    out.size, time.time() - t0))
    sys.stdout.flush()


def synth(rs, shape, bandpass=True):
    """float32 amplitudes: smooth bandpass + noise + deterministic RFI."""
    n_bl, n_corr, T, F = shape
    x = np.linspace(0, np.pi, F)
    bp = (5.0 + 0.4 * np.sin(x) + 0.05 * np.cos(5 * x)) if bandpass else np.full(F, 5.0)
    amp = bp[None, None, None, :] + rs.standard_normal(shape) * 0.5
    amp = amp.astype(np.float32)
    amp[..., F // 3] += 6.0                       # bad channel
    amp[..., T // 4, :] += 4.0                    # bad time
    amp[..., T // 2: T // 2 + 3, F // 2: F // 2 + 12] += 2.5   # broadband-ish blob
    for _ in range(max(2, amp.size // 2000)):     # isolated spikes
        idx = tuple(rs.randint(0, s) for s in shape)
        amp[idx] += 40.0
    return np.abs(amp)


def g0():
    """complex64 -> float32 amplitude KAT against libm hypotf (ctypes)."""
    libm = ctypes.CDLL("libm.so.6")
    libm.hypotf.restype = ctypes.c_float
    libm.hypotf.argtypes = [ctypes.c_float, ctypes.c_float]
    rs = np.random.RandomState(1)
    n = 100000
    mag = 10.0 ** rs.uniform(-20, 20, size=(n, 2))
    z = (rs.standard_normal((n, 2)) * mag).astype(np.float32)
    sub = np.float32(1e-45)
    special = np.array([[0, 0], [-0.0, 0], [0, 3], [np.inf, 1], [1, -np.inf],
                        [np.inf, np.nan], [np.nan, -np.inf], [np.nan, 1],
                        [1, np.nan], [np.nan, np.nan], [sub, sub], [sub, 0],
                        [3e38, 3e38], [1e-30, 1e-30], [3, 4], [5, 12]], np.float32)
    z[:len(special)] = special
    # equal and near-equal magnitudes (rounding stress)
    z[100:20000, 1] = z[100:20000, 0] * rs.uniform(0.5, 2.0, size=19900).astype(np.float32)
    out = np.array([libm.hypotf(float(a), float(b)) for a, b in z], np.float32)
    np.savez_compressed("G0_hypotf.npz", re=z[:, 0].copy(), im=z[:, 1].copy(), amp=out)
    print("G0: %d pairs" % n)


def main(names):
    fl = load_reference_flagging()
    todo = lambda n: not names or n in names  # noqa: E731

    if todo("G0"):
        g0()

    if todo("G1"):
        t0 = time.time(); rs = np.random.RandomState(1)
        vis = synth(rs, (2, 1, 48, 96))
        flags = np.zeros(vis.shape, np.bool_); flags[..., 70:72] = True
        kw = {}
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G1_defaults", vis, flags, kw, out, inter, t0)

    if todo("G2"):
        t0 = time.time(); rs = np.random.RandomState(2)
        vis = synth(rs, (1, 1, 64, 160))
        flags = np.zeros(vis.shape, np.bool_); flags[..., 100:103] = True
        kw = dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                  background_reject=2.0, background_iterations=5, spike_width_time=12.5,
                  spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                  average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                  num_major_iterations=3)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G2_stage1", vis, flags, kw, out, inter, t0)

    if todo("G2b"):
        # spike_width_time=37.5 -> r=32, d=65: a D1 site (float32(65)**4)
        t0 = time.time(); rs = np.random.RandomState(3)
        vis = synth(rs, (1, 1, 72, 96))
        flags = np.zeros(vis.shape, np.bool_)
        kw = dict(outlier_nsigma=6.0, background_iterations=1, spike_width_time=37.5,
                  spike_width_freq=39.5, num_major_iterations=1, freq_chunks=4)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G2b_radius32", vis, flags, kw, out, inter, t0)

    if todo("G3"):
        t0 = time.time(); rs = np.random.RandomState(4)
        vis = synth(rs, (1, 1, 32, 512))
        vis[..., 200:260] += 1.2   # broad low-level RFI for the wide windows
        flags = np.zeros(vis.shape, np.bool_)
        kw = dict(outlier_nsigma=10, windows_freq=[32, 48, 64, 128], background_iterations=2,
                  spike_width_time=6.5, spike_width_freq=64.0, num_major_iterations=1)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G3_broad", vis, flags, kw, out, inter, t0)

    if todo("G4"):
        # heavy pre-flagging: fully flagged column/row blocks wider than the
        # filter support -> NaN background -> interpolation (D2 site)
        t0 = time.time(); rs = np.random.RandomState(5)
        vis = synth(rs, (1, 1, 60, 200))
        flags = np.zeros(vis.shape, np.bool_)
        flags[..., 40:110] = True
        flags[..., 10:40, 150:] = True
        flags[..., :, 190:] = True
        flags |= rs.uniform(size=vis.shape) < 0.05
        kw = dict(spike_width_time=2.5, spike_width_freq=3.0, background_iterations=2,
                  num_major_iterations=2, freq_chunks=5)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G4_preflagged", vis, flags, kw, out, inter, t0)

    if todo("G5"):
        # complex64 input with NaN / zero / inf samples; one component zero so
        # that |z| is exact under any hypot (D6 neutralised)
        t0 = time.time(); rs = np.random.RandomState(6)
        amp = synth(rs, (1, 2, 40, 100))
        pick = rs.uniform(size=amp.shape) < 0.5
        sign = np.where(rs.uniform(size=amp.shape) < 0.5, -1.0, 1.0).astype(np.float32)
        vis = np.where(pick, amp * sign + 0j, 1j * amp * sign).astype(np.complex64)
        vis[0, 0, 5, 7] = np.nan + 1j
        vis[0, 0, 6, 9] = 1 + np.nan * 1j
        vis[0, 1, 7, 11] = complex(np.nan, np.nan)
        vis[0, 1, 8, 13] = complex(np.inf, np.nan)
        vis[0, 0, 9:12, 20:24] = 0
        vis[0, 1, :, 50] = np.nan
        flags = np.zeros(vis.shape, np.bool_); flags[0, 0, 30:, 80:] = True
        kw = dict(num_major_iterations=2)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G5_complex_nan", vis, flags, kw, out, inter, t0)

    if todo("G6"):
        t0 = time.time()
        vis = np.zeros((2, 1, 24, 40), np.float32)
        flags = np.ones(vis.shape, np.bool_)
        kw = dict(num_major_iterations=2)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G6_all_flagged", vis, flags, kw, out, inter, t0)

    if todo("G7"):
        # T < 8 and F < 128: window clipping (flagging.py:1176-1179)
        t0 = time.time(); rs = np.random.RandomState(7)
        vis = synth(rs, (1, 2, 6, 40))
        flags = np.zeros(vis.shape, np.bool_)
        kw = dict(windows_freq=[32, 48, 64, 128], windows_time=[1, 2, 4, 8],
                  spike_width_time=1.5, spike_width_freq=4.0, freq_chunks=3,
                  num_major_iterations=2, time_extend=5, freq_extend=1)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G7_clipping", vis, flags, kw, out, inter, t0)

    if todo("G10"):
        # average_freq=2 with windows that survive the function-path division
        t0 = time.time(); rs = np.random.RandomState(8)
        vis = synth(rs, (1, 1, 40, 101))
        flags = np.zeros(vis.shape, np.bool_); flags[..., 33:36] = True
        kw = dict(average_freq=2, windows_freq=[2, 4, 8, 16], freq_chunks=4,
                  num_major_iterations=2, freq_extend=5)
        out, inter = run_reference(fl, vis, flags, **kw)
        save("G10_average2", vis, flags, kw, out, inter, t0)

    if todo("G14"):
        # float64 / complex128 input WITH channel averaging: abs in float64, and `avg_data[...] += data` adds a float64 to a
        # float32 element -- rounded to float32 at every step (flagging.py:856-859).  Amplitudes carry bits below float32
        # precision so that this differs from rounding first; the complex samples have one zero component (|z| exact).
        t0 = time.time(); rs = np.random.RandomState(14)
        amp = synth(rs, (1, 2, 40, 101)).astype(np.float64) * (1.0 + 1e-9 * rs.standard_normal((1, 2, 40, 101)))
        flags = np.zeros(amp.shape, np.bool_); flags[..., 33:36] = True; flags[0, 1, 7, :] = True
        amp[0, 0, 3, 50] = np.nan
        kw = dict(average_freq=3, windows_freq=[3, 6, 12, 24], freq_chunks=4,
                  num_major_iterations=2, freq_extend=5)
        out, inter = run_reference(fl, amp, flags, **kw)
        save("G14_wide_average3", amp, flags, kw, out, inter, t0)
        pick = rs.uniform(size=amp.shape) < 0.5
        z = np.where(pick, amp + 0j, 1j * amp).astype(np.complex128)
        z[0, 1, 9, 13] = complex(np.inf, np.nan)
        flags2 = flags.copy(); flags2[0, 1, 9, 13] = True
        out, inter = run_reference(fl, z, flags2, **kw)
        save("G14b_complex128_average3", z, flags2, kw, out, inter, t0)


if __name__ == "__main__":
    main(sys.argv[1:])
