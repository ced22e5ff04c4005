"""Parity tests proper: the HIP path (through the C ABI) against the oracle
on the same inputs.  Flags must be bit-exact; float32 intermediates must be
bit-identical too (the kernels follow the reference's evaluation order)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu


def _same_f32(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32).reshape(a.shape)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def _compare(gpu, oracle, vis, flags, kw, label):
    dbg = {}
    out = gpu.sum_threshold_flagger(vis, flags, _debug=dbg, **kw)
    exp, inter = oracle.sum_threshold_flagger(vis, flags, dump=True, **kw)
    report = []
    if kw.get("num_major_iterations", 5) > 0:
        for k in ("spec_resid", "background", "residual"):
            bad = int((~_same_f32(inter[k], dbg[k])).sum())
            if bad:
                report.append("%s: %d float32 words differ" % (k, bad))
        for k in ("spec_flags", "time_flags", "freq_flags"):
            bad = int((inter[k].astype(bool) != dbg[k].reshape(inter[k].shape)).sum())
            if bad:
                report.append("%s: %d flags differ" % (k, bad))
    bad = int((out != exp).sum())
    if bad:
        report.append("out: %d of %d flags differ" % (bad, out.size))
    assert not report, "%s: %s" % (label, "; ".join(report))
    return out


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_inputs_vs_oracle(gpu, oracle, name):
    """Golden inputs, numba-canonical oracle (the GPU's contract)."""
    d, kw = load_golden(name)
    out = _compare(gpu, oracle, d["vis"], d["flags"], kw, name)
    # the committed reference output differs from the canonical semantics at
    # most at the D1/D2 float sites, never in these fixtures' flags
    assert np.array_equal(out, d["out"])


def test_hypotf_kat(gpu):
    """G0: |complex64| must equal libm hypotf bit-for-bit."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    d, _ = load_golden("G0_hypotf.npz")
    z = (d["re"] + 1j * d["im"]).astype(np.complex64)
    z.real[:] = d["re"]
    z.imag[:] = d["im"]
    zt = torch.from_numpy(z).cuda()
    out = torch.empty(z.shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tri_abs_c64(zt.data_ptr(), out.data_ptr(), z.size, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ok = _same_f32(d["amp"], got)
    assert ok.all(), "%d of %d amplitudes differ from hypotf" % ((~ok).sum(), ok.size)


def test_division_by_box_denominator(gpu):
    """The register-ring filter kernels divide by float32(2r+1)**4 through its reciprocal (box_divide):
    every one of the 2^32 float32 inputs, for every radius those kernels accept (and beyond), must give
    the correctly rounded IEEE quotient (flagging.py:419 is a plain float32 division)."""
    import ctypes as C
    from tricolour_amd import _lib
    bad = {}
    for r in list(range(1, 129)) + [166, 221, 277, 397, 795]:
        n = C.c_uint64(0)
        _lib.check(_lib.lib().tri_test_box_divide(r, C.byref(n), None))
        if n.value:
            bad[r] = n.value
    assert not bad, "inputs whose quotient differs from IEEE division, per radius: %s" % bad


def _spectrum_filter(data, flags, radius, variant):
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    n, c = data.shape
    d = torch.from_numpy(np.ascontiguousarray(data, np.float32)).cuda()
    f = torch.from_numpy(np.ascontiguousarray(flags, np.uint8)).cuda()
    ow = torch.full((n, c), -7.0, dtype=torch.float32, device="cuda")
    oo = torch.full((n, c), -7.0, dtype=torch.float32, device="cuda")
    ms = C.c_float(0)
    rc = _lib.lib().tri_bench_boxfilter(d.data_ptr(), f.data_ptr(), ow.data_ptr(), oo.data_ptr(), 1, n, c, radius,
                                        2, variant, 1, C.byref(ms), None)
    if rc != 0:
        return None
    torch.cuda.synchronize()
    return ow.cpu().numpy(), oo.cpu().numpy()


@pytest.mark.parametrize("shape,radius", [((96, 2), 8), ((96, 4), 8), ((200, 64), 8), ((333, 12), 21), ((1000, 8), 54),
                                          ((1000, 6), 54), ((64, 130), 5), ((500, 1008), 32), ((40, 4), 43)])
def test_spectrum_filter_routes(gpu, shape, radius):
    """The 1-D box cascade of the spectrum path: register-ring kernel (variant 1) against the stage-pipelined
    kernel with blocks of 16 and of 8 positions (variants 2, 3) and the flagger's own route (0) -- all four must
    agree bit for bit, including NaN data under flags, lines shorter than the filter and ragged column counts."""
    rs = np.random.RandomState(radius * 1000 + shape[0])
    data = (rs.standard_normal(shape) * 3 + 10).astype(np.float32)
    flags = rs.uniform(size=shape) < 0.1
    flags[shape[0] // 2: shape[0] // 2 + 3 * radius, 0] = True     # a gap wider than the filter
    data[flags & (rs.uniform(size=shape) < 0.3)] = np.nan
    ref = _spectrum_filter(data, flags, radius, 1)
    assert ref is not None
    ran = 0
    for variant in (0, 2, 3):
        got = _spectrum_filter(data, flags, radius, variant)
        if got is None:
            continue                                                # that block length does not apply to this shape
        ran += 1
        for name, a, b in (("weights", ref[0], got[0]), ("data", ref[1], got[1])):
            ok = _same_f32(a, b)
            assert ok.all(), "variant %d, %s image: %d of %d words differ (first at %s)" % (
                variant, name, (~ok).sum(), ok.size, np.argwhere(~ok)[0])
    assert ran >= 1


def test_interpolation_across_segments(gpu, oracle):
    """Background NaNs (weight 0 under wide flagged bands) are repaired per 512-channel segment: bands that
    start a line, end it, span several segments or a whole line must come out as the one-thread-per-line walk
    (and the oracle) has them, in the 2-D background and in the median spectrum alike."""
    rs = np.random.RandomState(77)
    shape = (3, 1, 16, 2048)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64) * 2 + 5
    flags = rs.uniform(size=shape) < 0.02
    flags[..., 0:40] = True            # band at the start of every line
    flags[..., 400:1300] = True        # spans two segment boundaries
    flags[..., 1530:1545] = True       # short band just across a boundary (1536)
    flags[..., 2000:2048] = True       # band at the end
    flags[1, 0, 3, :] = True           # a fully flagged time row
    flags[2] = True                    # a fully flagged window
    vis[0, 0, 5, 700] = np.nan
    kw = dict(num_major_iterations=2, spike_width_freq=6.0, spike_width_time=3.0, freq_chunks=4)
    _compare(gpu, oracle, vis, flags, kw, "segmented interpolation")


def _time_stage(data, flags, radius, variant):
    """tri_bench_boxfilter stage 0: data (W, n, C) float32, flags (W, n, C) bool -> filtered weight / data images."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    w, n, c = data.shape
    d = torch.from_numpy(np.ascontiguousarray(data, np.float32)).cuda()
    # TF4 packing: byte k of word [t // 4][c] = flag of time 4 (t // 4) + k
    f4 = np.ascontiguousarray(flags.astype(np.uint8).reshape(w, n // 4, 4, c).transpose(0, 1, 3, 2))
    f = torch.from_numpy(f4).cuda()
    ow = torch.full((w, n, c), -7.0, dtype=torch.float32, device="cuda")
    oo = torch.full((w, n, c), -7.0, dtype=torch.float32, device="cuda")
    ms = C.c_float(0)
    _lib.check(_lib.lib().tri_bench_boxfilter(d.data_ptr(), f.data_ptr(), ow.data_ptr(), oo.data_ptr(), w, n, c, radius,
                                              0, variant, 1, C.byref(ms), None))
    torch.cuda.synchronize()
    return ow.cpu().numpy(), oo.cpu().numpy()


@pytest.mark.parametrize("shape,radius", [((2, 64, 70), 8), ((1, 256, 64), 10), ((2, 1024, 130), 21), ((1, 512, 64), 32),
                                          ((2, 1024, 70), 43), ((1, 1024, 200), 54), ((1, 128, 64), 27), ((2, 64, 6), 15),
                                          ((1, 16, 64), 54), ((1, 2048, 64), 12), ((1, 256, 64), 28), ((1, 260, 70), 36),
                                          ((2, 256, 64), 40), ((1, 128, 130), 16), ((1, 512, 64), 35), ((1, 512, 64), 45),
                                          ((1, 256, 70), 47), ((2, 512, 64), 52), ((1, 300, 64), 55), ((1, 64, 64), 44), ((1, 256, 64), 50),
                                          ((1, 512, 64), 58)])
def test_time_stage_routes(gpu, shape, radius):
    """Time-axis stage of the 2-D background filter: LDS delay lines (variant 1), register delay lines K4r (2), the
    four-wave stage pipeline K4q with blocks of 8 where they apply (3) and of 16 throughout (5) and the flagger's own
    route (0) agree bit for bit -- every (register part, FIFO delay) split of the delay line, lines shorter than the
    filter, ragged column counts, NaN data under flags."""
    rs = np.random.RandomState(radius * 100 + shape[1])
    data = (rs.standard_normal(shape) * 3 + 10).astype(np.float32)
    flags = rs.uniform(size=shape) < 0.1
    flags[:, shape[1] // 3: shape[1] // 3 + 3 * radius, 0] = True       # a gap wider than the filter
    data[flags & (rs.uniform(size=shape) < 0.3)] = np.nan
    ref = _time_stage(data, flags, radius, 1)
    for variant in (2, 3, 5, 0):
        got = _time_stage(data, flags, radius, variant)
        for name, a, b in (("weights", ref[0], got[0]), ("data", ref[1], got[1])):
            ok = _same_f32(a, b)
            assert ok.all(), "variant %d, %s image: %d of %d words differ (first at %s)" % (
                variant, name, (~ok).sum(), ok.size, np.argwhere(~ok)[0])


def _freq_stage(wimg, oimg, data, radius, variant):
    """tri_bench_boxfilter stage 1: time-filtered weight / data images (W, T, F), amplitudes (W, F, T) ->
    |data - background| (W, F, T)."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    w, t, f = wimg.shape
    both = torch.from_numpy(np.ascontiguousarray(np.stack([wimg, oimg], axis=1), np.float32)).cuda()
    d = torch.from_numpy(np.ascontiguousarray(data, np.float32)).cuda()
    ow = torch.full((w, f, t), -7.0, dtype=torch.float32, device="cuda")
    oo = torch.full((w, f, t), -7.0, dtype=torch.float32, device="cuda")
    ms = C.c_float(0)
    _lib.check(_lib.lib().tri_bench_boxfilter(d.data_ptr(), both.data_ptr(), ow.data_ptr(), oo.data_ptr(), w, t, f, radius,
                                              1, variant, 1, C.byref(ms), None))
    torch.cuda.synchronize()
    return oo.cpu().numpy()


@pytest.mark.parametrize("shape,radius", [((2, 64, 128), 8), ((1, 72, 256), 10), ((2, 132, 1024), 21), ((1, 64, 512), 32),
                                          ((2, 36, 1024), 43), ((1, 200, 1024), 54), ((1, 64, 128), 27), ((1, 8, 64), 17),
                                          ((1, 64, 16), 54), ((1, 72, 256), 34), ((1, 64, 512), 38), ((2, 64, 256), 40),
                                          ((1, 132, 512), 30), ((1, 64, 512), 46), ((1, 64, 256), 13), ((2, 68, 128), 15),
                                          ((1, 64, 300), 12)])
def test_frequency_stage_routes(gpu, shape, radius):
    """Frequency-axis stage fused with the masked division: LDS delay lines (variant 1), register delay lines
    K4r (2), the eight-wave stage pipeline K4qf with blocks of 8 where they apply (3) and of 16 throughout (5) and the
    flagger's own route (0) agree bit for bit, including zero-weight bands (NaN background), lines shorter than the
    filter and ragged line counts."""
    w, t, f = shape
    rs = np.random.RandomState(radius * 100 + f)
    wimg = (rs.uniform(size=shape) * 0.9 + 0.05).astype(np.float32)
    wimg[:, :, f // 3: f // 3 + 5 * radius] = 0.0               # fully flagged band wider than the filter -> NaN background
    wimg[:, 1, :] = 0.0                                          # a line without any weight
    oimg = (wimg * (rs.standard_normal(shape) * 3 + 10)).astype(np.float32)
    data = (rs.standard_normal((w, f, t)) * 3 + 10).astype(np.float32)
    ref = _freq_stage(wimg, oimg, data, radius, 1)
    for variant in (2, 3, 5, 0):
        got = _freq_stage(wimg, oimg, data, radius, variant)
        ok = _same_f32(ref, got)
        assert ok.all(), "variant %d: %d of %d words differ (first at %s)" % (variant, (~ok).sum(), ok.size, np.argwhere(~ok)[0])


def _freq_stage_expected(oracle, wimg, oimg, data, radius):
    """flagging.py:362-419 row by row on both images, :506-513 masked division, :563-566 |data - background|,
    evaluated with the oracle's sequential box filter (any radius)."""
    w, t, f = wimg.shape
    out = np.empty((w, f, t), np.float32)
    with np.errstate(all="ignore"):
        for k in range(w):
            fw = np.stack([oracle.box_gaussian_filter1d(wimg[k, i], radius, 4) for i in range(t)])
            fo = np.stack([oracle.box_gaussian_filter1d(oimg[k, i], radius, 4) for i in range(t)])
            bg = np.where(fw == 0, np.float32(np.nan), fo / fw).astype(np.float32)
            out[k] = np.abs(data[k] - bg.T)
    return out


def _boxx_stats():
    import ctypes as C
    from tricolour_amd import _lib
    a, b = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.lib().tri_boxx_last_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


@pytest.mark.parametrize("shape,radius", [((1, 24, 4096), 277), ((1, 12, 4096), 221), ((1, 12, 4096), 166), ((1, 16, 4096), 110),
                                          ((2, 20, 4096), 43), ((1, 16, 1024), 17), ((1, 12, 256), 8), ((1, 12, 512), 166),
                                          ((1, 8, 64), 30), ((1, 8, 5376), 55)])
def test_exact_row_filter_vs_oracle(gpu, oracle, shape, radius):
    """K4x (kernels_boxexact.hpp): lanes = positions of one LDS-resident line, exactness of the float64 sums checked
    per pass, sequential redo otherwise.  Bit for bit against the oracle's sequential filter for (a) ordinary images
    (no line may need the redo), (b) images whose dynamic range breaks exactness, negative / infinite / NaN terms."""
    w, t, f = shape
    rs = np.random.RandomState(radius * 7 + f)
    wimg = (rs.uniform(size=shape) * 0.45 + 0.5).astype(np.float32)
    wimg[:, :, f // 3: f // 3 + min(5 * radius, f // 4)] = 0.0     # fully flagged band -> NaN background when wider than the filter
    wimg[:, 1, :] = 0.0                                            # a line without any weight
    oimg = (wimg * (rs.uniform(size=shape) * 10 + 5)).astype(np.float32)   # (positive, a few bits of dynamic range: amplitudes)
    data = (rs.standard_normal((w, f, t)) * 3 + 10).astype(np.float32)
    exp = _freq_stage_expected(oracle, wimg, oimg, data, radius)
    got = _freq_stage(wimg, oimg, data, radius, 4)
    ok = _same_f32(exp, got)
    assert ok.all(), "ordinary images: %d of %d words differ (first at %s)" % ((~ok).sum(), ok.size, np.argwhere(~ok)[0])
    passes, seq = _boxx_stats()
    assert passes == 4 * 2 * w * t and seq == 0, (passes, seq)
    # (b) hostile lines
    o2 = oimg.copy()
    o2[:, 2, :] *= (10.0 ** rs.uniform(-15, 15, size=f)).astype(np.float32)       # 2^100 of dynamic range: sums inexact
    o2[:, 3, f // 2] = -3.0                                                        # a negative term
    o2[:, 4, f // 5] = np.inf
    o2[:, 5, 2 * f // 3] = np.nan
    o2[:, 0, :4] = np.float32(1e-30)                                               # tiny terms at the line's start only
    w2 = wimg.copy()
    w2[:, 2, ::7] = np.float32(1e-12)
    exp = _freq_stage_expected(oracle, w2, o2, data, radius)
    got = _freq_stage(w2, o2, data, radius, 4)
    ok = _same_f32(exp, got)
    assert ok.all(), "hostile images: %d of %d words differ (first at %s)" % ((~ok).sum(), ok.size, np.argwhere(~ok)[0])
    passes, seq = _boxx_stats()
    assert passes == 4 * 2 * w * t and 0 < seq <= 4 * 2 * w * 6, (passes, seq)


def test_exact_row_filter_random_stress(gpu, oracle):
    """Seeded sweep over line lengths, radii and dynamic ranges that straddle the exactness boundary of K4x (2^28 between a
    window's total and its smallest term): whichever lines and passes take the sequential redo, the result must equal the
    oracle's sequential filter bit for bit.  Also: zero bands of random width, tiny values at the line ends only (where the
    padded tail votes or not), lines of constant value."""
    import ctypes as C
    from tricolour_amd import _lib
    rs = np.random.RandomState(2024)
    ran = redo = 0
    for case in range(48):
        f = int(rs.choice([64, 128, 260, 512, 1000, 2048, 4096]))
        r = int(rs.choice([8, 9, 13, 21, 33, 55, 56, 80, 110, 166, 277]))
        t = 8
        shape = (1, t, f)
        spread = rs.choice([0.5, 4.0, 9.0, 12.0])                     # decades of dynamic range
        wimg = (rs.uniform(size=shape) * 0.45 + 0.5).astype(np.float32)
        oimg = (wimg * (10.0 ** rs.uniform(-spread / 2, spread / 2, size=shape))).astype(np.float32)
        z0 = int(rs.randint(0, f))
        zw = int(rs.randint(0, max(1, f // 2)))
        wimg[:, 2, z0:z0 + zw] = 0.0
        oimg[:, 2, z0:z0 + zw] = 0.0
        if case % 2:
            oimg[:, 3, :3] = np.float32(1e-25)                       # tiny at the start only
            oimg[:, 4, -3:] = np.float32(1e-25)                      # tiny at the end only (the tail of the padded line)
        oimg[:, 5, :] = np.float32(3.25)
        wimg[:, 5, :] = np.float32(1.0)
        data = (rs.standard_normal((1, f, t)) * 3 + 10).astype(np.float32)
        both = np.ascontiguousarray(np.stack([wimg, oimg], axis=1), np.float32)
        try:
            got = _freq_stage(wimg, oimg, data, r, 4)
        except NotImplementedError:
            continue                                                 # no chunk length for this (line, radius)
        exp = _freq_stage_expected(oracle, wimg, oimg, data, r)
        ok = _same_f32(exp, got)
        assert ok.all(), "case %d (f = %d, r = %d, spread 10^%.1f): %d of %d words differ (first at %s)" % (
            case, f, r, spread, (~ok).sum(), ok.size, np.argwhere(~ok)[0])
        ran += 1
        redo += _boxx_stats()[1] > 0
    assert ran >= 30 and 0 < redo, (ran, redo)
    print("exact row filter stress: %d cases, %d with at least one pass redone sequentially" % (ran, redo))


def test_random_windows_multi_batch(gpu, oracle):
    """Several windows, tiny workspace budget -> several internal batches."""
    import os
    rs = np.random.RandomState(11)
    shape = (3, 2, 40, 70)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 20] *= 6
    vis[1, 0, 7] *= 5
    flags = rs.uniform(size=shape) < 0.03
    kw = dict(num_major_iterations=2, background_iterations=2, freq_chunks=4)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    from tricolour_amd import flagging
    old = os.environ.get("TRICOLOUR_AMD_WORKSPACE_GB")
    try:
        p = flagging.prepare_params(40, 70, **kw)
        import ctypes
        from tricolour_amd import _lib
        two = _lib.lib().tri_workspace_bytes(2, 40, 70, ctypes.byref(p))
        os.environ["TRICOLOUR_AMD_WORKSPACE_GB"] = repr((two + 4096) / 2**30)
        flagging.release_workspace()
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
    finally:
        if old is None:
            os.environ.pop("TRICOLOUR_AMD_WORKSPACE_GB", None)
        else:
            os.environ["TRICOLOUR_AMD_WORKSPACE_GB"] = old
        flagging.release_workspace()
    assert np.array_equal(out, exp)
    out_all = gpu.sum_threshold_flagger(vis, flags, **kw)
    assert np.array_equal(out_all, exp)


def test_nan_bitmap_is_per_batch(gpu, oracle):
    """Cached-amplitude path: the NaN bitmap the final pass reads instead of the amplitudes is rebuilt
    for every internal batch (NaNs in the first and the last window only, one window per batch)."""
    import ctypes
    from tricolour_amd import _lib, flagging
    rs = np.random.RandomState(23)
    shape = (4, 1, 32, 64)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[0, 0, 3, 5] = np.nan
    vis[3, 0, 30, 63] = complex(1.0, np.nan)
    vis[1, 0, 9] *= 7
    flags = rs.uniform(size=shape) < 0.03
    kw = dict(num_major_iterations=2)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    assert exp[0, 0, 3, 5] and exp[3, 0, 30, 63]
    old = os.environ.get("TRICOLOUR_AMD_WORKSPACE_GB")
    try:
        p = flagging.prepare_params(32, 64, **kw)
        one = _lib.lib().tri_workspace_bytes(1, 32, 64, ctypes.byref(p))
        os.environ["TRICOLOUR_AMD_WORKSPACE_GB"] = repr((one + 4096) / 2**30)
        flagging.release_workspace()
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
    finally:
        if old is None:
            os.environ.pop("TRICOLOUR_AMD_WORKSPACE_GB", None)
        else:
            os.environ["TRICOLOUR_AMD_WORKSPACE_GB"] = old
        flagging.release_workspace()
    assert np.array_equal(out, exp)
    assert np.array_equal(gpu.sum_threshold_flagger(vis, flags, **kw), exp)


def test_inputs_not_modified_and_torch_roundtrip(gpu, oracle):
    """tests/test_flagging.py:562-567 of the reference: inputs untouched."""
    import torch
    rs = np.random.RandomState(5)
    shape = (2, 1, 32, 64)
    amp = np.abs(rs.standard_normal(shape)).astype(np.float32) + 3
    amp[0, 0, 5, 9] = 50
    flags = np.zeros(shape, bool)
    vt, ft = torch.from_numpy(amp).cuda(), torch.from_numpy(flags).cuda()
    v0, f0 = vt.clone(), ft.clone()
    out = gpu.sum_threshold_flagger(vt, ft, num_major_iterations=1)
    assert out.is_cuda and out.dtype == torch.bool and tuple(out.shape) == shape
    assert torch.equal(vt, v0) and torch.equal(ft, f0)
    exp = oracle.sum_threshold_flagger(amp, flags, num_major_iterations=1)
    assert np.array_equal(out.cpu().numpy(), exp)


def _run_st_kernel(data, mad, windows, nsigma, rho, variant):
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    d = torch.from_numpy(np.ascontiguousarray(data, np.float32)).cuda()
    m = torch.from_numpy(np.ascontiguousarray(mad, np.float64)).cuda()
    out = torch.full(d.shape, 7, dtype=torch.uint8, device="cuda")
    warr = (C.c_int64 * len(windows))(*windows)
    ms = C.c_float(0)
    n_win, n_line, n_col = d.shape
    _lib.check(_lib.lib().tri_bench_sumthreshold(d.data_ptr(), m.data_ptr(), out.data_ptr(), n_win,
                                                 n_line, n_col, warr, len(windows), nsigma, rho,
                                                 variant, 1, C.byref(ms), None))
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(bool)


@pytest.mark.parametrize("shape", [(2, 200, 70), (1, 5, 300), (3, 1024, 130), (1, 17, 64), (2, 33, 257), (2, 300, 192), (1, 1024, 256)])
def test_fused_sumthreshold_kernel_vs_generic_and_oracle(gpu, oracle, shape):
    """The register cascade (windows 1,2,4,8) against the generic kernel and
    against the oracle's _sum_threshold along axis 0, including lines without
    any unflagged sample (NaN MAD -> infinite threshold) and short lines."""
    rs = np.random.RandomState(shape[1])
    data = rs.standard_normal(shape).astype(np.float32) * 2.0
    data[:, shape[1] // 3, :] += 30.0
    data[:, :, 5] -= 25.0
    data[0, -1, :] += 40.0        # hit at the very end of the line
    data[0, 0, :3] -= 40.0        # and at the very start
    flags = rs.uniform(size=shape) < 0.1
    flags[:, :, 7] = True         # a fully flagged line
    windows = [w for w in (1, 2, 4, 8) if w <= shape[1]]
    mad = np.empty((shape[0], shape[2]), np.float64)
    for w in range(shape[0]):
        mad[w] = oracle.median_abs_axis0(data[w], flags[w]).astype(np.float64).reshape(-1)
    gen = _run_st_kernel(data, mad, windows, 4.5, 1.3, 1)
    for w in range(shape[0]):
        exp = oracle.sum_threshold(data[w], flags[w], 0, np.array(windows), 4.5, 1.3)
        assert np.array_equal(gen[w], exp), "generic kernel, window %d" % w
    if windows == [1, 2, 4, 8]:
        fused = _run_st_kernel(data, mad, windows, 4.5, 1.3, 2)
        assert np.array_equal(fused, gen)
        mask = _run_st_kernel(data, mad, windows, 4.5, 1.3, 3)
        assert np.array_equal(mask, gen), "lane-mask cascade"
        if shape[2] % 64 == 0:
            panel = _run_st_kernel(data, mad, windows, 4.5, 1.3, 5)
            assert np.array_equal(panel, gen), "lane-mask cascade on column panels"


@pytest.mark.parametrize("shape,windows", [((2, 700, 70), (32, 48, 64, 128)), ((1, 300, 64), (32, 48, 64, 128)),
                                           ((1, 100, 130), (32, 48, 64, 128)), ((2, 200, 64), (1, 2, 4, 8)),
                                           ((1, 500, 200), (3, 5, 16, 17, 40)), ((1, 64, 64), (1, 2, 4, 8, 16, 32, 64, 128)),
                                           ((1, 2000, 64), (7,))])
def test_sumthreshold_stage_pipeline_vs_generic_and_oracle(gpu, oracle, shape, windows):
    """K7p (one window per wave, prefix rings in LDS) against the generic global-scratch kernel and the oracle:
    final_st_very_broad's windows, lists of one to eight windows, windows wider than the line, lines of many
    blocks, NaN MADs, ragged column counts."""
    rs = np.random.RandomState(shape[1] + len(windows))
    data = rs.standard_normal(shape).astype(np.float32) * 2.0
    data[:, shape[1] // 3: shape[1] // 3 + 40, :] += 3.0      # broad bump: only the wide windows see it
    data[:, :, 5] -= 25.0
    data[0, -1, :] += 40.0
    data[0, 0, :3] -= 40.0
    flags = rs.uniform(size=shape) < 0.1
    flags[:, :, 7] = True
    mad = np.empty((shape[0], shape[2]), np.float64)
    for w in range(shape[0]):
        mad[w] = oracle.median_abs_axis0(data[w], flags[w]).astype(np.float64).reshape(-1)
    wl = [w for w in windows if w <= shape[1]]
    gen = _run_st_kernel(data, mad, wl, 4.5, 1.3, 1)
    pipe = _run_st_kernel(data, mad, wl, 4.5, 1.3, 4)
    assert np.array_equal(pipe, gen), "%d flags differ" % (pipe != gen).sum()
    for w in range(shape[0]):
        exp = oracle.sum_threshold(data[w], flags[w], 0, np.array(wl), 4.5, 1.3)
        assert np.array_equal(pipe[w], exp), "window %d" % w


@pytest.mark.parametrize("n_line", [16, 40, 100])
def test_sumthreshold_hit_only_in_last_window_position(gpu, oracle, n_line):
    """A sample that only the widest window STARTING at it can flag (seven
    strong outliers follow it): the dilation of the last stage has to reach back
    w-1 positions before that position's flag is written out."""
    n_col = 64
    rs = np.random.RandomState(11)
    data = (rs.standard_normal((1, n_line, n_col)) * 0.3).astype(np.float32)
    for c in range(n_col):
        p = 1 + (c * 5) % (n_line - 9)
        sign = -1.0 if c % 2 else 1.0
        data[0, p, c] = sign * np.float32(0.92)
        data[0, p + 1:p + 8, c] = sign * (5.2 + 0.01 * np.arange(7, dtype=np.float32))
    flags = np.zeros(data.shape, bool)
    mad = np.full((1, n_col), 0.2714695930480957, np.float64)
    windows = [1, 2, 4, 8]
    gen = _run_st_kernel(data, mad, windows, 4.5, 1.3, 1)
    assert gen.any() and not gen.all()
    for variant in (2, 3):
        assert np.array_equal(_run_st_kernel(data, mad, windows, 4.5, 1.3, variant), gen), "variant %d" % variant
    # the pattern does what it is meant to: position p is flagged, p - 1 is not
    for c in range(n_col):
        p = 1 + (c * 5) % (n_line - 9)
        assert gen[0, p, c] and not gen[0, p - 1, c], "column %d" % c


def _np_median_abs(vals):
    """numba's np.median on float32: odd -> middle; even -> f32(a + b) / 2 in f64."""
    v = np.sort(np.abs(vals).astype(np.float32))
    n = v.size
    if n == 0:
        return np.nan
    if n & 1:
        return float(v[n // 2])
    return float(np.float32(v[n // 2 - 1] + v[n // 2])) / 2.0


@pytest.mark.parametrize("rows,row_len,ends,variants", [
    (5, 1024, [0, 1024], (1, 2, 3, 11)),
    (3, 1000, [0, 1000], (1, 2, 3, 11)),
    (2, 772, [0, 772], (1, 11)),
    (3, 410, [0, 41, 82, 123, 410], (1, 2, 11)),
    (3, 4096, [0, 409, 819, 1228, 1638, 2048, 2457, 2867, 3276, 3686, 4096], (1, 4, 11, 14)),
    (2, 1024, [3, 1021], (1, 4, 11, 14)),
    (2, 64, [1, 2, 7, 61, 64], (1, 4, 11, 14)),
    (2, 4096, [0, 4, 8, 2048, 4096], (2, 3)),
    (1, 6000, [0, 1, 2, 3001, 6000], (2,)),
    (4, 64, [0, 0, 1, 2, 64], (1, 2, 11)),
    (1, 300000, [0, 100000, 300000], (2, 3, 5, 6)),
    (2, 4096, [0, 4, 8, 2048, 4096], (5, 6)),
    (3, 40000, [0, 4000, 4004, 40000], (5, 6)),
    (1, 6000, [0, 1, 2, 3001, 6000], (6, 7)),
    (3, 40000, [0, 3999, 4001, 26214, 40000], (6, 7)),      # two-pass kernel, vector loads, segments that start / end inside a 16-byte group
    (2, 8, [0, 1, 3, 8], (7,)),
    (2, 1 << 20, [0, 1 << 20], (8, 5)),                     # multi-workgroup select (16 slices per row)
    (3, 70001, [0, 70001], (8, 10)),                        # ... scalar loads, ragged last slice
    (1, 300000, [0, 100000, 300000], (9, 10)),              # predicted candidate window (hits, misses on the odd rows)
    (2, 1 << 20, [0, 1 << 20], (9,)),
    (3, 419432, [0, 8, 65544, 419432], (9, 10)),            # short / just-long-enough / block-median-sized segments
])
def test_median_kernels(gpu, rows, row_len, ends, variants):
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    rs = np.random.RandomState(rows * 1000 + row_len)
    n_win = 2
    data = rs.standard_normal((n_win, rows, row_len)).astype(np.float32)
    data[0, 0, : row_len // 2] = np.float32(0.5)          # heavy duplicates
    data[1, -1] *= np.float32(1e-20)                       # tiny magnitudes
    data[0, -1, ::3] *= np.float32(1e12)                   # wide dynamic range
    flags = rs.uniform(size=data.shape) < 0.3
    flags[1, 0, :] = True                                   # nothing unflagged
    if rows > 1:
        flags[0, 1, :] = False
    G = len(ends) - 1
    exp = np.empty((n_win, rows, G))
    for w in range(n_win):
        for r in range(rows):
            for g in range(G):
                seg = slice(ends[g], ends[g + 1])
                exp[w, r, g] = _np_median_abs(data[w, r, seg][~flags[w, r, seg]])
    d = torch.from_numpy(data).cuda()
    f = torch.from_numpy(flags).cuda().view(torch.uint8)
    e = (C.c_int64 * len(ends))(*ends)
    for variant in variants:
        med = torch.full((n_win, rows, G), -1.0, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().tri_test_median(d.data_ptr(), f.data_ptr(), med.data_ptr(), n_win, rows,
                                              row_len, e, len(ends), variant, None))
        got = med.cpu().numpy()
        same = (got == exp) | (np.isnan(got) & np.isnan(exp))
        assert same.all(), "variant %d: %d medians differ" % (variant, (~same).sum())


@pytest.mark.parametrize("row_len,ends", [(512, [0, 512]), (1024, [0, 1024]), (2048, [0, 409, 819, 1228, 1640, 2048])])
def test_median_wave_corners(gpu, row_len, ends):
    """The wave medians (kernels_median.hpp k_median_wave; several rows of a segment per wave in variants 1 / 4, one segment
    per wave in 11) on keys laid out to hit every branch of the radix select: key spreads of 7 ... 20 bits (one to three
    digits, a short last digit), 63 ... 130 keys sharing the median's first-digit bin, duplicates straddling the median rank,
    the lower middle element alone in a lower bin, odd and even counts, row counts that do not divide by the rows per wave."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    rs = np.random.RandomState(row_len)
    base = np.float32(1.5).view(np.uint32)
    rows = []

    def from_offsets(off):
        return (base + off.astype(np.uint32)).view(np.float32) * rs.choice(np.float32([-1, 1]), size=off.size)

    for bits in (7, 8, 9, 10, 11, 12, 20):
        rows.append(from_offsets(rs.randint(0, 1 << bits, size=row_len)))
    for dup in (63, 64, 65, 66, 130):
        # `dup` keys share the bin of the median (spread 2^20: bins of 1024 key values), the others lie well outside it
        for shift in (0, 1):
            off = np.concatenate([rs.randint(0, 1 << 19, size=(row_len - dup) // 2 + shift),
                                  (1 << 19) + rs.randint(0, 1000, size=dup),
                                  (1 << 19) + 4096 + rs.randint(0, 1 << 19, size=row_len - dup - (row_len - dup) // 2 - shift)])
            off[0], off[-1] = 0, (1 << 20) - 1
            rows.append(from_offsets(rs.permutation(off)))
    # equal keys around the median rank; the lower middle element alone in a lower bin
    off = rs.randint(0, 1 << 16, size=row_len); off[: row_len // 3] = 1 << 15
    rows.append(from_offsets(off))
    off = np.concatenate([np.full(row_len // 2, 5), np.full(row_len - row_len // 2, 5 + (1 << 14))]); off[0] = 0; off[-1] = 1 << 15
    rows.append(from_offsets(off))
    data = np.stack(rows + rows[:3])[None].astype(np.float32)       # (21 rows: a ragged last block of rows per wave)
    data = np.concatenate([data, data[:, :, ::-1]], axis=0)
    flags = np.zeros(data.shape, dtype=bool)
    flags[1] = rs.uniform(size=data.shape[1:]) < 0.01          # shifts the ranks by a few, odd / even counts
    flags[1, :, 5] = True
    G = len(ends) - 1
    n_win, R = data.shape[:2]
    exp = np.empty((n_win, R, G))
    for w in range(n_win):
        for r in range(R):
            for g in range(G):
                seg = slice(ends[g], ends[g + 1])
                exp[w, r, g] = _np_median_abs(data[w, r, seg][~flags[w, r, seg]])
    d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    f = torch.from_numpy(flags).cuda().view(torch.uint8)
    e = (C.c_int64 * len(ends))(*ends)
    for variant in ((1, 11) if max(np.diff(ends)) + 3 > 1024 else (1, 4, 11)):
        med = torch.full((n_win, R, G), -1.0, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().tri_test_median(d.data_ptr(), f.data_ptr(), med.data_ptr(), n_win, R, row_len, e, len(ends), variant, None))
        got = med.cpu().numpy()
        assert np.array_equal(got, exp), "variant %d: %d medians differ" % (variant, (got != exp).sum())


@pytest.mark.parametrize("n_win,F,T,ends,reject", [
    (2, 256, 1024, [0, 128, 256], 2.0),
    (3, 200, 1024, [0, 67, 131, 200], 1.5),        # chunks that are not multiples of the 64-row tiles
    (1, 512, 512, [0, 512], 3.0),
])
def test_rejection_step_hook_vs_numpy(gpu, n_win, F, T, ends, reject):
    """tri_bench_reject = one rejection step of the background loop by the one-pass route (k_mr_predict / k_mr_pass /
    k_mr_finish + redo): flags |= resid > median_abs(resid[~flags]) * 1.4826 * reject per (window, chunk) block
    (flagging.py:553-574), the flags also as TF4 words, the block medians -- against numpy, with duplicate-heavy, constant,
    all-flagged and wide-range blocks among the ordinary ones."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    rs = np.random.RandomState(n_win * 100 + F)
    resid = np.abs(rs.standard_normal((n_win, F, T))).astype(np.float32)
    resid[0, : ends[1], ::7] = np.float32(0.25)                       # duplicates around the median
    resid[-1, ends[-2]:, :] *= np.float32(1e-12)
    resid[-1, ends[-2]:, ::5] *= np.float32(1e20)                      # 32 decimal orders inside one block
    flags = rs.uniform(size=resid.shape) < 0.1
    if n_win > 1:
        flags[1, : ends[1]] = True                                     # nothing unflagged in a block
        resid[1, ends[1]: ends[2]] = np.float32(3.0)                   # a constant block
    resid[0, 5, 5] = np.nan
    G = len(ends) - 1
    exp_med = np.empty((n_win, G))
    exp = flags.copy()
    for w in range(n_win):
        for g in range(G):
            blk = (w, slice(ends[g], ends[g + 1]))
            m = _np_median_abs(resid[blk][~flags[blk]])
            exp_med[w, g] = m
            with np.errstate(invalid="ignore"):
                exp[blk] |= resid[blk].astype(np.float64) > m * (1.4826 * reject)
    d = torch.from_numpy(resid).cuda()
    f = torch.from_numpy(flags).cuda().view(torch.uint8)
    fo = torch.full((n_win, F, T), 0xEE, dtype=torch.uint8, device="cuda")
    t4 = torch.full((n_win, T // 4, F, 4), 0xDD, dtype=torch.uint8, device="cuda")
    med = torch.full((n_win, G), -1.0, dtype=torch.float64, device="cuda")
    e = (C.c_int64 * len(ends))(*ends)
    ms = C.c_float(0)
    _lib.check(_lib.lib().tri_bench_reject(d.data_ptr(), f.data_ptr(), fo.data_ptr(), t4.data_ptr(), med.data_ptr(), n_win, F, T,
                                           e, len(ends), reject, 1, C.byref(ms), None))
    got_med = med.cpu().numpy()
    assert ((got_med == exp_med) | (np.isnan(got_med) & np.isnan(exp_med))).all(), (got_med, exp_med)
    got = fo.cpu().numpy()
    assert np.array_equal(got != 0, exp), "%d FT flags differ" % ((got != 0) != exp).sum()
    assert set(np.unique(got)) <= {0, 1}
    got4 = t4.cpu().numpy().transpose(0, 2, 1, 3).reshape(n_win, F, T)
    assert np.array_equal(got4, got), "%d TF4 flag bytes differ from the FT image" % (got4 != got).sum()
    assert ms.value > 0


@pytest.mark.parametrize("kind", ["bimodal", "cauchy", "mostly_flagged", "constant", "plateau", "zeros", "wide_exponents",
                                  "ramp", "near_power_of_two", "inf_nan", "two_values", "tiny_sample"])
def test_rejection_step_adversarial_distributions(gpu, kind):
    """The one-pass rejection step (prediction from 4 % of a block, provisional threshold, second round, redo) on
    residual distributions built to break a prediction: two populations, heavy tails, almost everything flagged, a
    constant block, a plateau of equal values around the median, exact zeros, 80 binary orders of magnitude, a sorted
    ramp (the sampled runs are not representative), values hugging a power of two, unflagged infinities and NaNs.
    Whatever route a block takes, flags (both layouts) and medians must equal numpy's."""
    import ctypes as C
    import torch
    from tricolour_amd import _lib
    rs = np.random.RandomState(sum(map(ord, kind)))
    n_win, F, T, ends, reject = 2, 300, 1024, [0, 100, 170, 300], 2.0
    shape = (n_win, F, T)
    x = np.abs(rs.standard_normal(shape)).astype(np.float32)
    flags = rs.uniform(size=shape) < 0.05
    if kind == "bimodal":
        x[:, :, : T // 2] *= np.float32(100.0)
        x[1, :150] *= np.float32(1e-3)
    elif kind == "cauchy":
        x = np.abs(rs.standard_cauchy(shape)).astype(np.float32)
    elif kind == "mostly_flagged":
        flags = rs.uniform(size=shape) < 0.97
        flags[1, 100:170] = rs.uniform(size=(70, T)) < 0.999
    elif kind == "constant":
        x[0] = np.float32(2.5)
        x[1, :100] = np.float32(0.0)
    elif kind == "plateau":
        x[rs.uniform(size=shape) < 0.52] = np.float32(0.71875)
    elif kind == "zeros":
        x[rs.uniform(size=shape) < 0.6] = np.float32(0.0)
    elif kind == "wide_exponents":
        x = (x * np.exp2(rs.randint(-40, 41, size=shape)).astype(np.float32)).astype(np.float32)
    elif kind == "ramp":
        x = np.sort(x.reshape(n_win, -1), axis=1).reshape(shape)
        x[1] = x[1, ::-1, ::-1]
    elif kind == "near_power_of_two":
        x = (np.float32(1.0) + (x * np.float32(1e-6))).astype(np.float32)
        x[1] = (np.float32(2.0) - x[1] * np.float32(1e-7)).astype(np.float32)
    elif kind == "inf_nan":
        x[rs.uniform(size=shape) < 1e-3] = np.inf
        x[rs.uniform(size=shape) < 1e-3] = np.nan
    elif kind == "two_values":
        x = np.where(rs.uniform(size=shape) < 0.5, np.float32(1.0), np.float32(3.0)).astype(np.float32)
    elif kind == "tiny_sample":
        # unflagged samples only where the prediction does not look: its runs find (almost) nothing
        flags[:] = True
        flags[:, ::7, 900:] = False
    G = len(ends) - 1
    exp_med = np.empty((n_win, G))
    exp = flags.copy()
    for w in range(n_win):
        for g in range(G):
            blk = (w, slice(ends[g], ends[g + 1]))
            m = _np_median_abs(x[blk][~flags[blk]])                     # (np.sort puts NaNs last: the key order of the select)
            exp_med[w, g] = m
            with np.errstate(invalid="ignore"):
                exp[blk] |= x[blk].astype(np.float64) > m * (1.4826 * reject)
    d = torch.from_numpy(x).cuda()
    f = torch.from_numpy(flags).cuda().view(torch.uint8)
    fo = torch.full(shape, 0xEE, dtype=torch.uint8, device="cuda")
    t4 = torch.full((n_win, T // 4, F, 4), 0xDD, dtype=torch.uint8, device="cuda")
    med = torch.full((n_win, G), -1.0, dtype=torch.float64, device="cuda")
    e = (C.c_int64 * len(ends))(*ends)
    ms = C.c_float(0)
    _lib.check(_lib.lib().tri_bench_reject(d.data_ptr(), f.data_ptr(), fo.data_ptr(), t4.data_ptr(), med.data_ptr(), n_win, F, T,
                                           e, len(ends), reject, 1, C.byref(ms), None))
    got_med = med.cpu().numpy()
    assert ((got_med == exp_med) | (np.isnan(got_med) & np.isnan(exp_med))).all(), (kind, got_med, exp_med)
    got = fo.cpu().numpy()
    assert np.array_equal(got != 0, exp), "%s: %d FT flags differ" % (kind, ((got != 0) != exp).sum())
    got4 = t4.cpu().numpy().transpose(0, 2, 1, 3).reshape(shape)
    assert np.array_equal(got4 != 0, exp), "%s: %d TF4 flags differ" % (kind, ((got4 != 0) != exp).sum())


EDGE_CASES = [
    # (shape, kwargs)
    ((1, 1, 1, 16), dict(num_major_iterations=1)),
    ((1, 1, 16, 1), dict(num_major_iterations=1, freq_chunks=1)),
    ((2, 1, 3, 5), dict(num_major_iterations=2, freq_chunks=7)),          # more chunks than channels
    ((1, 2, 20, 33), dict(num_major_iterations=1, background_iterations=0)),
    ((1, 1, 24, 48), dict(num_major_iterations=1, time_extend=0, freq_extend=0)),
    ((1, 1, 24, 48), dict(num_major_iterations=2, time_extend=4, freq_extend=6)),
    ((1, 1, 24, 48), dict(num_major_iterations=1, time_extend=1, freq_extend=33)),
    ((1, 1, 30, 64), dict(num_major_iterations=1, windows_time=[8, 1, 4], windows_freq=[3, 5.5, 7])),
    ((1, 1, 12, 40), dict(num_major_iterations=1, windows_time=[1, 2, 4, 8, 16, 64], windows_freq=[1, 64])),
    ((1, 1, 40, 64), dict(num_major_iterations=1, flag_all_time_frac=0.0, flag_all_freq_frac=0.0)),
    ((1, 1, 40, 64), dict(num_major_iterations=1, flag_all_time_frac=1.0, flag_all_freq_frac=1.0)),
    ((1, 1, 40, 64), dict(num_major_iterations=1, spike_width_time=0.1, spike_width_freq=0.1)),   # r = 0 on both axes
    ((1, 1, 40, 64), dict(num_major_iterations=1, spike_width_time=0.1, spike_width_freq=6.0)),
    ((1, 1, 40, 64), dict(num_major_iterations=1, spike_width_time=6.0, spike_width_freq=0.1)),
    ((1, 1, 36, 60), dict(num_major_iterations=2, average_freq=3, windows_freq=[3, 6, 12], freq_chunks=2)),
    ((1, 1, 64, 128), dict(num_major_iterations=1, outlier_nsigma=0.0)),
    ((3, 1, 16, 32), dict(num_major_iterations=0)),
    ((1, 1, 128, 64), dict(num_major_iterations=1, spike_width_time=45.0, background_iterations=2)),  # radii up to 77
    # awkward extents for the single-sweep filters: lines / positions that are no multiple of the
    # 16-, 64- or 128-line workgroups and of the 32-step blocks
    ((1, 1, 100, 150), dict(num_major_iterations=2)),                                   # K4b time, fused K4b'' frequency
    ((2, 1, 68, 97), dict(num_major_iterations=1, background_iterations=2)),            # odd channel count
    ((1, 1, 76, 200), dict(num_major_iterations=1, spike_width_time=30.0, spike_width_freq=25.0)),   # K4c both axes (r = 25, 21)
    ((1, 1, 256, 96), dict(num_major_iterations=1, spike_width_time=130.0, spike_width_freq=20.0)),  # r = 112 > n / 4
    ((1, 2, 52, 120), dict(num_major_iterations=2, background_iterations=3)),           # radii 32 / 25 ... 10 / 8
]


@pytest.mark.parametrize("case", range(len(EDGE_CASES)))
def test_edge_cases_vs_oracle(gpu, oracle, case):
    shape, kw = EDGE_CASES[case]
    rs = np.random.RandomState(1000 + case)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    if shape[3] > 4:
        vis[..., shape[3] // 3] *= 7
    if shape[2] > 4:
        vis[:, :, shape[2] // 2, :] *= 5
    flags = rs.uniform(size=shape) < 0.05
    out = gpu.sum_threshold_flagger(vis, flags, **kw)
    if kw.get("num_major_iterations", 5) == 0:
        assert out.shape == shape and out.dtype == np.bool_   # reference: np.empty_like
        return
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    assert np.array_equal(out, exp), "%d of %d flags differ" % ((out != exp).sum(), out.size)


def test_empty_inputs(gpu):
    for shape in ((0, 4, 8, 16), (2, 0, 8, 16)):
        out = gpu.sum_threshold_flagger(np.zeros(shape, np.complex64), np.zeros(shape, bool))
        assert out.shape == shape and out.dtype == np.bool_


def test_full_size_window_vs_oracle(gpu, oracle):
    """One MeerKAT-sized (1024 x 4096) window pair, bit-exact against the
    oracle (the oracle needs a few seconds per window and iteration)."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    shape = (1, 2, 1024, 4096)
    re = torch.randn(shape, generator=g, device="cuda")
    im = torch.randn(shape, generator=g, device="cuda")
    re[..., ::97] += 8.0
    re[:, :, ::211, :] += 6.0
    re[0, 0, 100:140, 2000:2300] += 2.0
    re.view(-1)[torch.randint(0, re.numel(), (500,), generator=g, device="cuda")] += 50.0
    re.view(-1)[torch.randint(0, re.numel(), (50,), generator=g, device="cuda")] = float("nan")
    vis = torch.complex(re, im)
    flags = torch.zeros(shape, dtype=torch.bool, device="cuda")
    flags[..., ::50] = True
    flags[0, 1, 300:320, :] = True
    kw = dict(num_major_iterations=2)
    out = gpu.sum_threshold_flagger(vis, flags, **kw).cpu().numpy()
    exp = oracle.sum_threshold_flagger(vis.cpu().numpy(), flags.cpu().numpy(), n_threads=2, **kw)
    nbad = int((out != exp).sum())
    assert nbad == 0, "%d of %d flags differ" % (nbad, out.size)


# Every sum_threshold parameter set of the shipped strategy file (conf/default.yaml:17-35, 59-105),
# restated here because the reference tree does not travel to the GPU box.
SHIPPED_KWARGS = {
    # "background_flags": box radii [54,43] [43,34] [32,25] [21,17] [10,8], all five major iterations
    "stage1": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                   background_reject=2.0, background_iterations=5, spike_width_time=12.5,
                   spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                   average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                   num_major_iterations=5),
    # "final_st_broad" (:74-89): time radii 28 / 22 / 16 / 11 / 5, frequency radii as stage 1
    "broad": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                  background_reject=2.0, background_iterations=5, spike_width_time=6.5,
                  spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                  average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                  num_major_iterations=1),
    # "final_st_narrow" (:90-105): spike_width_time arrives as a Python int (2): time radii 8 / 6 / 5 / 3 / 1
    "narrow": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                   background_reject=2.0, background_iterations=5, spike_width_time=2,
                   spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                   average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                   num_major_iterations=1),
    # "final_st_very_broad": frequency radii 277 / 221 / 166 / 110 / 55 (the first three take the
    # in-place multi-pass filter by the default route), frequency windows 32...128 (generic
    # SumThreshold kernel on full lines)
    "very_broad": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8],
                       windows_freq=[32, 48, 64, 128], background_reject=2.0,
                       background_iterations=5, spike_width_time=6.5, spike_width_freq=64.0,
                       time_extend=3, freq_extend=3, freq_chunks=10, average_freq=1,
                       flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                       num_major_iterations=1),
}


def _full_size_inputs(seed):
    rs = np.random.RandomState(seed)
    shape = (1, 2, 1024, 4096)
    vis = np.empty(shape, np.complex64)
    vis.real = rs.standard_normal(shape).astype(np.float32)
    vis.imag = rs.standard_normal(shape).astype(np.float32)
    vis.real[..., ::97] += 8.0                      # bad channels
    vis.real[:, :, ::211, :] += 6.0                 # bad times
    vis.real[0, 0, 100:140, 2000:2300] += 2.0       # broad-band block
    vis.real[0, 1, :, 3000:3060] += 1.5             # faint wide feature (broad windows)
    idx = rs.randint(0, vis.size, 500)
    vis.real.reshape(-1)[idx] += 50.0
    vis.real.reshape(-1)[rs.randint(0, vis.size, 50)] = np.nan
    flags = np.zeros(shape, np.bool_)
    flags[..., ::50] = True
    flags[0, 1, 300:320, :] = True
    flags[0, 0, :, 1200:1900] = True                # 700 flagged channels: wider than any filter support but r = 277 x 4
    return vis, flags


@pytest.mark.parametrize("name", sorted(SHIPPED_KWARGS))
def test_full_size_shipped_kwargs_vs_oracle(gpu, oracle, name):
    """One 1024 x 4096 window pair per shipped parameter set through the DEFAULT kernel
    routes, flags and the six last-iteration intermediates bit-for-bit against the
    canonical oracle (flagging.py:422-466, 610-681)."""
    vis, flags = _full_size_inputs({"stage1": 7, "very_broad": 8}.get(name, 9 + len(name)))
    kw = SHIPPED_KWARGS[name]
    dbg = {}
    out = gpu.sum_threshold_flagger(vis, flags, _debug=dbg, **kw)
    exp, inter = oracle.sum_threshold_flagger(vis, flags, n_threads=2, dump=True, **kw)
    report = []
    for k in ("spec_resid", "background", "residual"):
        bad = int((~_same_f32(inter[k], dbg[k])).sum())
        if bad:
            report.append("%s: %d float32 words differ" % (k, bad))
    for k in ("spec_flags", "time_flags", "freq_flags"):
        bad = int((inter[k].astype(bool) != dbg[k].reshape(inter[k].shape)).sum())
        if bad:
            report.append("%s: %d flags differ" % (k, bad))
    bad = int((out != exp).sum())
    if bad:
        report.append("out: %d of %d flags differ" % (bad, out.size))
    assert not report, "%s: %s" % (name, "; ".join(report))
    assert 0 < out.mean() < 1


def test_inf_nan_visibility_is_flagged_on_the_cached_amplitude_route(gpu, oracle):
    """flagging.py:777-781: the final `isnan(in_data)` is of the complex visibility -- either part NaN.  A sample
    with one infinite and one NaN part has amplitude +inf (C99 hypot), so only that final test flags it.  F is a
    multiple of 16 (the cached-amplitude route, k_amplitude4's NaN bitmap); the samples are pre-flagged so that the
    infinite amplitude never enters a running sum (unflagged infinities are undefined in the reference itself)."""
    rs = np.random.RandomState(5)
    shape = (2, 2, 32, 64)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 20] *= 7
    flags = rs.uniform(size=shape) < 0.03
    special = [((0, 0, 3, 5), complex(np.inf, np.nan)), ((0, 1, 9, 17), complex(np.nan, -np.inf)),
               ((1, 0, 30, 63), complex(np.nan, 1.0)), ((1, 1, 0, 0), complex(2.0, np.nan)),
               ((1, 1, 31, 48), complex(np.nan, np.nan))]
    for idx, z in special:
        vis.real[idx] = z.real
        vis.imag[idx] = z.imag
        flags[idx] = True
    for kw in (dict(num_major_iterations=1), dict(num_major_iterations=3, background_iterations=2)):
        exp = oracle.sum_threshold_flagger(vis, flags, **kw)
        got = gpu.sum_threshold_flagger(vis, flags, **kw)
        assert all(exp[idx] for idx, _ in special)
        assert np.array_equal(got, exp), "%d flags differ" % (got != exp).sum()


def test_two_stream_schedule_vs_oracle(gpu, oracle):
    """TRI_SUBSTREAMS=1: the window set is split in two halves on two internal streams (the schedule SKA-sized
    windows take by default); odd window counts, several batches per half, numpy and device inputs."""
    import torch
    rs = np.random.RandomState(77)
    shape = (5, 1, 48, 160)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 40] *= 8
    vis[2, 0, 17] *= 5
    flags = rs.uniform(size=shape) < 0.04
    kw = dict(num_major_iterations=2, background_iterations=2)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    old = os.environ.get("TRI_SUBSTREAMS")
    try:
        os.environ["TRI_SUBSTREAMS"] = "1"
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
        vd, fd = torch.from_numpy(vis).cuda(), torch.from_numpy(flags).cuda()
        out_d = [gpu.sum_threshold_flagger(vd, fd, **kw) for _ in range(3)]     # back to back, unsynchronised
        cnt = [int(o.sum().item()) for o in out_d]
    finally:
        if old is None:
            os.environ.pop("TRI_SUBSTREAMS", None)
        else:
            os.environ["TRI_SUBSTREAMS"] = old
    assert np.array_equal(out, exp)
    assert all(np.array_equal(o.cpu().numpy(), exp) for o in out_d) and cnt == [int(exp.sum())] * 3


def test_size_independent_properties_at_slab_scale(gpu):
    """Properties that need no oracle, on a multi-GB slab: determinism,
    independence of a window's result from its batch neighbours, all-flagged
    input -> no output flags (tests/test_flagging.py:619-630 of the reference),
    NaN samples always flagged."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    shape = (12, 4, 1024, 4096)
    vis = torch.complex(torch.randn(shape, generator=g, device="cuda"),
                        torch.randn(shape, generator=g, device="cuda"))
    vis.real[..., 1000] += 9.0
    vis.real[3, 1, 17, :] += 7.0
    nan_idx = torch.randint(0, vis.numel(), (200,), generator=g, device="cuda")
    torch.view_as_real(vis).view(-1, 2)[nan_idx, 0] = float("nan")
    flags = torch.zeros(shape, dtype=torch.bool, device="cuda")
    flags[..., ::64] = True
    flags[5] = True                                   # a fully flagged baseline
    kw = dict(num_major_iterations=2)
    a = gpu.sum_threshold_flagger(vis, flags, **kw)
    b = gpu.sum_threshold_flagger(vis, flags, **kw)
    assert torch.equal(a, b)
    sub = gpu.sum_threshold_flagger(vis[7:9].contiguous(), flags[7:9].contiguous(), **kw)
    assert torch.equal(sub, a[7:9])
    nanmask = torch.isnan(vis.real) | torch.isnan(vis.imag)
    assert bool(a[nanmask].all())
    allflag = a[5] & ~nanmask[5]
    assert not bool(allflag.any())


def test_concurrent_calls_from_threads(gpu, oracle):
    """The reference is called from a dask ThreadPool (app.py:266-271): the
    library must be re-entrant (per-thread workspaces, no global state)."""
    import threading
    rs = np.random.RandomState(21)
    shape = (2, 2, 48, 96)
    cases = []
    for k in range(6):
        vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
        vis[..., 10 + 7 * k] *= 6
        flags = rs.uniform(size=shape) < 0.03
        kw = dict(num_major_iterations=1 + k % 2, background_iterations=1 + k % 3)
        cases.append((vis, flags, kw, oracle.sum_threshold_flagger(vis, flags, **kw)))
    results = [None] * len(cases)
    errors = []

    def work(i):
        try:
            for _ in range(3):
                results[i] = gpu.sum_threshold_flagger(cases[i][0], cases[i][1], **cases[i][2])
        except Exception as e:   # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, c in enumerate(cases):
        assert np.array_equal(results[i], c[3]), "thread %d" % i


def test_ska_shaped_window_vs_oracle(gpu, oracle):
    """BASELINE config 5 geometry (512 time x 65536 chan): long frequency lines,
    6554-channel chunks (the workgroup median kernel serves the per-(time,
    chunk) MADs), bit-exact against the oracle."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(55)
    shape = (1, 2, 512, 65536)
    re = torch.randn(shape, generator=g, device="cuda")
    im = torch.randn(shape, generator=g, device="cuda")
    re[..., ::1013] += 8.0
    re[:, :, ::101, :] += 5.0
    re[0, 1, 100:120, 30000:31000] += 2.5
    re.view(-1)[torch.randint(0, re.numel(), (2000,), generator=g, device="cuda")] += 50.0
    vis = torch.complex(re, im)
    flags = torch.zeros(shape, dtype=torch.bool, device="cuda")
    flags[..., ::64] = True
    kw = dict(num_major_iterations=1)
    out = gpu.sum_threshold_flagger(vis, flags, **kw).cpu().numpy()
    exp = oracle.sum_threshold_flagger(vis.cpu().numpy(), flags.cpu().numpy(), n_threads=2, **kw)
    nbad = int((out != exp).sum())
    assert nbad == 0, "%d of %d flags differ" % (nbad, out.size)


def test_ska_shaped_window_stage1_kwargs_vs_oracle(gpu, oracle):
    """The same geometry with the stage-1 parameter set of default.yaml (radii [54, 43] ... [10, 8]) and two major
    iterations: the routes a 64-window SKA slab takes at every radius -- the eight-wave stage pipeline from r = 8 on
    (few, long lines), the four-wave spectrum pipeline at r = 43 ... 8 on 65536-channel lines, segment-wise NaN
    interpolation -- flags and the six last-iteration intermediates bit for bit."""
    rs = np.random.RandomState(56)
    shape = (1, 2, 512, 65536)
    vis = np.empty(shape, np.complex64)
    vis.real = rs.standard_normal(shape).astype(np.float32)
    vis.imag = rs.standard_normal(shape).astype(np.float32)
    vis.real[..., ::1013] += 8.0
    vis.real[:, :, ::101, :] += 5.0
    vis.real[0, 1, 100:120, 30000:31000] += 2.5
    vis.real.reshape(-1)[rs.randint(0, vis.size, 2000)] += 50.0
    vis.real.reshape(-1)[rs.randint(0, vis.size, 40)] = np.nan
    flags = np.zeros(shape, np.bool_)
    flags[..., ::64] = True
    flags[0, 0, :, 20000:20600] = True              # wider than 4 r = 172 channels: NaN background to interpolate
    kw = dict(SHIPPED_KWARGS["stage1"], num_major_iterations=2)
    _compare(gpu, oracle, vis, flags, kw, "ska stage1")


def test_spectrum_exact_row_filter_taken_by_a_later_iteration_only(gpu, oracle):
    """the spectrum background picks the exact row filter (K4x) per iteration.  With 5400 channels and
    spike_width_freq = 64 the first radius (277: padded line 6508 > the longest LDS line) takes the column filter and the
    second (221: 6284) takes K4x -- which then needs the row copy of the spectra that only the first iteration used to make."""
    rs = np.random.RandomState(77)
    shape = (1, 2, 16, 5400)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis.real += 5.0 + 3.0 * np.sin(np.arange(shape[3]) / 300.0).astype(np.float32)
    vis[..., ::131] *= 6
    flags = rs.uniform(size=shape) < 0.03
    kw = dict(num_major_iterations=2, spike_width_freq=64.0, background_iterations=5)
    _compare(gpu, oracle, vis, flags, kw, "late exact rows")


def test_flag_dtypes_and_non_contiguous_inputs(gpu, oracle):
    """flags of any integer type (non-zero = flagged, flagging.py:833-835;
    cf. the reference's TestAsbool, tests/test_flagging.py:12-34) and
    non-contiguous views give the same result as contiguous bool input."""
    import torch
    rs = np.random.RandomState(8)
    shape = (2, 2, 32, 64)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 9] *= 7
    flags = rs.uniform(size=shape) < 0.05
    kw = dict(num_major_iterations=1)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    for dt, val in ((np.uint8, 1), (np.int8, -1), (np.uint16, 300), (np.int32, 7), (np.int64, 2**40)):
        f = (flags.astype(dt) * dt(val))
        assert np.array_equal(gpu.sum_threshold_flagger(vis, f, **kw), exp), dt
    # non-contiguous numpy views
    big_v = np.zeros((2, 2, 32, 128), np.complex64)
    big_f = np.zeros((2, 2, 32, 128), bool)
    big_v[..., ::2] = vis
    big_f[..., ::2] = flags
    assert np.array_equal(gpu.sum_threshold_flagger(big_v[..., ::2], big_f[..., ::2], **kw), exp)
    # permuted torch tensors on the device
    vt = torch.from_numpy(np.ascontiguousarray(vis.transpose(1, 0, 2, 3))).cuda().permute(1, 0, 2, 3)
    ft = torch.from_numpy(np.ascontiguousarray(flags.transpose(1, 0, 2, 3))).cuda().permute(1, 0, 2, 3)
    assert not vt.is_contiguous()
    assert np.array_equal(gpu.sum_threshold_flagger(vt, ft, **kw).cpu().numpy(), exp)
    # float32 amplitudes and float64 amplitudes
    amp = np.abs(vis).astype(np.float32)
    e2 = oracle.sum_threshold_flagger(amp, flags, **kw)
    assert np.array_equal(gpu.sum_threshold_flagger(amp, flags, **kw), e2)
    assert np.array_equal(gpu.sum_threshold_flagger(amp.astype(np.float64), flags, **kw), e2)
    # complex128 (the reference takes "real or complex" of either width, flagging.py:830-835): its float64
    # amplitude, rounded once to float32 (flagging.py:856-859) -- equal to the complex64 result wherever the
    # float64 hypot of the widened parts rounds to the float32 hypot, which holds for these samples
    out128 = gpu.sum_threshold_flagger(vis.astype(np.complex128), flags, **kw)
    amp64 = np.hypot(vis.real.astype(np.float64), vis.imag.astype(np.float64)).astype(np.float32)
    assert np.array_equal(out128, oracle.sum_threshold_flagger(amp64, flags, **kw))
    with pytest.raises(TypeError):
        gpu.sum_threshold_flagger(vis.real.astype(np.int32), flags, **kw)


def test_wide_input_types_with_channel_averaging(gpu, oracle):
    """float64 / complex128 visibilities with average_freq > 1 (missing item 5; flagging.py:856-859): the
    float64 amplitude is added to the float32 accumulator in float64 and rounded at every step -- not the same as
    rounding the amplitude first.  Against the oracle, whose wide path is pinned bit for bit by the reference-generated
    fixtures G14 / G14b (tests/test_oracle_golden.py); numpy and device inputs; NaN parts and pre-flagged infinities."""
    import torch
    rs = np.random.RandomState(41)
    shape = (2, 2, 36, 90)
    for avg, wf in ((2, [2, 4, 8, 16]), (3, [3, 6, 12]), (4, [4, 8, 16])):
        amp = (np.abs(rs.standard_normal(shape) * 0.7 + 5.0) * (1.0 + 1e-9 * rs.standard_normal(shape))).astype(np.float64)
        amp[..., 30] += 6.0
        amp[0, 1, 9, :] += 4.0
        flags = rs.uniform(size=shape) < 0.04
        amp[1, 0, 5, 7] = np.nan
        kw = dict(average_freq=avg, windows_freq=wf, freq_chunks=3, num_major_iterations=2, freq_extend=avg + 2)
        exp = oracle.sum_threshold_flagger(amp, flags, **kw)
        assert np.array_equal(gpu.sum_threshold_flagger(amp, flags, **kw), exp), avg
        got = gpu.sum_threshold_flagger(torch.from_numpy(amp).cuda(), torch.from_numpy(flags).cuda(), **kw)
        assert np.array_equal(got.cpu().numpy(), exp), avg
        # the same amplitudes split over real / imaginary parts (one part zero: |z| is exact in any hypot)
        z = np.where(rs.uniform(size=shape) < 0.5, amp + 0j, 1j * amp).astype(np.complex128)
        z[0, 0, 3, 11] = complex(np.inf, np.nan)
        f2 = flags.copy()
        f2[0, 0, 3, 11] = True
        expz = oracle.sum_threshold_flagger(z, f2, **kw)
        assert expz[0, 0, 3, 11]
        assert np.array_equal(gpu.sum_threshold_flagger(z, f2, **kw), expz), avg
    # float32-rounded amplitudes give a different answer on some sample of such data: the float64 path is not a no-op
    sums32 = (amp.astype(np.float32)[..., 0::2] + amp.astype(np.float32)[..., 1::2])
    sums64 = (amp[..., 0::2] + amp[..., 1::2]).astype(np.float32)
    assert (sums32 != sums64).any()


def _random_case(rs):
    """A random small window set + kwargs inside the reference's contract."""
    T = int(rs.choice([1, 2, 5, 16, 31, 48, 64, 100, 130, 257, 300]))
    F = int(rs.choice([1, 3, 16, 33, 64, 97, 128, 200, 520, 640]))
    shape = (int(rs.randint(1, 4)), int(rs.randint(1, 3)), T, F)
    avg = int(rs.choice([1, 1, 1, 2, 3]))
    kw = dict(
        outlier_nsigma=float(rs.choice([3.0, 4.5, 10.0])),
        windows_time=[[1, 2, 4, 8], [1, 2, 4, 8], [1, 3], [2, 8, 16]][int(rs.randint(0, 4))],
        windows_freq=[[1, 2, 4, 8], [1, 2, 4, 8], [1, 5, 9], [4]][int(rs.randint(0, 4))],
        background_reject=float(rs.choice([2.0, 3.5])),
        background_iterations=int(rs.choice([0, 1, 1, 2, 3])),
        spike_width_time=float(rs.choice([0.3, 3.0, 6.5, 12.5, 20.0, 40.0])),
        spike_width_freq=float(rs.choice([0.3, 2.0, 10.0, 18.0, 30.0])),
        time_extend=int(rs.randint(0, 5)), freq_extend=int(rs.randint(0, 5)),
        freq_chunks=int(rs.choice([1, 2, 5, 10])), average_freq=avg,
        flag_all_time_frac=float(rs.choice([0.3, 0.6, 1.0])), flag_all_freq_frac=float(rs.choice([0.4, 0.8, 1.0])),
        rho=float(rs.choice([1.3, 1.5])), num_major_iterations=int(rs.choice([1, 2, 3])))
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    if rs.uniform() < 0.7 and F > 4:
        vis[..., int(rs.randint(0, F))] *= rs.choice([4.0, 9.0, 30.0])
    if rs.uniform() < 0.7 and T > 4:
        vis[:, :, int(rs.randint(0, T)), :] += rs.choice([3.0, 7.0])
    if rs.uniform() < 0.3:
        vis[rs.uniform(size=shape) < 0.01] = np.nan
    if rs.uniform() < 0.3:
        vis[rs.uniform(size=shape) < 0.01] = 0
    flags = rs.uniform(size=shape) < rs.choice([0.0, 0.02, 0.3])
    if rs.uniform() < 0.2:
        flags[:, :, :, : max(1, F // 3)] = True
    return vis, flags, kw


@pytest.mark.parametrize("block", range(10))
def test_random_cases_vs_oracle(gpu, oracle, block):
    """Seeded sweep over shapes and kwargs (radii 0..69, odd extents, NaNs, zeros,
    heavy pre-flagging, channel averaging, arbitrary window lists): flags bit-exact."""
    rs = np.random.RandomState(4242 + block)
    for k in range(20):
        vis, flags, kw = _random_case(rs)
        try:
            exp = oracle.sum_threshold_flagger(vis, flags, **kw)
        except ValueError:
            # window lists the reference fails on (no window fits the axis, or a frequency window
            # averaged down to size 0; flagging.py:630, 663): the product must refuse them too
            with pytest.raises(ValueError):
                gpu.sum_threshold_flagger(vis, flags, **kw)
            continue
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
        assert np.array_equal(out, exp), "block %d case %d shape %s kw %s: %d flags differ" % (
            block, k, vis.shape, kw, int((out != exp).sum()))


def test_large_windows_random_kwargs_vs_oracle(gpu, oracle):
    """Windows of the production sizes with RANDOM spike widths / iteration counts (radii 2 ... 64 on both axes): every
    radius class of the filter routes -- register delay lines, stage pipelines with blocks of 8 / 16 and the deep FIFOs,
    LDS delay lines -- against the oracle at line lengths the small random cases do not reach."""
    rs = np.random.RandomState(777)
    for case in range(8):
        t, f = [(1024, 4096), (512, 2048), (1024, 1024), (256, 8192)][case % 4]
        shape = (2, 1, t, f)
        kw = dict(background_iterations=int(rs.randint(1, 6)), spike_width_time=float(rs.uniform(3, 14)),
                  spike_width_freq=float(rs.uniform(3, 12)), num_major_iterations=int(rs.randint(1, 3)),
                  background_reject=float(rs.choice([2.0, 3.0])), freq_chunks=int(rs.choice([4, 10])))
        vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
        vis[..., rs.randint(0, f, 8)] *= 6
        vis[:, :, rs.randint(0, t, 4), :] += 4
        vis[rs.uniform(size=shape) < 1e-4] = np.nan
        flags = rs.uniform(size=shape) < 0.03
        exp = oracle.sum_threshold_flagger(vis, flags, **kw)
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
        assert np.array_equal(out, exp), "case %d shape %s kw %s: %d flags differ" % (case, shape, kw, int((out != exp).sum()))


ALT_PATH_SCRIPT = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import tricolour_amd
from oracle import oracle
rs = np.random.RandomState(77)
bad = 0
for shape, kw in (((2, 1, 64, 128), dict(num_major_iterations=2)),
                  ((1, 2, 100, 144), dict(num_major_iterations=2, background_iterations=2, spike_width_time=20.0,
                                          spike_width_freq=15.0)),
                  ((1, 1, 52, 300), dict(num_major_iterations=1, windows_freq=[1, 2, 4, 8, 16])),
                  # final_st_very_broad's frequency windows on three chunks (the stage pipeline K7p, windows wider than a chunk's share of the line)
                  ((1, 1, 24, 700), dict(num_major_iterations=1, windows_freq=[32, 48, 64, 128], freq_chunks=3)),
                  # frequency radii 110 / 55: the exact row filter K4x (short chunks; long chunks on the 2048-channel lines)
                  ((1, 2, 40, 512), dict(num_major_iterations=2, background_iterations=2, spike_width_freq=64.0)),
                  ((1, 1, 16, 2048), dict(num_major_iterations=1, background_iterations=2, spike_width_freq=64.0))):
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., shape[3] // 3] *= 8
    vis[:, :, shape[2] // 2, :] += 5
    vis[0, 0, 3, 5] = np.nan
    flags = rs.uniform(size=shape) < 0.03
    out = tricolour_amd.sum_threshold_flagger(vis, flags, **kw)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    bad += int((out != exp).sum())
print("DIFF", bad)
'''


@pytest.mark.parametrize("knob", ["TRI_NO_AMPL_CACHE", "TRI_FILTER_NO_FUSED_DIV", "TRI_FILTER_NO_TIN", "TRI_FILTER_NO_LANE4",
                                  "TRI_NO_PACKED_FLAGS", "TRI_FILTER_MULTIPASS", "TRI_ST_GENERIC", "TRI_ST_REGISTER",
                                  "TRI_FILTER_DIRECT_FT", "TRI_FILTER_NO_REGRING", "TRI_FILTER_NO_REGRING_F", "TRI_MEDIAN_3PASS",
                                  "TRI_NO_FUSED_REJECT", "TRI_NO_FT_SPEC_OR", "TRI_NO_FUSED_BEGIN", "TRI_NO_FUSED_DILATE", "TRI_SPEC_NO_PIPE", "TRI_INTERP_ONE_PASS", "TRI_MEDIAN_NO_PREDICT", "TRI_FILTER_NO_PIPE_T", "TRI_FILTER_NO_PIPE_F", "TRI_ST_NO_PIPE",
                                  "TRI_FILTER_NO_EXACT", "TRI_BOXX_NTI=256", "TRI_FILTER_NO_TF_REJECT", "TRI_FILTER_PIPE_T_B8=0", "TRI_FILTER_PIPE_F_B8=0", "TRI_NO_FUSED_OR", "TRI_FILTER_NO_BOXW", "TRI_ST_NO_PANEL", "TRI_FUSED_MEDREJ", "TRI_FUSED_MEDREJ=1;TRI_MEDREJ_FORCE_FALLBACK=1", "TRI_NO_TILE_MEDREJ", "TRI_MEDIAN_WAVE_OLD", "TRI_MEDREJ_FORCE_FALLBACK",
                                  "DEFAULT_ROUTES"])
def test_alternate_kernel_paths(gpu, knob):
    """Every fallback / A-B path selectable through an environment knob (read once
    per process, hence the subprocess) stays bit-exact against the oracle."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for one in knob.split(";"):
        name, _, value = one.partition("=")
        env[name] = value or "1"
    p = subprocess.run([sys.executable, "-c", ALT_PATH_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DIFF 0" in p.stdout, p.stdout + p.stderr
