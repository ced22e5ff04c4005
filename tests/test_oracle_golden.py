"""Pins the oracle: (a) bit-for-bit against the committed outputs of the
reference itself (run un-jitted by tests/golden/make_golden.py), in the mode
that mirrors un-jitted NumPy-2 arithmetic; (b) the numba-canonical mode may
differ from those only in float32 intermediates at the documented D1/D2
sites, never in flags."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden


def _same_f32(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32).reshape(a.shape)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_equals_reference_bitwise(oracle, name):
    d, kw = load_golden(name)
    oracle.set_modes(oracle.POW_POWF, oracle.INTERP_F32)
    try:
        out, inter = oracle.sum_threshold_flagger(d["vis"], d["flags"], dump=True, **kw)
    finally:
        oracle.set_modes(oracle.POW_SQMUL, oracle.INTERP_F64)
    assert np.array_equal(out, d["out"])
    for k in ("spec_resid", "background", "residual"):
        assert _same_f32(inter[k], d["i_" + k]).all(), k
    for k in ("spec_flags", "time_flags", "freq_flags"):
        assert np.array_equal(inter[k].astype(bool), d["i_" + k].reshape(inter[k].shape)), k


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_canonical_mode_flags_equal(oracle, name):
    d, kw = load_golden(name)
    out, inter = oracle.sum_threshold_flagger(d["vis"], d["flags"], dump=True, **kw)
    assert np.array_equal(out, d["out"])
    same = _same_f32(inter["background"], d["i_background"])
    if name not in ("G2b_radius32.npz", "G4_preflagged.npz"):
        assert same.all()        # no D1 / D2 site touched
    else:
        # D1 (float32(65)**4 on both axes, then o / w) / D2 (float64 interpolation): a few ulp
        a = inter["background"].view(np.int32).astype(np.int64)
        b = d["i_background"].reshape(inter["background"].shape).view(np.int32).astype(np.int64)
        assert np.abs(a - b)[~same].max() <= 4 and (~same).any()


def test_threads_do_not_change_results(oracle):
    d, kw = load_golden("G1_defaults.npz")
    a = oracle.sum_threshold_flagger(d["vis"], d["flags"], n_threads=1, **kw)
    b = oracle.sum_threshold_flagger(d["vis"], d["flags"], n_threads=4, **kw)
    assert np.array_equal(a, b)


def test_hypotf_kat(oracle):
    d, _ = load_golden("G0_hypotf.npz")
    z = np.empty(d["re"].shape, np.complex64)
    z.real, z.imag = d["re"], d["im"]
    got = oracle.abs_c64(z)
    assert _same_f32(got, d["amp"]).all()
    # the device formula, evaluated on the host: (float)sqrt((double)re^2 + (double)im^2)
    re, im = d["re"].astype(np.float64), d["im"].astype(np.float64)
    with np.errstate(all="ignore"):
        alt = np.sqrt(re * re + im * im).astype(np.float32)
    alt[np.isinf(d["re"]) | np.isinf(d["im"])] = np.inf
    assert _same_f32(alt, d["amp"]).all()
