"""Stokes intensity transforms: tricolour_amd.stokes against the
reference's own output (fixture G13, tests/golden/make_golden_stokes.py; the
un-jitted reference computes in complex64, hence 1e-6), against the oracle's
restatement with numba's complex128 typing (exact up to one float32 ulp) and
against the closed forms of tricolour/tests/test_stokes.py."""
import json
import os
import subprocess
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CONDA_PY = "/opt/conda/bin/python3.9"


def _cases():
    d = np.load(os.path.join(GOLDEN, "G13_stokes.npz"))
    return d, json.loads(str(d["cases"]))


def _terms(case, keep):
    return tuple((v[0], v[1], complex(v[2][0], v[2][1]), v[3], v[4]) for s, v in case["map"].items() if keep(s))


def test_stokes_corr_map_matches_reference():
    from tricolour_amd import stokes
    _, cases = _cases()
    for case in cases:
        got = stokes.stokes_corr_map(case["codes"])
        assert list(got) == list(case["map"])                      # same Stokes, same order
        for s, v in case["map"].items():
            assert got[s] == (v[0], v[1], complex(v[2][0], v[2][1]), v[3], v[4])
    # a two-correlation dataset only forms I and Q (linear) or I and V (circular)
    T = stokes.STOKES_TYPES
    assert set(stokes.stokes_corr_map([T["XX"], T["YY"]])) == {"I", "Q"}
    assert set(stokes.stokes_corr_map([T["RR"], T["LL"]])) == {"I", "V"}
    assert stokes.stokes_corr_map([T["XX"]]) == {}


def test_oracle_restatement_matches_reference_fixture(oracle):
    d, cases = _cases()
    vis = d["vis"]
    for k, case in enumerate(cases):
        pol, unpol = _terms(case, lambda s: s != "I"), _terms(case, lambda s: s == "I")
        every = _terms(case, lambda s: True)
        for name, got in (("pol", oracle.polarised_intensity(vis, pol)), ("total", oracle.polarised_intensity(vis, every)),
                          ("unpol", oracle.unpolarised_intensity(vis, unpol, pol))):
            exp = d["%s_%d" % (name, k)]
            assert got.shape == exp.shape == vis.shape[:2] + (1,) and got.dtype == vis.dtype
            scale = np.abs(d["total_%d" % k]).real + 1e-30          # unpol is a difference: scale by the total power
            assert np.all(np.abs(got - exp) <= 2e-6 * scale), name


def test_oracle_closed_forms_of_reference_tests(oracle):
    from tricolour_amd import stokes
    vis = np.asarray([[[1 + 1j, 2 + 2j, 3 + 3j, 4 + 4j]]], np.complex128)
    for names in (["YX", "XX", "XY", "YY"], ["XX", "XY", "YX", "YY"], ["RR", "RL", "LR", "LL"], ["RL", "RR", "LL", "LR"]):
        cmap = stokes.stokes_corr_map([stokes.STOKES_TYPES[n] for n in names])
        pol_terms = tuple(v for s, v in cmap.items() if s != "I")
        unpol_terms = tuple(v for s, v in cmap.items() if s == "I")
        pol = sum(np.abs(a * (s1 * vis[0, 0, c1] + s2 * vis[0, 0, c2])) ** 2 for c1, c2, a, s1, s2 in pol_terms)
        unpol = sum(np.abs(a * (s1 * vis[0, 0, c1] + s2 * vis[0, 0, c2])) for c1, c2, a, s1, s2 in unpol_terms)
        assert np.allclose(oracle.polarised_intensity(vis, pol_terms), np.sqrt(pol))
        assert np.allclose(oracle.unpolarised_intensity(vis, unpol_terms, pol_terms), unpol - np.sqrt(pol))


SCRIPT = textwrap.dedent('''
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import dask.array as da
    import tricolour_amd.dask_wrappers as dw
    from tricolour_amd.stokes import stokes_corr_map, STOKES_TYPES
    from oracle import oracle

    dw.amd_polarised_intensity = oracle.polarised_intensity        # no GPU here
    dw.amd_unpolarised_intensity = oracle.unpolarised_intensity
    rs = np.random.RandomState(0)
    vis_np = (rs.standard_normal((30, 16, 4)) + 1j * rs.standard_normal((30, 16, 4))).astype(np.complex64)
    vis = da.from_array(vis_np, chunks=((10, 20), (16,), (4,)))
    cmap = stokes_corr_map([STOKES_TYPES[n] for n in ("XX", "XY", "YX", "YY")])
    pol = tuple(v for s, v in cmap.items() if s != "I")
    unpol = tuple(v for s, v in cmap.items() if s == "I")
    out = dw.polarised_intensity(vis, pol)
    assert out.chunks == ((10, 20), (16,), (1,)) and out.dtype == vis.dtype
    assert np.array_equal(out.compute(scheduler="single-threaded"), oracle.polarised_intensity(vis_np, pol))
    out = dw.unpolarised_intensity(vis, unpol, pol)
    assert out.shape == (30, 16, 1)
    assert np.array_equal(out.compute(scheduler="single-threaded"), oracle.unpolarised_intensity(vis_np, unpol, pol))
    print("OK")
''')


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with dask in this image")
def test_dask_wrappers():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([CONDA_PY, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr


def _ulp32(a, b):
    a = np.ascontiguousarray(a.real.astype(np.float32)).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b.real.astype(np.float32)).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


@pytest.mark.gpu
def test_gpu_intensities(gpu, oracle):
    import torch
    from tricolour_amd import stokes
    d, cases = _cases()
    vis = d["vis"]
    for k, case in enumerate(cases):
        pol, unpol = _terms(case, lambda s: s != "I"), _terms(case, lambda s: s == "I")
        every = _terms(case, lambda s: True)
        for name, got, exp in (("pol", stokes.polarised_intensity(vis, pol), oracle.polarised_intensity(vis, pol)),
                               ("total", stokes.polarised_intensity(vis, every), oracle.polarised_intensity(vis, every)),
                               ("unpol", stokes.unpolarised_intensity(vis, unpol, pol),
                                oracle.unpolarised_intensity(vis, unpol, pol))):
            assert got.shape == exp.shape and got.dtype == np.complex64 and not got.imag.any()
            if name != "unpol":    # device hypot vs libm hypot: at most the last float32 bit, almost never
                u = _ulp32(got, exp)
                assert u.max() <= 1 and (u == 0).mean() > 0.999, (name, u.max())
            ref = d["%s_%d" % (name, k)]
            scale = np.abs(d["total_%d" % k]).real + 1e-30
            assert np.all(np.abs(got - ref) <= 2e-6 * scale), name      # the reference's own output
            assert np.all(np.abs(got - exp) <= 2e-7 * scale), name
    # device tensors stay on the device; complex128 works as in the reference's tests
    t = torch.from_numpy(vis).cuda()
    out = stokes.polarised_intensity(t, _terms(cases[0], lambda s: s != "I"))
    assert out.is_cuda and tuple(out.shape) == vis.shape[:2] + (1,)
    v128 = np.asarray([[[1 + 1j, 2 + 2j, 3 + 3j, 4 + 4j]]], np.complex128)
    cmap = stokes.stokes_corr_map([stokes.STOKES_TYPES[n] for n in ("RL", "RR", "LL", "LR")])
    pol = tuple(v for s, v in cmap.items() if s != "I")
    unpol = tuple(v for s, v in cmap.items() if s == "I")
    assert np.allclose(stokes.polarised_intensity(v128, pol), oracle.polarised_intensity(v128, pol), rtol=1e-15)
    assert stokes.unpolarised_intensity(v128, unpol, pol).dtype == np.complex128
    with pytest.raises(ValueError):
        stokes.unpolarised_intensity(vis, (), pol)
    with pytest.raises(ValueError):
        stokes.unpolarised_intensity(vis, unpol, ())
    with pytest.raises(TypeError):
        stokes.polarised_intensity(vis.real.copy(), pol)
    assert stokes.polarised_intensity(vis[:0], pol).shape == (0, vis.shape[1], 1)
