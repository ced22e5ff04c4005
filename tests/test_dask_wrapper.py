"""Graph-level boundary: tricolour_amd.dask_wrappers.sum_threshold_flagger
must build the same blockwise layer as the reference's wrapper
(tricolour/dask_wrappers.py:23-46).  dask only exists in the image's conda
interpreter, so the check runs there in a subprocess (skipped where that
interpreter is missing); the per-block callable is replaced by a recorder, the
real kernels are covered by the GPU tests."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

CONDA_PY = "/opt/conda/bin/python3.9"

SCRIPT = textwrap.dedent('''
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import dask
    import dask.array as da
    import tricolour_amd.dask_wrappers as dw

    calls = []

    def fake_flagger(vis, flags, **kw):
        calls.append((vis.shape, flags.shape, vis.dtype, flags.dtype, dict(kw)))
        return np.logical_or(flags, np.abs(vis) > 2.5)

    dw.amd_sum_threshold_flagger = fake_flagger
    rs = np.random.RandomState(0)
    shape = (7, 2, 12, 16)
    vis_np = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    flag_np = rs.uniform(size=shape) < 0.1
    chunks = ((3, 2, 2), (2,), (12,), (16,))          # chunked along bl only (app.py:451)
    vis = da.from_array(vis_np, chunks=chunks)
    flag = da.from_array(flag_np, chunks=chunks)
    kw = dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], num_major_iterations=2)
    out = dw.sum_threshold_flagger(vis, flag, **kw)
    assert out.chunks == vis.chunks, out.chunks
    assert out.dtype == flag.dtype
    assert out.name.startswith('sum-threshold-flagger-')
    # same token recipe as the reference: tokenize(vis, flag, kwargs)
    assert out.name == 'sum-threshold-flagger-' + da.core.tokenize(vis, flag, kw)
    res = out.compute(scheduler='single-threaded')
    assert len(calls) == 3
    assert sorted(c[0][0] for c in calls) == [2, 2, 3]
    assert all(c[4] == kw for c in calls)
    assert np.array_equal(res, np.logical_or(flag_np, np.abs(vis_np) > 2.5))
    print('OK')
''')


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with dask in this image")
def test_dask_wrapper_graph():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([CONDA_PY, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr


STEPS_SCRIPT = textwrap.dedent('''
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import dask.array as da
    import tricolour_amd.dask_wrappers as dw
    from oracle import oracle

    # no GPU here: the numpy restatements stand in for the per-block kernels
    dw.amd_flag_nans_and_zeros = oracle.flag_nans_and_zeros
    dw.amd_flag_autos = oracle.flag_autos
    dw.amd_apply_static_mask = oracle.apply_static_mask
    rs = np.random.RandomState(2)
    nant = 4
    a1, a2 = np.triu_indices(nant, 0)
    ubl_np = np.stack([np.arange(a1.size), a1, a2], axis=1)
    shape = (ubl_np.shape[0], 2, 6, 32)
    vis_np = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis_np[1, 0, 2, 3] = 0
    vis_np[4, 1, 0, 7] = np.nan
    flag_np = rs.uniform(size=shape) < 0.1
    chunks = ((4, 3, 3), (2,), (6,), (32,))
    vis, flag = da.from_array(vis_np, chunks=chunks), da.from_array(flag_np, chunks=chunks)
    ubl = da.from_array(ubl_np, chunks=(chunks[0], 3))
    out = dw.flag_nans_and_zeros(vis, flag)
    assert out.chunks == flag.chunks and out.dtype == flag.dtype
    assert np.array_equal(out.compute(scheduler="single-threaded"), oracle.flag_nans_and_zeros(vis_np, flag_np))
    out = dw.flag_autos(flag, ubl)
    assert np.array_equal(out.compute(scheduler="single-threaded"), oracle.flag_autos(flag_np, [ubl_np]))
    antspos = rs.uniform(-500, 500, size=(nant, 3))
    freqs = np.linspace(1.0e9, 1.1e9, 32)
    widths = np.full(32, freqs[1] - freqs[0])
    masks = [np.array([freqs[5], freqs[20] + 1e5])[:, None]]
    # (the oracle stand-in takes the uv-range already parsed; the product parses the CASA string)
    out = dw.apply_static_mask(flag, ubl, antspos, masks, freqs, widths, accumulation_mode="or", uvrange=(0.0, np.inf))
    exp = oracle.apply_static_mask(flag_np, ubl_np, antspos, masks, freqs, widths, accumulation_mode="or",
                                   uvrange=(0.0, np.inf))
    got = out.compute(scheduler="single-threaded")
    assert got.dtype == flag_np.dtype and np.array_equal(got, exp) and got[:, :, :, 5].all()
    print("OK")
''')


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with dask in this image")
def test_strategy_step_wrappers():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([CONDA_PY, "-c", STEPS_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr
