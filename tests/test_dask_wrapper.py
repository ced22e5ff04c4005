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
