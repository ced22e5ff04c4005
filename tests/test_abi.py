"""CPU-side checks of the boundary: the C-ABI library loads, exports every
symbol include/tricolour_amd.h declares, prepares parameters exactly like the
reference's plain-Python preamble (flagging.py:1156-1179), and the host mirror
keeps the reference's error behaviour.  No compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def test_library_exports_every_declared_symbol():
    from tricolour_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tricolour_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tri_[a-z0-9_]+)\s*\(", hdr))
    assert {"tri_sum_threshold_flagger", "tri_prepare_params", "tri_workspace_bytes",
            "tri_pack_data", "tri_unpack_data", "tri_last_error"} <= declared
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
    assert set(_lib.EXPORTS) == declared
    assert lib.tri_version() >= 100


CASES = [
    dict(ntime=1024, nchan=4096),
    dict(ntime=6, nchan=40, windows_freq=[32, 48, 64, 128], freq_chunks=3),
    dict(ntime=100, nchan=101, average_freq=2, windows_freq=[2, 4, 8, 16], freq_chunks=4),
    dict(ntime=5, nchan=7, windows_time=[8, 4, 2, 1, 2], windows_freq=[1.2, 2.0, 2.9, 8], freq_chunks=7),
    dict(ntime=300, nchan=65536, freq_chunks=13, average_freq=3, windows_freq=[3, 6, 12, 96]),
    dict(ntime=64, nchan=160, freq_chunks=1),
]


@pytest.mark.parametrize("case", CASES)
def test_prepare_params_matches_reference_preamble(oracle, case):
    from tricolour_amd import flagging
    kw = dict(case)
    T, F = kw.pop("ntime"), kw.pop("nchan")
    p = flagging.prepare_params(T, F, **kw)
    full = dict(oracle.DEFAULTS)
    full.update(kw)
    wt, wf, ce, fa = oracle.prepare(F, T, full["windows_time"], full["windows_freq"],
                                    full["freq_chunks"], full["average_freq"])
    assert list(p.windows_time)[:p.n_windows_time] == list(wt)
    assert list(p.windows_freq)[:p.n_windows_freq] == list(wf)
    assert [p.chunk_ends[i] for i in range(p.n_chunk_ends)] == list(ce)


def test_prepare_params_matches_numpy_linspace():
    from tricolour_amd import flagging
    for fa in (1, 7, 100, 409, 4096, 65536, 12345):
        for chunks in (1, 3, 10, 17):
            p = flagging.prepare_params(16, fa, freq_chunks=chunks)
            exp = np.linspace(0, fa, chunks + 1).astype(np.int64)
            assert [p.chunk_ends[i] for i in range(p.n_chunk_ends)] == list(exp)


def test_zero_window_is_a_value_error():
    """average_freq=2 with the default windows gives a size-0
    window; the reference dies with a broadcasting ValueError."""
    from tricolour_amd import flagging
    with pytest.raises(ValueError):
        flagging.prepare_params(64, 64, average_freq=2)
    with pytest.raises(ValueError):
        flagging.prepare_params(64, 64, freq_chunks=0)


def test_workspace_bytes_monotone():
    from tricolour_amd import _lib, flagging
    p = flagging.prepare_params(1024, 4096)
    lib = _lib.lib()
    one = lib.tri_workspace_bytes(1, 1024, 4096, C.byref(p))
    four = lib.tri_workspace_bytes(4, 1024, 4096, C.byref(p))
    assert 0 < one < four <= 4 * one
    assert one < 400 * 2**20


def test_shape_mismatch_raises_value_error():
    import tricolour_amd
    vis = np.zeros((1, 1, 8, 8), np.complex64)
    with pytest.raises(ValueError):
        tricolour_amd.sum_threshold_flagger(vis, np.zeros((1, 1, 8, 9), bool))
    with pytest.raises(ValueError):
        tricolour_amd.sum_threshold_flagger(vis[0], np.zeros((1, 8, 8), bool))


@pytest.mark.skipif(has_gpu(), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback():
    import tricolour_amd
    vis = np.zeros((1, 1, 8, 8), np.complex64)
    with pytest.raises(RuntimeError):
        tricolour_amd.sum_threshold_flagger(vis, np.zeros(vis.shape, bool))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under tricolour_amd/ or the
    HIP sources may reference it."""
    pkg = os.path.join(ROOT, "tricolour_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("# oracle", ""), os.path.join(dirpath, f)
