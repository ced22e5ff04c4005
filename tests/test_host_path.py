"""The drop-in path as the dask graph drives it: host (numpy) blocks handed to the per-block callable of
tricolour_amd.dask_wrappers from a ThreadPool (tricolour/dask_wrappers.py:23-46, app.py:266-271), plus the
stream / workspace hand-over rules of host-side inputs."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_blockwise_layer_emulation_threadpool_vs_reference_capture(gpu):
    """G9 holds a block set the REFERENCE's own graph processed (pack_data -> dask_wrappers.sum_threshold_flagger
    -> unpack_data under oracle B).  dask cannot be imported next to torch here, so the blockwise layer is
    emulated: the wrapper module's per-block callable, three unequal `bl` chunks (chunking is along bl only,
    app.py:451), dask's ThreadPool -- real kernels -- and the blocks re-joined along bl."""
    from tricolour_amd import dask_wrappers as dw
    d, kw = load_golden("G9_config1_plumbing.npz")
    vis, flags = d["vis_windows"], d["flag_windows"]
    bounds = [0, 1, 4, 6]                                # chunks of 1, 3 and 2 baselines
    blocks = [(vis[a:b], flags[a:b]) for a, b in zip(bounds, bounds[1:])]
    with ThreadPoolExecutor(3) as pool:
        outs = list(pool.map(lambda vf: dw.amd_sum_threshold_flagger(vf[0], vf[1], **kw), blocks))
    assert all(isinstance(o, np.ndarray) and o.dtype == np.bool_ for o in outs)
    out = np.concatenate(outs, axis=0)
    assert np.array_equal(out, d["out_windows"])


def test_cpu_tensor_input_and_mixed_streams(gpu, oracle):
    """host inputs that are not numpy (CPU torch tensors) come back as a device tensor produced on
    the per-thread side stream -- it must be safe to use on the caller's stream at once; and a device-tensor
    call followed by a numpy call (two streams in flight from one thread) must not share scratch."""
    import torch
    rs = np.random.RandomState(21)
    shape = (3, 2, 96, 160)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 50] *= 7
    flags = rs.uniform(size=shape) < 0.04
    kw = dict(num_major_iterations=2)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    vis2 = np.ascontiguousarray(vis[::-1])
    exp2 = oracle.sum_threshold_flagger(vis2, flags, **kw)
    # CPU tensors in -> device tensor out, consumed immediately on the current stream
    out_t = gpu.sum_threshold_flagger(torch.from_numpy(vis), torch.from_numpy(flags), **kw)
    assert out_t.is_cuda
    cnt = out_t.sum()                                    # a kernel on the caller's stream, no explicit sync
    assert int(cnt.item()) == int(exp.sum())
    assert np.array_equal(out_t.cpu().numpy(), exp)
    # asynchronous device-tensor call on the current stream, then a numpy call (side stream) before syncing
    vd, fd = torch.from_numpy(vis).cuda(), torch.from_numpy(flags).cuda()
    for _ in range(3):
        o1 = gpu.sum_threshold_flagger(vd, fd, **kw)     # not synchronised
        o2 = gpu.sum_threshold_flagger(vis2, flags, **kw)
        assert np.array_equal(o2, exp2)
        assert np.array_equal(o1.cpu().numpy(), exp)


def test_wide_input_types(gpu, oracle):
    """The reference takes real or complex input of either width (flagging.py:830-835, 856-859): float64
    amplitudes and complex128 visibilities that are exactly representable in the narrow type give the
    narrow type's flags."""
    rs = np.random.RandomState(8)
    shape = (2, 1, 64, 96)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 30] *= 9
    vis[0, 0, 5, 7] = np.nan
    flags = rs.uniform(size=shape) < 0.05
    kw = dict(num_major_iterations=2)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    out64 = gpu.sum_threshold_flagger(vis, flags, **kw)
    assert np.array_equal(out64, exp)
    amp = np.abs(rs.standard_normal(shape)).astype(np.float32) + 2
    amp[..., 11] *= 8
    expa = oracle.sum_threshold_flagger(amp, flags, **kw)
    assert np.array_equal(gpu.sum_threshold_flagger(amp.astype(np.float64), flags, **kw), expa)
    # complex128 whose amplitude is exact in float64: real-only and imaginary-only samples
    z = np.zeros(shape, np.complex128)
    z.real[:, :, ::2] = amp[:, :, ::2]
    z.imag[:, :, 1::2] = -amp[:, :, 1::2].astype(np.float64)
    assert np.array_equal(gpu.sum_threshold_flagger(z, flags, **kw), expa)


def test_numpy_result_paths_agree(gpu, oracle, monkeypatch):
    """numpy in / numpy out: the result through the pinned staging buffer (default), straight into the fresh
    array (results above the staging limit, or pinning unavailable) and with the link turns switched off are
    the same flags -- from several threads at once."""
    from concurrent.futures import ThreadPoolExecutor
    from tricolour_amd import flagging
    rs = np.random.RandomState(3)
    shape = (4, 2, 48, 96)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 40] *= 7
    flags = rs.uniform(size=shape) < 0.04
    exp = oracle.sum_threshold_flagger(vis, flags, num_major_iterations=2)

    def run(_):
        out = gpu.sum_threshold_flagger(vis, flags, num_major_iterations=2)
        assert isinstance(out, np.ndarray) and out.dtype == np.bool_ and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
        return out

    with ThreadPoolExecutor(3) as pool:
        outs = list(pool.map(run, range(6)))
    assert all(np.array_equal(o, exp) for o in outs)
    monkeypatch.setattr(flagging, "_D2H_STAGE_MAX", 0)             # staging refused: direct copy
    assert np.array_equal(run(0), exp)
    monkeypatch.setattr(flagging, "_LINK_TURNS", False)            # the earlier free-for-all path
    with ThreadPoolExecutor(3) as pool:
        outs = list(pool.map(run, range(3)))
    assert all(np.array_equal(o, exp) for o in outs)


@pytest.mark.gpu
def test_numpy_block_pipelined_inside_the_call(gpu, oracle):
    """A numpy block of >= 32 windows is cut into pieces along the baseline axis and pipelined inside the call (copy of
    piece i + 1 under the kernels of piece i, two side streams): same flags as the oracle and as the one-piece path,
    ragged piece sizes, repeated calls, and calls from several threads at once."""
    import threading
    from tricolour_amd import flagging
    rs = np.random.RandomState(12)
    shape = (11, 4, 32, 64)                      # 44 windows: two pieces of 5 / 6 baselines
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 30] *= 7
    vis[4, 2, 9] *= 5
    vis[7, 1, 3, 5] = np.nan
    flags = rs.uniform(size=shape) < 0.03
    kw = dict(num_major_iterations=2, background_iterations=2)
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    assert flagging._PIPELINE_PIECES > 1
    v0, f0 = vis.copy(), flags.copy()
    for _ in range(2):
        got = gpu.sum_threshold_flagger(vis, flags, **kw)
        assert got.dtype == np.bool_ and got.shape == shape and np.array_equal(got, exp)
    assert np.array_equal(vis.view(np.uint64), v0.view(np.uint64)) and np.array_equal(flags, f0)   # inputs untouched
    old = flagging._PIPELINE_PIECES
    try:
        flagging._PIPELINE_PIECES = 1
        one = gpu.sum_threshold_flagger(vis, flags, **kw)
    finally:
        flagging._PIPELINE_PIECES = old
    assert np.array_equal(one, exp)
    results, errors = [None] * 3, []

    def work(i):
        try:
            results[i] = gpu.sum_threshold_flagger(vis, flags, **kw)
        except Exception as e:   # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors and all(np.array_equal(r, exp) for r in results)
