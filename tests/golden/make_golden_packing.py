#!/opt/conda/bin/python3.9
"""Generates G8 (pack / unpack) and G9 (config-1 dask plumbing capture) with
the reference's own packing.py / dask_wrappers.py, run un-jitted under
/opt/conda/bin/python3.9 (the only interpreter here with dask; SURVEY.md 8c
"oracle B").  Build-container only; just the .npz data travels.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 make_golden_packing.py
"""
import sys
import time
import types

import numpy as np

if not hasattr(np, "exceptions"):
    np.exceptions = types.SimpleNamespace(RankWarning=np.RankWarning)
elif not hasattr(np.exceptions, "RankWarning"):
    np.exceptions.RankWarning = np.RankWarning

# stub zarr (only the zarr-disk backend, out of scope, would use it)
zarr = types.ModuleType("zarr")
zarr.Array = type("Array", (), {})
zarr.ThreadSynchronizer = lambda *a, **k: None
zarr.creation = types.SimpleNamespace(create=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no zarr")))
sys.modules["zarr"] = zarr

from refshim import load_reference_flagging  # noqa: E402

fl = load_reference_flagging()
import dask  # noqa: E402
import dask.array as da  # noqa: E402
import tricolour.packing as packing  # noqa: E402
import tricolour.dask_wrappers as dw  # noqa: E402


def make_rows(rs, na, ntime, nchan, ncorr, delete_frac, real_only):
    a1, a2 = np.triu_indices(na, 0)
    nbl = len(a1)
    times = np.sort(rs.choice(np.arange(1000, 1000 + 3 * ntime), ntime, replace=False)).astype(np.float64)
    ant1 = np.tile(a1, ntime).astype(np.int32)
    ant2 = np.tile(a2, ntime).astype(np.int32)
    time = np.repeat(times, nbl)
    keep = rs.uniform(size=ant1.size) >= delete_frac
    perm = rs.permutation(np.nonzero(keep)[0]) if delete_frac > 0 else np.nonzero(keep)[0]
    ant1, ant2, time = ant1[perm], ant2[perm], time[perm]
    rows = ant1.size
    shape = (rows, nchan, ncorr)
    if real_only:
        data = (np.abs(rs.standard_normal(shape)) + 4).astype(np.complex64)
    else:
        data = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    flag = rs.uniform(size=shape) < 0.1
    return ant1, ant2, time, data, flag


def pack_unpack(name, rs, na, ntime, nchan, ncorr, delete_frac, row_chunks, bl_chunks, flagger_kw=None):
    t0 = time.time()
    ant1, ant2, tm, data, flag = make_rows(rs, na, ntime, nchan, ncorr, delete_frac, flagger_kw is not None)
    if flagger_kw is not None:
        data[:, nchan // 3, :] += 9.0
    d_ant1 = da.from_array(ant1, chunks=row_chunks)
    d_ant2 = da.from_array(ant2, chunks=row_chunks)
    d_data = da.from_array(data, chunks=(row_chunks, nchan, ncorr))
    d_flag = da.from_array(flag, chunks=(row_chunks, nchan, ncorr))
    ubl = dask.compute(packing.unique_baselines(d_ant1, d_ant2))[0]
    ubl = ubl.view(np.int32).reshape(-1, 2)
    utime, time_inv = np.unique(tm, return_inverse=True)
    time_inv = time_inv.astype(np.int32)
    ubl3 = np.concatenate([np.arange(ubl.shape[0], dtype=np.int32)[:, None], ubl], axis=1)
    d_ubl = da.from_array(ubl3, chunks=(bl_chunks, 3))
    d_tinv = da.from_array(time_inv, chunks=row_chunks)
    vis_w, flag_w = packing.pack_data(d_tinv, d_ubl, d_ant1, d_ant2, d_data, d_flag, utime.size, backend="numpy")
    out = {}
    if flagger_kw is None:
        new_flag_w = flag_w
    else:
        kw = {k: (np.float64(v) if isinstance(v, float) else v) for k, v in flagger_kw.items()}
        new_flag_w = dw.sum_threshold_flagger(vis_w, flag_w, **kw)
    unpacked = packing.unpack_data(d_ant1, d_ant2, d_tinv, d_ubl, new_flag_w)
    vw, fw, nfw, up = dask.compute(vis_w, flag_w, new_flag_w, unpacked, scheduler="single-threaded")
    out.update(ant1=ant1, ant2=ant2, time_inv=time_inv, ubl=ubl3, data=data, flag=flag,
               vis_windows=vw, flag_windows=fw, out_windows=nfw, unpacked=up,
               ntime=np.int64(utime.size))
    if flagger_kw is not None:
        out.update({"kw_" + k: np.asarray(v) for k, v in flagger_kw.items()})
    np.savez_compressed(name + ".npz", **out)
    print("%s: rows %d, windows %s, flagged %d -> %d (%.0f s)" % (name, ant1.size, vw.shape, fw.sum(), nfw.sum(), time.time() - t0))
    sys.stdout.flush()


if __name__ == "__main__":
    names = sys.argv[1:]
    if not names or "G8" in names:
        pack_unpack("G8_packing", np.random.RandomState(8), na=7, ntime=10, nchan=16, ncorr=4,
                    delete_frac=0.1, row_chunks=57, bl_chunks=5)
    if not names or "G9" in names:
        # config 1: 4 bl x 1 corr x 256 time x 256 chan, purely-real complex
        # visibilities (|z| exact under any hypot), one major iteration
        pack_unpack("G9_config1_plumbing", np.random.RandomState(9), na=3, ntime=256, nchan=256, ncorr=1,
                    delete_frac=0.0, row_chunks=400, bl_chunks=2,
                    flagger_kw=dict(num_major_iterations=1, outlier_nsigma=4.5, background_reject=2.0,
                                    spike_width_time=12.5, spike_width_freq=10.0, flag_all_time_frac=0.6,
                                    flag_all_freq_frac=0.8, rho=1.3))
