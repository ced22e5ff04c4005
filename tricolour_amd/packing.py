"""Window packing on the device: MS row order (row, chan, corr) <-> windows
(bl, corr, time, chan).  Mirrors ``tricolour/packing.py`` for the in-HBM
("numpy") backend; the zarr-disk spill store is out of scope (windows live in
288 GB of HBM instead).

The reference matches every row against every baseline
(``_numba_pack_data``, packing.py:262-276, O(nbl * rows)); here the
``row -> (baseline, time)`` map is computed once on the host
(:func:`row_map`) and the scatter / gather run as HIP kernels.
"""
import numpy as np

from tricolour_amd import _lib

_WINDOW_SCHEMA = ("bl", "corr", "time", "chan")    # packing.py:15


def unique_baselines(ant1, ant2):
    """(nbl, 3) int32 rows ``(bl_index, ant1, ant2)`` in the reference's
    order: ascending 64-bit value of the (ant1, ant2) int32 pair viewed as one
    little-endian int64, i.e. sorted by (ant2, ant1) (packing.py:36-56,
    apps/tricolour/app.py:444-450)."""
    ant1 = np.ascontiguousarray(ant1)
    ant2 = np.ascontiguousarray(ant2)
    if not (ant1.dtype == np.int32 and ant2.dtype == np.int32):
        raise TypeError("antenna1 '%s' and antenna2 '%s' dtypes "
                        "must both be np.int32" % (ant1.dtype, ant2.dtype))
    bl = np.stack([ant1, ant2], axis=1).copy().view(np.int64).reshape(-1)
    u = np.unique(bl)
    pairs = u.view(np.int32).reshape(-1, 2)
    idx = np.arange(pairs.shape[0], dtype=np.int32)[:, None]
    return np.concatenate([idx, pairs], axis=1).astype(np.int32)


def row_map(ant1, ant2, ubl, time_inv, ntime=None):
    """Per-row window coordinates: ``row_bl[r]`` = position in ``ubl`` of the
    row's baseline (-1 if absent), ``row_time[r] = time_inv[r]``.  When
    several rows map to the same (baseline, time) cell the reference's serial
    loop lets the LAST row win (packing.py:262-276); earlier duplicates are
    masked out here so that the parallel scatter is deterministic and equal."""
    ant1 = np.asarray(ant1, np.int64)
    ant2 = np.asarray(ant2, np.int64)
    ubl = np.asarray(ubl)
    key = ubl[:, 1].astype(np.int64) | (ubl[:, 2].astype(np.int64) << 32)
    order = np.argsort(key, kind="stable")
    rkey = ant1 | (ant2 << 32)
    pos = np.searchsorted(key[order], rkey)
    pos_c = np.clip(pos, 0, max(len(key) - 1, 0))
    found = (len(key) > 0) & (pos < len(key))
    if len(key):
        found = found & (key[order][pos_c] == rkey)
    row_bl = np.where(found, order[pos_c] if len(key) else 0, -1).astype(np.int32)
    row_time = np.asarray(time_inv, np.int32).copy()
    if ntime is None:
        ntime = int(row_time.max()) + 1 if row_time.size else 0
    cell = row_bl.astype(np.int64) * max(int(ntime), 1) + row_time
    valid = row_bl >= 0
    # keep only the last row of each occupied cell
    rev = np.arange(len(cell))[::-1]
    _, first_in_rev = np.unique(cell[rev], return_index=True)
    keep = np.zeros(len(cell), bool)
    keep[rev[first_in_rev]] = True
    row_bl_pack = np.where(valid & keep, row_bl, -1).astype(np.int32)
    return row_bl, row_bl_pack, row_time


def _torch_gpu():
    import torch
    _lib.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("tricolour_amd.packing needs a ROCm GPU; there is no CPU fallback")
    return torch


def _dev(torch, a, dtype=None):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    a = a.cuda() if not a.is_cuda else a
    if dtype is not None and a.dtype != dtype:
        a = a.to(dtype)
    return a.contiguous()


def pack_data(time_inv, ubl, antenna1, antenna2, data, flags, ntime):
    """Device version of ``packing.pack_data`` (packing.py:306-366) for one
    dataset: returns ``(vis_windows, flag_windows)`` torch tensors of shape
    (bl, corr, time, chan); cells no row maps to hold NaN+NaNj / True
    (packing.py:97,117)."""
    torch = _torch_gpu()
    lib = _lib.lib()
    ubl = np.asarray(ubl)
    rows, nchan, ncorr = (int(s) for s in data.shape)
    if tuple(flags.shape) != tuple(data.shape):
        raise ValueError("vis_windows.shape != flag_windows.shape")   # packing.py:253
    nbl = int(ubl.shape[0])
    _, row_bl, row_time = row_map(np.asarray(antenna1), np.asarray(antenna2), ubl,
                                  np.asarray(time_inv), ntime)
    d = _dev(torch, data, torch.complex64)
    f = _dev(torch, flags)
    f8 = f.view(torch.uint8) if f.dtype == torch.bool else (f != 0).view(torch.uint8)
    rb, rt = _dev(torch, row_bl), _dev(torch, row_time)
    vis_w = torch.empty((nbl, ncorr, int(ntime), nchan), dtype=torch.complex64, device=d.device)
    flag_w = torch.empty((nbl, ncorr, int(ntime), nchan), dtype=torch.uint8, device=d.device)
    stream = torch.cuda.current_stream(d.device).cuda_stream
    _lib.check(lib.tri_fill_windows(vis_w.data_ptr(), flag_w.data_ptr(), vis_w.numel(), stream))
    _lib.check(lib.tri_pack_data(d.data_ptr(), f8.data_ptr(), rb.data_ptr(), rt.data_ptr(),
                                 rows, nchan, ncorr, nbl, int(ntime), vis_w.data_ptr(),
                                 flag_w.data_ptr(), stream))
    return vis_w, flag_w.view(torch.bool)


def unpack_data(antenna1, antenna2, time_inv, ubl, flag_windows, equalize_corr=False):
    """Device version of ``packing.unpack_data`` (packing.py:391-425): gathers
    flag windows back to (row, chan, corr); rows whose baseline is not in
    ``ubl`` stay 0.  ``equalize_corr=True`` additionally flags every
    correlation of a visibility if any is flagged (the step the application
    applies right after unpacking, apps/tricolour/app.py:479-480)."""
    torch = _torch_gpu()
    lib = _lib.lib()
    ubl = np.asarray(ubl)
    nbl, ncorr, ntime, nchan = (int(s) for s in flag_windows.shape)
    if nbl != int(ubl.shape[0]):
        raise ValueError("flag_windows and ubl disagree on the number of baselines")
    row_bl, _, row_time = row_map(np.asarray(antenna1), np.asarray(antenna2), ubl,
                                  np.asarray(time_inv), ntime)
    rows = len(row_bl)
    fw = _dev(torch, flag_windows)
    fw8 = fw.view(torch.uint8) if fw.dtype == torch.bool else (fw != 0).view(torch.uint8)
    rb, rt = _dev(torch, row_bl), _dev(torch, row_time)
    out = torch.empty((rows, nchan, ncorr), dtype=torch.uint8, device=fw.device)
    stream = torch.cuda.current_stream(fw.device).cuda_stream
    _lib.check(lib.tri_unpack_data(fw8.data_ptr(), rb.data_ptr(), rt.data_ptr(), rows, nchan,
                                   ncorr, nbl, ntime, out.data_ptr(), 1 if equalize_corr else 0, stream))
    return out.view(torch.bool)
