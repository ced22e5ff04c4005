"""Flag summary statistics: drop-in for ``tricolour.window_statistics``
(reference ``tricolour/window_statistics.py``), with the counting done on the
GPU.

Every number of the reference's per-block ``_window_stats`` (:12-66) is a sum
of per-baseline or per-channel flag counts, so a block costs ONE pass over the
flag window (``tri_window_counts``: ``per_bl`` and ``per_chan`` vectors) and a
little host arithmetic on ``nbl + nchan`` integers, instead of one masked
reduction of the whole window per antenna, per baseline and per frequency bin.

Same public names and call signatures as the reference: ``WindowStatistics``,
``window_stats`` (dask graph, one task per baseline chunk), ``combine_window_stats``
and ``summarise_stats``.  dask is imported lazily.
"""
import ctypes as C
from collections import defaultdict

import numpy as np

from tricolour_amd import _lib
from tricolour_amd.packing import _WINDOW_SCHEMA

_KEYED = ("ant", "bl", "scan", "field")     # categories keyed by a name / number


def window_counts(flag_window):
    """``(per_bl, per_chan)`` uint64 numpy vectors of set-flag counts of a
    ``(bl, corr, time, chan)`` window (numpy array or ROCm torch tensor)."""
    import torch
    if isinstance(flag_window, torch.Tensor):
        fw = flag_window
        if not fw.is_cuda:
            fw = fw.cuda()
    else:
        fw = torch.from_numpy(np.ascontiguousarray(np.asarray(flag_window) != 0)).cuda()
    if fw.dim() != 4:
        raise ValueError("flag_window must have shape (bl, corr, time, chan)")
    if fw.dtype == torch.bool:
        fw = fw.view(torch.uint8)
    elif fw.dtype != torch.uint8:
        fw = (fw != 0).view(torch.uint8)
    fw = fw.contiguous()
    nbl, ncorr, ntime, nchan = (int(v) for v in fw.shape)
    per_bl = torch.empty(max(nbl, 1), dtype=torch.int64, device=fw.device)
    per_chan = torch.empty(max(nchan, 1), dtype=torch.int64, device=fw.device)
    stream = torch.cuda.current_stream(fw.device).cuda_stream
    _lib.check(_lib.lib().tri_window_counts(fw.data_ptr(), nbl, ncorr, ntime, nchan, per_bl.data_ptr(),
                                            per_chan.data_ptr(), C.c_void_p(stream)))
    return (per_bl[:nbl].cpu().numpy().astype(np.uint64), per_chan[:nchan].cpu().numpy().astype(np.uint64))


class WindowStatistics(object):
    """Running flag tallies (``window_statistics.py:180-244``): counts and sizes
    per antenna, baseline, scan and field, and per-ddid counts in frequency
    bins.  The attribute names are the reference's, ``summarise_stats`` and the
    reference's tests read them."""

    def __init__(self, nchanbins):
        self._nchanbins = nchanbins
        for what in _KEYED:
            setattr(self, "_counts_per_" + what, defaultdict(int))
            setattr(self, "_size_per_" + what, defaultdict(int))
        self._counts_per_ddid = defaultdict(lambda: np.zeros(nchanbins, dtype=np.uint64))
        self._size_per_ddid = defaultdict(int)
        self._bins_per_ddid = defaultdict(int)          # ddid -> frequency labels of the bins

    def update(self, other):
        """Adds the tallies of ``other`` (:202-236)."""
        for what in _KEYED + ("ddid",):
            mine = getattr(self, "_counts_per_" + what)
            for key, value in getattr(other, "_counts_per_" + what).items():
                mine[key] += value
            mine = getattr(self, "_size_per_" + what)
            for key, value in getattr(other, "_size_per_" + what).items():
                mine[key] += value
        for ddid, edges in other._bins_per_ddid.items():
            self._bins_per_ddid[ddid] = edges

    def copy(self):
        twin = WindowStatistics(self._nchanbins)
        twin.update(self)
        return twin


def stats_from_counts(per_bl, per_chan, samples_per_bl, ubls, chan_freqs, antenna_names,
                      scan_no, field_name, ddid, nchanbins):
    """The tallies of ``_window_stats`` (:12-66) from the two count vectors of
    one block.  ``samples_per_bl`` = corr * time * chan of the window."""
    ubls = np.asarray(ubls)
    chan_freqs = np.asarray(chan_freqs)
    per_bl = np.asarray(per_bl, dtype=np.uint64)
    per_chan = np.asarray(per_chan, dtype=np.uint64)
    out = WindowStatistics(nchanbins)
    # antennas: every baseline the antenna takes part in (:27-32)
    for index, name in enumerate(antenna_names):
        member = (ubls[:, 1] == index) | (ubls[:, 2] == index)
        out._counts_per_ant[name] += per_bl[member].sum(dtype=np.uint64)
        out._size_per_ant[name] += int(member.sum()) * samples_per_bl
    # baselines, labelled "<ant1>&<ant2>" (:35-43)
    for number in np.unique(ubls[:, 0]):
        member = ubls[:, 0] == number
        first = np.nonzero(member)[0][0]
        label = "{0:s}&{1:s}".format(antenna_names[ubls[first, 1]], antenna_names[ubls[first, 2]])
        out._counts_per_bl[label] += per_bl[member].sum(dtype=np.uint64)
        out._size_per_bl[label] += int(member.sum()) * samples_per_bl
    # scan and field: the whole block (:46-52)
    everything = per_bl.sum(dtype=np.uint64)
    block_size = per_bl.size * samples_per_bl
    out._counts_per_field[field_name] += everything
    out._size_per_field[field_name] += block_size
    out._counts_per_scan[scan_no] += everything
    out._size_per_scan[scan_no] += block_size
    # frequency bins: nchanbins edges make nchanbins - 1 half-open bins, the last
    # entry stays empty and the highest channel falls outside, as in :55-62
    edges = np.linspace(np.min(chan_freqs), np.max(chan_freqs), nchanbins)
    binned = np.zeros(nchanbins, dtype=np.uint32)
    for k in range(nchanbins - 1):
        inside = (chan_freqs >= edges[k]) & (chan_freqs < edges[k + 1])
        binned[k] = per_chan[inside].sum(dtype=np.uint64)
    out._counts_per_ddid[ddid] += binned
    out._bins_per_ddid[ddid] = edges
    out._size_per_ddid[ddid] += block_size
    return out


def window_stats_block(flag_window, ubls, chan_freqs, antenna_names, scan_no, field_name, ddid, nchanbins=10):
    """``_window_stats`` for one in-memory block: GPU counts + host tallies."""
    shape = tuple(int(v) for v in flag_window.shape)
    per_bl, per_chan = window_counts(flag_window)
    return stats_from_counts(per_bl, per_chan, shape[1] * shape[2] * shape[3], ubls, chan_freqs,
                             antenna_names, scan_no, field_name, ddid, nchanbins)


def _fold(chunk_stats, start):
    if isinstance(start, np.ndarray):
        start = start.item()
    total = start.copy()
    for one in np.asarray(chunk_stats, dtype=object).ravel():
        total.update(one)
    return total


def window_stats(flag_window, ubls, chan_freqs, antenna_names, scan_no, field_name, ddid,
                 nchanbins=10, prev_stats=None):
    """Dask graph with the reference's signature and result (:80-139): a 0-d
    object dask array holding one :class:`WindowStatistics` for the window,
    ``prev_stats`` folded in.  One task per baseline chunk."""
    import dask.array as da

    def per_chunk(fw, ub, freqs):
        box = np.empty((1,), dtype=object)
        box[0] = window_stats_block(fw, ub, freqs, antenna_names, scan_no, field_name, ddid, nchanbins)
        return box

    parts = da.blockwise(per_chunk, ("bl",), flag_window, _WINDOW_SCHEMA, ubls, ("bl", "bl-comp"),
                         chan_freqs, ("chan",), concatenate=True,
                         adjust_chunks={"bl": 1}, meta=np.empty((0,), dtype=object))
    if prev_stats is None:
        prev_stats = da.blockwise(lambda: WindowStatistics(nchanbins), (), meta=np.empty((), dtype=object))
    return da.blockwise(_fold, (), parts, ("bl",), prev_stats, (), concatenate=True,
                        meta=np.empty((), dtype=object))


def _merge_all(*stats):
    total = stats[0].copy()
    for one in stats[1:]:
        total.update(one)
    return total


def combine_window_stats(window_stats):
    """One :class:`WindowStatistics` out of a list of 0-d dask arrays (:150-168)."""
    import dask.array as da
    operands = []
    for ws in window_stats:
        operands += [ws, ()]
    return da.blockwise(_merge_all, (), *operands, dtype=object)


def _percent_lines(title, final_counts, final_sizes, orig_counts, orig_sizes, key_format):
    lines = [title]
    for key in final_counts:
        lines.append(("\t " + key_format + ": {1:.3f}%, original {2:.3f}%").format(
            key, final_counts[key] * 100.0 / final_sizes[key], orig_counts[key] * 100.0 / orig_sizes[key]))
    return lines


def summarise_stats(final, original):
    """The flag summary of the reference's log, line for line (:247-315)."""
    bar = "********************************"
    lines = [bar, "   BEGINNING OF FLAG SUMMARY    ", bar]
    for title, what, key_format in (("Per antenna:", "ant", "{0:s}"), ("Per scan:", "scan", "{0:d}"),
                                    ("Per field:", "field", "{0:s}"), ("Per baseline:", "bl", "{0:s}")):
        lines += _percent_lines(title, getattr(final, "_counts_per_" + what), getattr(final, "_size_per_" + what),
                                getattr(original, "_counts_per_" + what), getattr(original, "_size_per_" + what),
                                key_format)
    lines.append("Per data descriptor id:")
    for ddid in final._counts_per_ddid:
        percent = final._counts_per_ddid[ddid] * 100.0 / final._size_per_ddid[ddid]
        lines.append("\t {0:d}: {1:s}%".format(ddid, "\t".join("{0:<7.2f}".format(v) for v in percent)))
        mhz = final._bins_per_ddid[ddid] / 1e6
        lines.append("\t    {0:s} MHz".format("\t".join("{0:<7.1f}".format(v) for v in mhz)))
    lines += [bar, "       END OF FLAG SUMMARY      ", bar]
    return lines
