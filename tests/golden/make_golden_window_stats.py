#!/opt/conda/bin/python3.9
"""Generates G12 (flag summary statistics) with the reference's own
window_statistics.py under /opt/conda/bin/python3.9 (the only interpreter here
with dask).  Build-container only; just the .npz data travels.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 make_golden_window_stats.py
"""
import json
import sys
import types

import numpy as np

if not hasattr(np, "exceptions"):
    np.exceptions = types.SimpleNamespace(RankWarning=np.RankWarning)
elif not hasattr(np.exceptions, "RankWarning"):
    np.exceptions.RankWarning = np.RankWarning
zarr = types.ModuleType("zarr")
zarr.Array = type("Array", (), {})
zarr.ThreadSynchronizer = lambda *a, **k: None
sys.modules["zarr"] = zarr

from refshim import load_reference_flagging  # noqa: E402

load_reference_flagging()
import tricolour.window_statistics as ws  # noqa: E402

rs = np.random.RandomState(12)
names = ["m000", "m001", "m002", "m003", "m004"]
a1, a2 = np.triu_indices(len(names), 0)
ubl = np.stack([np.arange(a1.size), a1, a2], axis=1)
nbl, ncorr, ntime, nchan, nbins = ubl.shape[0], 2, 7, 37, 6
flags = rs.uniform(size=(nbl, ncorr, ntime, nchan)) < 0.3
flags[:, :, :, 5] = True
flags[3] = True
freqs = np.linspace(0.856e9, 1.712e9, nchan)

total = ws.WindowStatistics(nbins)
chunks = [(0, 4), (4, 9), (9, nbl)]
for scan, field, ddid in ((1, "PKS1934", 0), (2, "J0408", 0), (2, "J0408", 1)):
    for lo, hi in chunks:
        # the nesting dask hands to the block function (window_statistics.py:21-23)
        st = ws._window_stats([[[flags[lo:hi]]]], [ubl[lo:hi]], [freqs], names, scan, field, ddid, nbins)
        total.update(st)
summary = ws.summarise_stats(total, total)


def plain(d):
    return {str(k): (v.tolist() if hasattr(v, "tolist") else int(v)) for k, v in d.items()}


doc = {k: plain(getattr(total, "_" + k)) for k in (
    "counts_per_ant", "size_per_ant", "counts_per_bl", "size_per_bl", "counts_per_field", "size_per_field",
    "counts_per_scan", "size_per_scan", "counts_per_ddid", "bins_per_ddid", "size_per_ddid")}
np.savez_compressed("G12_window_stats.npz", flags=flags, ubl=ubl, freqs=freqs, names=np.array(names),
                    nbins=nbins, chunks=np.array(chunks),
                    calls=np.array(json.dumps([[1, "PKS1934", 0], [2, "J0408", 0], [2, "J0408", 1]])),
                    expected=np.array(json.dumps(doc)), summary=np.array("\n".join(summary)))
print("G12 written", {k: len(v) for k, v in doc.items()})
