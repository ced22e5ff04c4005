#!/usr/bin/env python3
"""G11: reference tricolour.flagging.uvcontsub_flagger (plain NumPy, run here
under NumPy 2) on a small block.  Build-container only; the .npz travels."""
import numpy as np

from refshim import load_reference_flagging

fl = load_reference_flagging()
rs = np.random.RandomState(11)
shape = (2, 2, 40, 128)
x = np.linspace(0, 1, shape[3])
bp = (3.0 + 1.5 * np.sin(2 * np.pi * 1.5 * x) + 0.5 * x)[None, None, None, :]
vis = (bp + 0.2 * rs.standard_normal(shape) + 1j * (0.5 * bp + 0.2 * rs.standard_normal(shape))).astype(np.complex64)
vis[..., 40] += 4.0
vis[0, 1, 7, :] += 2.5
vis[1, 0, 20:24, 90:100] += 3.0
vis[0, 0, 3, 5] = np.nan
vis[1, 1, :, 77] = np.nan
flags = rs.uniform(size=shape) < 0.03
flags[1, 1] = True            # a fully flagged correlation product
flags[0, 0, :, 60:63] = True
cases = {}
for name, kw in (("a", dict(major_cycles=7, or_original_from_cycle=1, taylor_degrees=20, sigma=15.0)),
                 ("b", dict(major_cycles=3, or_original_from_cycle=0, taylor_degrees=25, sigma=5.0))):
    out = fl.uvcontsub_flagger(vis.copy(), flags.copy(), **kw)
    cases["out_" + name] = out
    for k, v in kw.items():
        cases["kw_%s_%s" % (name, k)] = np.asarray(v)
    print(name, kw, out.sum(), out.size)
np.savez_compressed("G11_uvcontsub.npz", vis=vis, flags=flags, numpy_version=np.asarray(np.__version__), **cases)
