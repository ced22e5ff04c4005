#!/usr/bin/env python3
"""Generates G13 (Stokes intensities) with the reference's own stokes.py, run
un-jitted through the identity numba stand-in (refshim.py).  Un-jitted NumPy
keeps complex64 arithmetic where numba computes in complex128, so this fixture
pins the device kernel to 1e-6 relative, the numba-typed oracle pins it exactly.
Build-container only; just the .npz data travels.

    cd tests/golden && PYTHONDONTWRITEBYTECODE=1 python3 make_golden_stokes.py
"""
import json

import numpy as np

from refshim import _install_numba_shim

_install_numba_shim()
import sys, types  # noqa: E402
sys.path.insert(0, "/root/reference")
pkg = types.ModuleType("tricolour")
pkg.__path__ = ["/root/reference/tricolour"]
sys.modules.setdefault("tricolour", pkg)
import tricolour.stokes as st  # noqa: E402

rs = np.random.RandomState(13)
vis = (rs.standard_normal((40, 33, 4)) + 1j * rs.standard_normal((40, 33, 4))).astype(np.complex64)
vis[3, 5] = 0
vis[7, :, 1] *= 1e4
orders = [["YX", "XX", "XY", "YY"], ["XX", "XY", "YX", "YY"], ["RR", "RL", "LR", "LL"], ["RL", "RR", "LL", "LR"]]
doc, arrays = [], {"vis": vis}
for k, names in enumerate(orders):
    codes = [st.STOKES_TYPES[n] for n in names]
    cmap = st.stokes_corr_map(codes)
    pol = tuple(v for s, v in cmap.items() if s != "I")
    unpol = tuple(v for s, v in cmap.items() if s == "I")
    allterms = tuple(cmap.values())
    arrays["pol_%d" % k] = st.polarised_intensity(vis, pol)
    arrays["total_%d" % k] = st.polarised_intensity(vis, allterms)
    arrays["unpol_%d" % k] = st.unpolarised_intensity(vis, unpol, pol)
    doc.append({"names": names, "codes": codes,
                "map": {s: [v[0], v[1], [v[2].real, v[2].imag], v[3], v[4]] for s, v in cmap.items()}})
arrays["cases"] = np.array(json.dumps(doc))
np.savez_compressed("G13_stokes.npz", **arrays)
print("G13 written", [d["names"] for d in doc])
