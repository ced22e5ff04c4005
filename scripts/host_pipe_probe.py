"""Single-thread numpy-in / numpy-out calls with the in-call pipeline at 1, 2, 4 pieces: wall time per block and the
phase trace of one call (TRICOLOUR_AMD_TRACE=1)."""
import os, sys, time
os.environ["TRICOLOUR_AMD_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tricolour_amd
from tricolour_amd import flagging
T, F, ncorr, nbl = 1024, 4096, 4, 16
rs = np.random.RandomState(0)
shape = (nbl, ncorr, T, F)
vis = np.empty(shape, np.complex64)
vis.real = rs.standard_normal(shape)
vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
for pieces in (1, 2):
    flagging._PIPELINE_PIECES = pieces
    for _ in range(2):
        tricolour_amd.sum_threshold_flagger(vis, flags)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(4):
        tricolour_amd.sum_threshold_flagger(vis, flags)
    dt = (time.time() - t0) / 4
    print("pieces %d: %.1f ms per block -> %.0f Mvis/s" % (pieces, dt * 1e3, vis.size / dt / 1e6), flush=True)
    del flagging._TRACE[:]
    t0 = time.time()
    tricolour_amd.sum_threshold_flagger(vis, flags)
    t1 = time.time()
    for ev in flagging._TRACE:
        print("   ", ev[1], " ".join("%.1f" % ((x - t0) * 1e3) for x in ev[2:]))
    print("    call end %.1f" % ((t1 - t0) * 1e3))
