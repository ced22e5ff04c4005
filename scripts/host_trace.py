"""TRICOLOUR_AMD_TRACE=1 python scripts/host_trace.py <threads> <blocks>: phase timeline of numpy-in / numpy-out calls."""
import os, sys, time
os.environ["TRICOLOUR_AMD_TRACE"] = "1"
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tricolour_amd
from tricolour_amd import flagging
N = int(sys.argv[1]); B = int(sys.argv[2])
T, F, ncorr, bl = 1024, 4096, 4, 16
rs = np.random.RandomState(0)
shape = (bl, ncorr, T, F)
vis = np.empty(shape, np.complex64); vis.real = rs.standard_normal(shape); vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
flagging.set_num_threads(N)
with ThreadPoolExecutor(N) as pool:
    list(pool.map(lambda i: tricolour_amd.sum_threshold_flagger(vis, flags), range(N)))
    flagging._TRACE.clear()
    t0 = time.time()
    list(pool.map(lambda i: tricolour_amd.sum_threshold_flagger(vis, flags), range(B)))
    dt = time.time() - t0
tids = sorted({e[0] for e in flagging._TRACE})
for e in sorted(flagging._TRACE, key=lambda e: e[2]):
    print("thread %d %-12s " % (tids.index(e[0]), e[1]) + " ".join("%.0f" % ((x - t0) * 1e3) for x in e[2:]))
print("total %.0f ms, %d blocks -> %.0f Mvis/s" % (dt * 1e3, B, B * vis.size / dt / 1e6))
