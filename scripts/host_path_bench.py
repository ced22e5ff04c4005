"""PCIe-inclusive rate: numpy in -> tricolour_amd.sum_threshold_flagger -> numpy out
(H2D of complex64 + flags, kernels, D2H of flags), library-default kwargs, one block per call as the
dask graph does, from 1..N concurrent threads (dask's ThreadPool).  Never bench.py's `value`."""
import argparse, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tricolour_amd
ap = argparse.ArgumentParser(); ap.add_argument("--bl", type=int, default=16); ap.add_argument("--blocks", type=int, default=8)
a = ap.parse_args()
T, F, ncorr = 1024, 4096, 4
rs = np.random.RandomState(0)
shape = (a.bl, ncorr, T, F)
vis = np.empty(shape, np.complex64)
vis.real = rs.standard_normal(shape)
vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
ref = tricolour_amd.sum_threshold_flagger(vis, flags)
flagging_threads = None
for threads in (1, 2, 4, 8):
    from tricolour_amd import flagging
    flagging.set_num_threads(threads)
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda i: tricolour_amd.sum_threshold_flagger(vis, flags), range(threads)))   # warm workspaces
        t0 = time.time()
        outs = list(pool.map(lambda i: tricolour_amd.sum_threshold_flagger(vis, flags), range(a.blocks)))
        dt = time.time() - t0
    assert all(np.array_equal(o, ref) for o in outs)
    print("host blocks of %d bl x 4 x 1024 x 4096, %d threads: %d blocks in %.3f s -> %.0f Mvis/s (10 B/vis over PCIe)" %
          (a.bl, threads, a.blocks, dt, a.blocks * vis.size / dt / 1e6), flush=True)
