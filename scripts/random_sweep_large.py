"""Production-size windows with random kwargs against the oracle, seeds outside the test suite
(the body of tests/test_gpu_parity.py::test_large_windows_random_kwargs_vs_oracle):  python scripts/random_sweep_large.py [seed] [cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tricolour_amd as gpu
from oracle import oracle
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rs = np.random.RandomState(seed)
bad = 0
for case in range(ncases):
    t, f = [(1024, 4096), (512, 2048), (1024, 1024), (256, 8192), (768, 3072), (128, 16384)][case % 6]
    shape = (2, 1, t, f)
    kw = dict(background_iterations=int(rs.randint(1, 6)), spike_width_time=float(rs.uniform(2, 16)),
              spike_width_freq=float(rs.uniform(2, 14)), num_major_iterations=int(rs.randint(1, 3)),
              background_reject=float(rs.choice([1.5, 2.0, 3.0])), freq_chunks=int(rs.choice([1, 3, 4, 10, 16])),
              outlier_nsigma=float(rs.choice([4.5, 6.0, 10.0])))
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., rs.randint(0, f, 8)] *= 6
    vis[:, :, rs.randint(0, t, 4), :] += 4
    if case % 3 == 0:                                  # RFI blocks: bimodal residuals, wide flagged bands
        c0 = rs.randint(0, f - f // 8); vis[..., c0:c0 + f // 16] *= 30
        t0 = rs.randint(0, t - t // 8); vis[:, :, t0:t0 + t // 10, :] *= 12
    vis[rs.uniform(size=shape) < 1e-4] = np.nan
    flags = rs.uniform(size=shape) < rs.choice([0.0, 0.03, 0.3])
    if case % 4 == 1:
        flags[..., rs.randint(0, f - 200):][..., :150] = True
    t0 = time.time()
    exp = oracle.sum_threshold_flagger(vis, flags, **kw)
    out = gpu.sum_threshold_flagger(vis, flags, **kw)
    d = int((out != exp).sum())
    bad += d != 0
    print("case %2d shape %s %s: %d flags differ (flagged %.3f, %.1f s)" % (case, shape, kw, d, exp.mean(), time.time() - t0), flush=True)
print("cases", ncases, "bad", bad)
