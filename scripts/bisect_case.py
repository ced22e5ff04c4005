"""Runs one case of tests/test_gpu_parity.py::test_large_windows_random_kwargs_vs_oracle (argv[1] = case) and prints the
number of differing flags; environment knobs select kernel routes (read once per process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tricolour_amd
from oracle import oracle
want = int(sys.argv[1])
rs = np.random.RandomState(777)
for case in range(8):
    t, f = [(1024, 4096), (512, 2048), (1024, 1024), (256, 8192)][case % 4]
    shape = (2, 1, t, f)
    kw = dict(background_iterations=int(rs.randint(1, 6)), spike_width_time=float(rs.uniform(3, 14)),
              spike_width_freq=float(rs.uniform(3, 12)), num_major_iterations=int(rs.randint(1, 3)),
              background_reject=float(rs.choice([2.0, 3.0])), freq_chunks=int(rs.choice([4, 10])))
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., rs.randint(0, f, 8)] *= 6
    vis[:, :, rs.randint(0, t, 4), :] += 4
    vis[rs.uniform(size=shape) < 1e-4] = np.nan
    flags = rs.uniform(size=shape) < 0.03
    if case != want:
        continue
    dbg = {}
    exp, inter = oracle.sum_threshold_flagger(vis[:1], flags[:1], dump=True, **kw)
    out = tricolour_amd.sum_threshold_flagger(vis[:1], flags[:1], _debug=dbg, **kw)
    msg = ["out %d" % int((out != exp).sum())]
    for k in ("spec_resid", "background", "residual"):
        a = np.ascontiguousarray(inter[k], np.float32); b = np.ascontiguousarray(dbg[k], np.float32).reshape(a.shape)
        bad = ~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))
        msg.append("%s %d" % (k, int(bad.sum())))
        if bad.any() and k == "background":
            idx = np.argwhere(bad)
            msg.append("bg rows %s cols %s" % (sorted(set(idx[:, 0]))[:8], sorted(set(idx[:, 1]))[:8]))
    for k in ("spec_flags", "time_flags", "freq_flags"):
        msg.append("%s %d" % (k, int((inter[k].astype(bool) != dbg[k].reshape(inter[k].shape)).sum())))
    print("case", case, kw, "|", " ".join(msg))
