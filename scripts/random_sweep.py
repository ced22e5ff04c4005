"""Extra randomized parity sweep against the oracle with seeds the test suite does not use (tests/test_gpu_parity.py::_random_case):
    python scripts/random_sweep.py [first block] [blocks]   (20 cases per block)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import tricolour_amd as gpu
from oracle import oracle
import test_gpu_parity as T
bad = 0; n = 0
b0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for block in range(b0, b0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 12)):
    rs = np.random.RandomState(91000 + block)
    for k in range(20):
        vis, flags, kw = T._random_case(rs)
        try:
            exp = oracle.sum_threshold_flagger(vis, flags, **kw)
        except ValueError:
            try:
                gpu.sum_threshold_flagger(vis, flags, **kw); bad += 1; print("no error raised", block, k, kw)
            except ValueError:
                pass
            continue
        out = gpu.sum_threshold_flagger(vis, flags, **kw)
        n += 1
        d = int((out != exp).sum())
        if d:
            bad += 1; print("MISMATCH block %d case %d shape %s kw %s: %d" % (block, k, vis.shape, kw, d), flush=True)
print("cases", n, "bad", bad)
