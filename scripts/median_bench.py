"""Times the block-median kernels (tri_test_median variants) on a slab-sized residual image:
(n_win, rows=1, row_len = T * F) with G chunk segments of T * chunk samples, as _median_abs sees them."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tricolour_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--win", type=int, default=252)
ap.add_argument("--time", type=int, default=1024)
ap.add_argument("--chan", type=int, default=4096)
ap.add_argument("--variants", default="3,5")
a = ap.parse_args()
lib = _lib.lib()
dev = torch.device("cuda", 0)
W, T, F = a.win, a.time, a.chan
g = torch.Generator(device=dev); g.manual_seed(5)
data = torch.randn((W, 1, T * F), generator=g, device=dev).abs_()
flags = (torch.rand((W, 1, T * F), generator=g, device=dev) < 0.1).view(torch.uint8)
ends = [int(x) * T for x in np.linspace(0, F, 11).astype(int)]
e = (C.c_int64 * len(ends))(*ends)
ref = None
for v in [int(x) for x in a.variants.split(",")]:
    med = torch.empty((W, 1, 10), dtype=torch.float64, device=dev)
    best = 1e9
    for rnd in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.tri_test_median(data.data_ptr(), flags.data_ptr(), med.data_ptr(), W, 1, T * F, e, len(ends), v, None))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    same = "" if ref is None else ("identical" if torch.equal(ref, med) else "DIFFERENT")
    ref = med if ref is None else ref
    print("variant %d: %.2f ms (%.2f TB/s @5B) %s" % (v, best * 1e3, W * T * F * 5 / best / 1e12, same), flush=True)
