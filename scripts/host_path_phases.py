"""Where a host-block call spends its time (one thread): H2D, kernels, D2H, result allocation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tricolour_amd
from tricolour_amd import flagging
T, F, ncorr, bl = 1024, 4096, 4, 16
rs = np.random.RandomState(0)
shape = (bl, ncorr, T, F)
vis = np.empty(shape, np.complex64); vis.real = rs.standard_normal(shape); vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
def tm(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms, out = tm(lambda: tricolour_amd.sum_threshold_flagger(vis, flags))
print("whole host call: %.1f ms (%.0f Mvis/s)" % (ms, vis.size / ms / 1e3))
ms, vd = tm(lambda: torch.from_numpy(vis).cuda())
print("H2D vis, driver path: %.1f ms (%.1f GB/s)" % (ms, vis.nbytes / ms / 1e6))
st = flagging._stager(torch)
vd2 = torch.empty(shape, dtype=torch.complex64, device="cuda")
ms, _ = tm(lambda: st.h2d(torch, vis, vd2))
print("H2D vis, pinned stager: %.1f ms (%.1f GB/s)" % (ms, vis.nbytes / ms / 1e6))
fd = torch.from_numpy(flags).cuda()
ms, od = tm(lambda: tricolour_amd.sum_threshold_flagger(vd, fd))
print("kernels (device tensors): %.1f ms" % ms)
ms, _ = tm(lambda: od.cpu().numpy())
print("D2H flags + fresh numpy result: %.1f ms" % ms)
ms, _ = tm(lambda: np.empty(shape, np.bool_).fill(1))
print("allocating + touching a fresh result array: %.1f ms" % ms)
res = np.empty(shape, np.bool_)
ms, _ = tm(lambda: st.d2h(torch, od.view(torch.uint8), res))
print("D2H via stager into a touched array: %.1f ms" % ms)
