"""Times one rejection step (tri_bench_reject) on a slab whose flags make blocks fall out of the one-pass route:
ordinary 5 % flags, 10 % of the windows fully flagged (flag_autos), a quarter of the channels flagged (static mask), 97 % flagged."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tricolour_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
W, F, T, G = 504, 4096, 1024, 10
g = torch.Generator(device=dev); g.manual_seed(3)
resid = torch.randn((W, F, T), generator=g, device=dev).abs_()
ends = [int(x) for x in np.linspace(0, F, G + 1)]
e = (C.c_int64 * len(ends))(*ends)
fo = torch.empty((W, F, T), dtype=torch.uint8, device=dev)
t4 = torch.empty((W, T // 4, F, 4), dtype=torch.uint8, device=dev)
med = torch.empty((W, G), dtype=torch.float64, device=dev)
ms = C.c_float(0)
for name in ("5 % random flags", "10 % of the windows fully flagged", "a quarter of the channels flagged", "97 % random flags"):
    fl = torch.rand((W, F, T), generator=g, device=dev) < (0.97 if name.startswith("97") else 0.05)
    if name.startswith("10 %"):
        fl[::10] = True
    if name.startswith("a quarter"):
        fl[:, ::4, :] = True
    f = fl.view(torch.uint8)
    stats0 = (C.c_uint64 * 20)()
    lib.tri_medrej_stats(stats0, 1)
    for reps in (1, 3):
        _lib.check(lib.tri_bench_reject(resid.data_ptr(), f.data_ptr(), fo.data_ptr(), t4.data_ptr(), med.data_ptr(), W, F, T, e, len(ends), 2.0, reps, C.byref(ms), None))
    st = (C.c_uint64 * 20)()
    lib.tri_medrej_stats(st, 0)
    print("%-36s %7.2f ms per step of %d windows (%d blocks); second rounds %d, blocks redone by one workgroup %d (4 launches)" % (
        name, ms.value, W, W * G, st[3], sum(st[4:20])), flush=True)
    del fl, f
