"""Throughput of the device pack / unpack (tri_pack_data / tri_unpack_data): a complete scan of `--bl` baselines x
`--time` times in MS row order (row, chan, corr) <-> windows (bl, corr, time, chan).  Algorithmic bytes:
pack 18 B/vis (8 + 1 read, 8 + 1 written), unpack 2 B/vis."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tricolour_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--bl", type=int, default=252); ap.add_argument("--time", type=int, default=1024)
ap.add_argument("--chan", type=int, default=4096); ap.add_argument("--corr", type=int, default=4)
a = ap.parse_args()
lib = _lib.lib()
nbl, T, F, nc = a.bl, a.time, a.chan, a.corr
rows = nbl * T
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
data = torch.view_as_complex(torch.randn((rows, F, nc, 2), generator=g, device=dev))
flag = (torch.rand((rows, F, nc), generator=g, device=dev) < 0.1).view(torch.uint8)
# MS order: time-major rows, baselines within a time
row_time = torch.arange(rows, device=dev, dtype=torch.int32) // nbl
row_bl = torch.arange(rows, device=dev, dtype=torch.int32) % nbl
vw = torch.empty((nbl, nc, T, F), dtype=torch.complex64, device=dev)
fw = torch.empty((nbl, nc, T, F), dtype=torch.uint8, device=dev)
out = torch.empty((rows, F, nc), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
nvis = rows * F * nc
def tm(f):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
tp = tm(lambda: _lib.check(lib.tri_pack_data(data.data_ptr(), flag.data_ptr(), row_bl.data_ptr(), row_time.data_ptr(), rows, F, nc, nbl, T, vw.data_ptr(), fw.data_ptr(), st)))
# round trip: windows back to rows must reproduce the flags
tu = tm(lambda: _lib.check(lib.tri_unpack_data(fw.data_ptr(), row_bl.data_ptr(), row_time.data_ptr(), rows, F, nc, nbl, T, out.data_ptr(), 0, st)))
assert torch.equal(out, flag), "pack -> unpack round trip differs"
assert torch.equal(vw[5, 1, 7], data[7 * nbl + 5, :, 1])
tue = tm(lambda: _lib.check(lib.tri_unpack_data(fw.data_ptr(), row_bl.data_ptr(), row_time.data_ptr(), rows, F, nc, nbl, T, out.data_ptr(), 1, st)))
print("pack   %d bl x %d corr x %d x %d: %.2f ms  %.0f Mvis/s  %.2f TB/s at 18 B/vis" % (nbl, nc, T, F, tp * 1e3, nvis / tp / 1e6, nvis * 18 / tp / 1e12))
print("unpack (flags): %.2f ms  %.0f Mvis/s  %.2f TB/s at 2 B/vis" % (tu * 1e3, nvis / tu / 1e6, nvis * 2 / tu / 1e12))
print("unpack + any-correlation equalisation: %.2f ms  %.0f Mvis/s" % (tue * 1e3, nvis / tue / 1e6))
