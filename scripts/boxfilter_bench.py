"""Times the masked box filter's time-axis stage (tri_bench_boxfilter) per radius and route, and checks that
the register-ring kernels (variant 2) reproduce the LDS-ring kernels (variant 1) bit for bit.
usage: boxfilter_bench.py [--win 252] [--radii 8,10,17,...] [--rounds 3]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tricolour_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--win", type=int, default=252)
ap.add_argument("--time", type=int, default=1024)
ap.add_argument("--chan", type=int, default=4096)
ap.add_argument("--radii", default="8,10,17,21,32,43,54")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="1,2")
ap.add_argument("--stage", type=int, default=0)
ap.add_argument("--tiny-lines", type=float, default=0.0,
                help="stage 1: this fraction of the lines gets one weight of 1e-30 -- their float64 sums are inexact, the exact row "
                     "filter (variant 4) has to redo them in the reference's sequential order")
ap.add_argument("--time-radius", type=int, default=28,
                help="stage 1: its input images are the time-axis stage's outputs at this radius (as in the flagger)")
a = ap.parse_args()
lib = _lib.lib()
dev = torch.device("cuda", 0)
W, T, F = a.win, a.time, a.chan
g = torch.Generator(device=dev); g.manual_seed(3)
data = torch.randn((W, T, F), generator=g, device=dev).abs_()
flags = torch.rand((W, T, F), generator=g, device=dev) < 0.05
flags[:, :, ::50] = True
# TF4 packing: byte k of word (q, c) = flag of time 4 q + k
f4 = flags.view(torch.uint8).view(W, T // 4, 4, F).permute(0, 1, 3, 2).contiguous()
# stage 1: per window the weight image followed by the weight * data image
both = torch.empty((W, 2, T, F), device=dev)
both[:, 1] = data
both[:, 0] = (~flags).float() * 0.9 + 0.05 * torch.rand((W, T, F), generator=g, device=dev)
both[:, 0, :, 1000:1200] = 0.0                                                       # fully flagged band -> NaN background
wimg = both
if a.stage == 1 and a.time_radius > 0:
    # realistic stage-1 inputs: what the time-axis stage makes of (data, flags)
    tw, to = torch.empty((W, T, F), device=dev), torch.empty((W, T, F), device=dev)
    ms0 = C.c_float(0)
    _lib.check(lib.tri_bench_boxfilter(data.data_ptr(), f4.data_ptr(), tw.data_ptr(), to.data_ptr(), W, T, F, a.time_radius, 0, 0, 1,
                                       C.byref(ms0), torch.cuda.current_stream().cuda_stream))
    both[:, 0] = tw
    both[:, 1] = to
    del tw, to
if a.stage == 1 and a.tiny_lines > 0:
    pick = torch.rand((W, T), generator=g, device=dev) < a.tiny_lines
    col = both[:, 0, :, 100]
    col[pick] = 1e-30
    both[:, 0, :, 100] = col
    print("tiny weights on %d of %d lines" % (int(pick.sum().item()), W * T))
ow = [torch.empty((W, T, F), device=dev) for _ in range(2)]
oo = [torch.empty((W, T, F), device=dev) for _ in range(2)]
ms = C.c_float(0)
st = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in a.variants.split(",")]
for r in [int(x) for x in a.radii.split(",")]:
    best = {}
    for rnd in range(a.rounds):
        for k, v in enumerate(variants):
            _lib.check(lib.tri_bench_boxfilter(data.data_ptr(), (f4 if a.stage == 0 else wimg).data_ptr(), ow[k % 2].data_ptr(), oo[k % 2].data_ptr(),
                                               W, T, F, r, a.stage, v, 3, C.byref(ms), st))
            best[v] = min(best.get(v, 1e9), ms.value)
            if v == 4 and rnd == 0:
                p_, q_ = C.c_uint64(0), C.c_uint64(0)
                lib.tri_boxx_last_stats(C.byref(p_), C.byref(q_))
                print("   r=%d exact row filter: %d line passes, %d redone sequentially" % (r, p_.value, q_.value))
    same = ""
    if len(variants) == 2:
        eq = lambda x, y: bool(((x.view(torch.int32) == y.view(torch.int32)) | (x.isnan() & y.isnan())).all())
        same = "identical" if ((a.stage == 1 or eq(ow[0], ow[1])) and eq(oo[0], oo[1])) else "DIFFERENT"
    gb = W * T * F * 16 / 1e9
    print("r=%3d  " % r + "  ".join("v%d %.2f ms (%.2f TB/s @16B)" % (v, best[v], gb / best[v]) for v in variants) + "  " + same, flush=True)
