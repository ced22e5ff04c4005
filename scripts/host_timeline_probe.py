"""Timeline of the numpy-in / numpy-out call from N threads: when does each phase of each call start and end?"""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tricolour_amd
from tricolour_amd import flagging
T, F, ncorr, bl = 1024, 4096, 4, 16
rs = np.random.RandomState(0)
shape = (bl, ncorr, T, F)
vis = np.empty(shape, np.complex64); vis.real = rs.standard_normal(shape); vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
flagging.set_num_threads(N)
streams = {}
log = []
T0 = [0.0]
def call(i):
    tid = threading.get_ident()
    s = streams.setdefault(tid, torch.cuda.Stream(dev))
    ev = []
    with torch.cuda.stream(s):
        t = time.time(); v = torch.from_numpy(vis).to(dev, non_blocking=True); f = torch.from_numpy(flags).to(dev, non_blocking=True); s.synchronize(); ev.append(("h2d", t, time.time()))
        t = time.time(); o = tricolour_amd.sum_threshold_flagger(v, f); ev.append(("enqueue", t, time.time())); s.synchronize(); ev.append(("kernels", t, time.time()))
        t = time.time(); x = o.cpu().numpy(); ev.append(("d2h", t, time.time()))
    log.append((i, tid, ev))
with ThreadPoolExecutor(N) as pool:
    list(pool.map(call, range(N)))
    log.clear()
    T0[0] = time.time()
    list(pool.map(call, range(2 * N)))
    total = time.time() - T0[0]
tids = sorted({t for _, t, _ in log})
for i, tid, ev in sorted(log):
    print("call %d thread %d: " % (i, tids.index(tid)) + "  ".join("%s %.0f-%.0f" % (n, (a - T0[0]) * 1e3, (b - T0[0]) * 1e3) for n, a, b in ev))
print("total %.0f ms for %d blocks -> %.0f Mvis/s" % (total * 1e3, 2 * N, 2 * N * vis.size / total / 1e6))
