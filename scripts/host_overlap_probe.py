"""Where does the numpy-in / numpy-out path lose its overlap?  (GPU box)
 a) pageable H2D copies of one 16-baseline block from 1, 2, 4 threads at once (aggregate GB/s)
 b) the kernels of a device-resident block, alone
 c) one thread copying blocks H2D while another runs the kernels on resident inputs: overlap or serial?
 d) the same with the copy thread reading from PINNED memory"""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tricolour_amd
T, F, ncorr, bl = 1024, 4096, 4, 16
rs = np.random.RandomState(0)
shape = (bl, ncorr, T, F)
vis = np.empty(shape, np.complex64); vis.real = rs.standard_normal(shape); vis.imag = rs.standard_normal(shape)
flags = rs.uniform(size=shape) < 0.02
dev = torch.device("cuda", 0)
gb = (vis.nbytes + flags.nbytes) / 1e9

def h2d(src_v, src_f, reps):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(reps):
            a = src_v.to(dev, non_blocking=True); b = src_f.to(dev, non_blocking=True)
        s.synchronize()

tv, tf = torch.from_numpy(vis), torch.from_numpy(flags)
h2d(tv, tf, 1)
for n in (1, 2, 4):
    with ThreadPoolExecutor(n) as pool:
        t0 = time.time(); list(pool.map(lambda i: h2d(tv, tf, 2), range(n))); dt = time.time() - t0
    print("a) pageable H2D, %d threads: %.1f GB/s aggregate" % (n, n * 2 * gb / dt), flush=True)
pv, pf = tv.pin_memory(), tf.pin_memory()
for n in (1, 2):
    with ThreadPoolExecutor(n) as pool:
        t0 = time.time(); list(pool.map(lambda i: h2d(pv, pf, 2), range(n))); dt = time.time() - t0
    print("a') pinned H2D, %d threads: %.1f GB/s aggregate" % (n, n * 2 * gb / dt), flush=True)

dv, df = tv.to(dev), tf.to(dev)
def kernels(reps):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(reps):
            o = tricolour_amd.sum_threshold_flagger(dv, df)
        s.synchronize()
kernels(2)
t0 = time.time(); kernels(4); tk = (time.time() - t0) / 4
print("b) kernels of one resident block: %.1f ms" % (tk * 1e3), flush=True)
t0 = time.time(); h2d(tv, tf, 4); tc = (time.time() - t0) / 4
print("   pageable H2D of one block: %.1f ms" % (tc * 1e3), flush=True)
for name, sv, sf in (("c) pageable", tv, tf), ("d) pinned", pv, pf)):
    th = threading.Thread(target=h2d, args=(sv, sf, 4))
    t0 = time.time(); th.start(); kernels(4); th.join(); dt = time.time() - t0
    print("%s copies + kernels concurrently, 4 blocks each: %.1f ms (serial would be %.1f, perfect overlap %.1f)" %
          (name, dt * 1e3, 4 * (tk + tc) * 1e3, 4 * max(tk, tc) * 1e3), flush=True)
# e) D2H of the flags into fresh numpy vs preallocated
o = tricolour_amd.sum_threshold_flagger(dv, df); torch.cuda.synchronize()
t0 = time.time(); x = o.cpu().numpy(); t1 = time.time() - t0
buf = torch.empty(o.shape, dtype=torch.bool).pin_memory()
t0 = time.time(); buf.copy_(o); torch.cuda.synchronize(); t2 = time.time() - t0
print("e) D2H of the flags: fresh numpy %.1f ms, pinned buffer %.1f ms" % (t1 * 1e3, t2 * 1e3), flush=True)
