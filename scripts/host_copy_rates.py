"""Host-side transfer components on this box: pageable->pinned memcpy, hipHostRegister, pinned / pageable H2D."""
import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 1 << 30
src = np.random.randint(0, 255, n, dtype=np.uint8)
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pn = pin.numpy()
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best / 1e9
print("np.copyto pageable -> pinned, 1 thread: %.1f GB/s" % t(lambda: np.copyto(pn, src)))
for k in (2, 4, 8):
    pool = ThreadPoolExecutor(k)
    step = n // k
    f = lambda: list(pool.map(lambda i: np.copyto(pn[i * step:(i + 1) * step], src[i * step:(i + 1) * step]), range(k)))
    print("np.copyto pageable -> pinned, %d threads: %.1f GB/s" % (k, t(f)))
print("H2D from pinned: %.1f GB/s" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("H2D from pageable (driver staging): %.1f GB/s" % t(lambda: dev.copy_(torch.from_numpy(src))))
print("D2H to pinned: %.1f GB/s" % t(lambda: pin.copy_(dev, non_blocking=True)))
rt = torch.cuda.cudart()
def reg():
    r = rt.cudaHostRegister(src.ctypes.data, n, 0)
    tt = torch.from_numpy(src)
    dev.copy_(tt, non_blocking=True)
    torch.cuda.synchronize()
    rt.cudaHostUnregister(src.ctypes.data)
try:
    print("hipHostRegister + H2D + unregister: %.1f GB/s" % t(reg))
except Exception as e:
    print("hipHostRegister failed:", e)
