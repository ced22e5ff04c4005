"""Practical HBM rates on this device (torch copy / read-reduce / fill), for context next to the 8 TB/s nominal peak."""
import torch, time
dev = torch.device("cuda:0")
n = 2 * 1024**3  # floats: 8 GiB
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
dt = t(lambda: b.copy_(a)); print("copy   (R+W) %.2f TB/s" % (2 * n * 4 / dt / 1e12))
dt = t(lambda: a.sum());    print("reduce (R)   %.2f TB/s" % (n * 4 / dt / 1e12))
dt = t(lambda: b.zero_());  print("fill   (W)   %.2f TB/s" % (n * 4 / dt / 1e12))
dt = t(lambda: torch.add(a, 1.0, out=b)); print("add    (R+W) %.2f TB/s" % (2 * n * 4 / dt / 1e12))
