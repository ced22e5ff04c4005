"""D2H rates on the GPU box: 268 MB / 2 GB, into pageable, pinned (torch) and hipHostMalloc'ed memory, bool and uint8."""
import os, sys, time
import numpy as np, torch
dev = torch.device("cuda", 0)
for nbytes in (268435456, 2147483648):
    src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pageable = torch.empty(nbytes, dtype=torch.uint8); pageable.fill_(1)
    for name, dst in (("pinned", pinned), ("pageable (touched)", pageable)):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.time(); dst.copy_(src, non_blocking=False); torch.cuda.synchronize(); dt = time.time() - t0
        print("D2H %4d MB -> %-20s %.1f ms  %.1f GB/s" % (nbytes >> 20, name, dt * 1e3, nbytes / dt / 1e9), flush=True)
    t0 = time.time(); x = src.cpu(); dt = time.time() - t0
    print("D2H %4d MB -> fresh .cpu()          %.1f ms  %.1f GB/s" % (nbytes >> 20, dt * 1e3, nbytes / dt / 1e9), flush=True)
    srcb = src.view(torch.bool)
    t0 = time.time(); x = srcb.cpu().numpy(); dt = time.time() - t0
    print("D2H %4d MB bool -> .cpu().numpy()   %.1f ms  %.1f GB/s" % (nbytes >> 20, dt * 1e3, nbytes / dt / 1e9), flush=True)
    # H2D for comparison
    for name, s in (("pinned", pinned), ("pageable", pageable)):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.time(); src.copy_(s, non_blocking=False); torch.cuda.synchronize(); dt = time.time() - t0
        print("H2D %4d MB <- %-20s %.1f ms  %.1f GB/s" % (nbytes >> 20, name, dt * 1e3, nbytes / dt / 1e9), flush=True)
