"""Soak / determinism check on the full bench slab: N calls on the same inputs must give identical flags.
    python scripts/soak.py [calls] [stage1|defaults|very_broad]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, tricolour_amd
dev = torch.device("cuda", 0)
vis, flags = bench.synth_slab(torch, 252, 4, 1024, 4096, dev, 1234)
ref = None
kw = bench.PARAM_SETS[sys.argv[2]] if len(sys.argv) > 2 else {}
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t0 = time.time(); out = tricolour_amd.sum_threshold_flagger(vis, flags, **kw); torch.cuda.synchronize(); dt = time.time() - t0
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    print("call %d: %.3f s, flagged %.6f, identical to call 0: %s, HBM in use %.1f GB" % (
        k, dt, out.float().mean().item(), same, torch.cuda.memory_allocated(dev) / 1e9), flush=True)
    assert same
    del out
print("OK")
