"""Fused block-median + rejection kernel (K3r): how many blocks took which path on a synthetic slab (stage-1 kwargs)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import tricolour_amd
from tricolour_amd import _lib
dev = torch.device("cuda", 0)
nbl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vis, flags = bench.synth_slab(torch, nbl, 4, 1024, 4096, dev, 1234)
kw = dict(bench.PARAM_SETS["stage1"], num_major_iterations=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
out = (C.c_uint64 * 20)()
_lib.check(_lib.lib().tri_medrej_stats(out, 1))
tricolour_amd.sum_threshold_flagger(vis, flags, **kw)
torch.cuda.synchronize()
_lib.check(_lib.lib().tri_medrej_stats(out, 1))
print("K3r blocks %d / fallbacks %d %d; K3t second rounds %d; K3t blocks redone by one workgroup, by reason:" % tuple(out[:4]),
      {k - 4: int(out[k]) for k in range(4, 20) if out[k]})
