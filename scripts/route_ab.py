"""Full-slab A/B of the kernel routes: the flags of one 252-baseline stage-1 call must not depend on which
filter / median / interpolation kernels produced them.  Runs bench's slab once per knob set in a subprocess
(the knobs are read once per process) and compares a 64-bit digest of the output.
    python scripts/route_ab.py [--bl 252] [--params stage1]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch, bench, tricolour_amd
dev = torch.device("cuda", 0)
bl, params = int(sys.argv[1]), sys.argv[2]
vis, flags = bench.synth_slab(torch, bl, 4, 1024, 4096, dev, 1234)
out = tricolour_amd.sum_threshold_flagger(vis, flags, **bench.PARAM_SETS[params])
b = out.view(torch.uint8).reshape(-1)
g = torch.Generator(device=dev); g.manual_seed(99)
digest = 0
for i in range(0, b.numel(), 1 << 28):
    c = b[i:i + (1 << 28)].to(torch.int64)
    w = torch.randint(1, 1 << 30, (c.numel(),), generator=g, device=dev, dtype=torch.int64)
    digest = (digest + int((c * w).sum().item())) & ((1 << 63) - 1)
print("DIGEST %%d %%d" %% (digest, int(b.sum().item())))
''' % ROOT
KNOBS = [(), ("TRI_FILTER_NO_PIPE_T", "TRI_FILTER_NO_PIPE_F"), ("TRI_MEDIAN_NO_PREDICT", "TRI_INTERP_ONE_PASS", "TRI_SPEC_NO_PIPE"),
         ("TRI_NO_FUSED_BEGIN", "TRI_NO_FT_SPEC_OR", "TRI_NO_FUSED_DILATE", "TRI_NO_FUSED_REJECT"),
         ("TRI_FILTER_NO_REGRING", "TRI_FILTER_NO_REGRING_F", "TRI_FILTER_NO_PIPE_T", "TRI_FILTER_NO_PIPE_F"),
         ("TRI_FILTER_NO_EXACT", "TRI_ST_NO_PIPE"), ("TRI_BOXX_NTI=256",), ("TRI_FILTER_NO_TF_REJECT",),
         ("TRI_FILTER_PIPE_T_B8=0", "TRI_FILTER_PIPE_F_B8=0"),
         # round 4: integer weight filter, column panels, one-pass rejection (tile form / off / one workgroup per block / forced redo), wave medians
         ("TRI_FILTER_NO_BOXW", "TRI_ST_NO_PANEL"), ("TRI_NO_TILE_MEDREJ",), ("TRI_FUSED_MEDREJ",), ("TRI_MEDREJ_FORCE_FALLBACK",),
         ("TRI_MEDIAN_WAVE_OLD",)]
ap = argparse.ArgumentParser(); ap.add_argument("--bl", type=int, default=252); ap.add_argument("--params", default="stage1")
a = ap.parse_args()
ref = None
for ks in KNOBS:
    env = dict(os.environ)
    for k in ks:
        name, _, value = k.partition("=")
        env[name] = value or "1"
    r = subprocess.run([sys.executable, "-c", CHILD, str(a.bl), a.params], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
    if not line:
        print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
    print("%-100s %s" % ("+".join(ks) or "(default routes)", line[0]), flush=True)
    ref = ref or line[0]
    if line[0] != ref:
        print("ROUTES DISAGREE"); sys.exit(1)
print("all routes agree")
