"""BASELINE config 4 in one scan: static mask -> flag_autos -> uvcontsub -> sum_threshold
(default.yaml kwargs), device-resident, on a slab of `--bl` baselines."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tricolour_amd.strategies import apply_strategies
ap = argparse.ArgumentParser(); ap.add_argument("--bl", type=int, default=64); a = ap.parse_args()
dev = torch.device("cuda", 0)
T, F, ncorr = 1024, 4096, 4
vis, flags = bench.synth_slab(torch, a.bl, ncorr, T, F, dev, 1234)
nant = 64
a1, a2 = np.triu_indices(nant, 0)
ubl = np.stack([np.arange(a.bl), a1[:a.bl], a2[:a.bl]], axis=1)
ants = np.random.RandomState(0).uniform(-4000, 4000, size=(nant, 3))
cf = np.linspace(0.856e9, 1.712e9, F); cw = np.full(F, cf[1] - cf[0])
masks = [cf[np.random.RandomState(1).choice(F, 300, replace=False)][:, None] + 10.0]
strategies = [dict(task="flag_nans_zeros"),
              dict(task="apply_static_mask", kwargs=dict(accumulation_mode="or", uvrange="")),
              dict(task="flag_autos"),
              dict(task="uvcontsub_flagger", kwargs=dict(major_cycles=7, or_original_from_cycle=1, taylor_degrees=20, sigma=15.0)),
              dict(task="sum_threshold", kwargs=bench.PARAM_SETS["stage1"])]
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = apply_strategies(strategies, flags, vis, ubl=ubl, ant_pos=ants, chan_freq=cf, chan_width=cw, masked_channels=masks)
    torch.cuda.synchronize(); dt = time.time() - t0
print("config-4 chain (%d bl x 4 corr x 1024 x 4096): %.2f s -> %.0f Mvis/s, flagged %.3f" % (a.bl, dt, vis.numel() / dt / 1e6, out.float().mean().item()))
