#!/usr/bin/env python3
"""Headline benchmark: Mvis/s flagged by sum_threshold_flagger on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload slab|chain|ska] [--params stage1|defaults|very_broad]

Workloads (BASELINE.json configs):
  slab   (default) one sum_threshold_flagger call per step over one HBM-resident slab of the MeerKAT-64
         configuration (configs[1]: 2016 bl x 4 corr x 1024 time x 4096 chan).  The full window set (270.6 GB of
         complex64 + 33.8 GB of flags) exceeds 288 GB of HBM, so it is processed as baseline slabs; every rank
         holds one slab of --bl baselines (weak scaling: with the default 252 baselines per rank, 8 ranks
         together hold exactly the 2016-baseline configuration = configs[2]).
  chain  configs[3]: per step S scans (independent datasets, apps/tricolour/app.py:295-313, 370), each run
         through flag_nans_zeros -> apply_static_mask(or) -> flag_autos -> uvcontsub_flagger -> sum_threshold
         (stage-1 kwargs) with the executor's replace / OR rules (strat_executor.py:39-78), device-resident.
  ska    configs[4] geometry: (bl, 2, 512, 65536) windows streamed slab by slab through two pinned host
         buffers with H2D / kernels / D2H overlapped on three streams; a step = one slab.  `value` is the
         device-resident rate over the same slabs; the PCIe-inclusive sustained rate is reported beside it.

Baselines are independent (flagging.py:765-774): ranks own contiguous baseline slabs and only meet at the
timing barriers.  For N > 1 a separately timed leg scatters a small slab set from rank 0 and gathers the
flags back over RCCL (tricolour_amd.distributed), which is how a single-root data set would be fanned out.

`--gpus N` without a torchrun environment starts the N ranks itself (python -m torch.distributed.run ...)
before any GPU call and relays rank 0's line.

Prints ONE JSON line (rank 0): metric / value = whole-job Mvis/s with inputs already resident in HBM, plus
  "roofline"      ONE object: the kernel with the largest share of the step (the fused frequency-axis box
                  filter stage), timed live with HIP events on its launch stream in the step's own launch
                  geometry; "roofline_kernels" lists every kernel measured that way (SumThreshold included);
  "parity_check"  the HIP path on the windows the cpu_baseline leg fed the oracle: {windows, vis, mismatches};
  "cpu_baseline"  the CPU oracle (a C restatement of the reference's numba path -- numba itself cannot run
                  here) on a bounded sample of the same workload on this box's host cores (rank 0, N=1);
  "other_params"  the same slab with the other shipped parameter sets (slab workload, N=1);
  "other_workloads"  short `chain` (configs[3]) and `ska` (configs[4], 1-GPU half) legs with their own parity_check,
                  so that the one command the driver times carries every single-GPU configuration (slab workload, N=1).
A parity_check that fails (any mismatch on a bit-exact leg, < 99.9 % agreement on the chain leg) is reported in the
line ("parity_ok": false) and the process exits with code 4 after printing it.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PARAM_SETS = {
    # library defaults (flagging.py:1076-1083)
    "defaults": dict(),
    # conf/default.yaml:20-35 "background_flags" -- the heaviest shipped stage (the headline)
    "stage1": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                   background_reject=2.0, background_iterations=5, spike_width_time=12.5,
                   spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                   average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                   num_major_iterations=5),
    # conf/default.yaml:59-73 "final_st_very_broad"
    "very_broad": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8],
                       windows_freq=[32, 48, 64, 128], background_reject=2.0,
                       background_iterations=5, spike_width_time=6.5, spike_width_freq=64.0,
                       time_extend=3, freq_extend=3, freq_chunks=10, average_freq=1,
                       flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                       num_major_iterations=1),
    # conf/default.yaml:74-89 "final_st_broad"
    "broad": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                  background_reject=2.0, background_iterations=5, spike_width_time=6.5,
                  spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                  average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                  num_major_iterations=1),
    # conf/default.yaml:90-105 "final_st_narrow" (spike_width_time arrives as the integer 2)
    "narrow": dict(outlier_nsigma=10, windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                   background_reject=2.0, background_iterations=5, spike_width_time=2,
                   spike_width_freq=10.0, time_extend=3, freq_extend=3, freq_chunks=10,
                   average_freq=1, flag_all_time_frac=0.6, flag_all_freq_frac=0.8, rho=1.3,
                   num_major_iterations=1),
}
# conf/default.yaml:37-45 "residual_flag_initial"
UVCONTSUB_KW = dict(major_cycles=7, or_original_from_cycle=1, taylor_degrees=20, sigma=15.0)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ST_BYTES_PER_SAMPLE = 5  # fused SumThreshold pass: 4 B residual in + 1 B flag out
# box filter, per axis stage of one masked filter (both images): SURVEY 8(d) prices a stage at
# 2 arrays x (4 R + 4 W) = 16 B/sample; the fused frequency stage also reads the amplitudes (4 B) and writes one
# image instead of two: 8 R + 4 R + 4 W = 16 B/sample
BOX_BYTES_PER_SAMPLE = 16
# the time-axis stage builds its weight image from the packed flag words (1 B/sample) instead of a 4 B float
# image: 4 B data + 1 B flags read, 2 x 4 B written = 13 B/sample actually required (PMC 0.875 x 16)
BOX_TIME_BYTES_PER_SAMPLE = 13


def box_radius(sigma):
    """int(0.5 * sqrt(12 sigma^2 / passes + 1)), flagging.py:451 (passes = 4)"""
    return int(0.5 * np.sqrt(12.0 * sigma * sigma / 4.0 + 1.0))


def synth_slab(torch, nbl, ncorr, T, F, device, seed):
    """SURVEY.md 8(d) synthetic inputs, generated on the device in pieces:
    complex Gaussian noise (sigma 1) plus deterministic RFI; 2 % of channels
    pre-flagged; 1e-5 NaN samples."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    vis = torch.empty((nbl, ncorr, T, F), dtype=torch.complex64, device=device)
    flags = torch.zeros((nbl, ncorr, T, F), dtype=torch.bool, device=device)
    bad_chan = torch.randperm(F, generator=g, device=device)[: max(1, F // 100)]
    bad_time = torch.randperm(T, generator=g, device=device)[: max(1, T // 200)]
    pre_chan = torch.randperm(F, generator=g, device=device)[: max(1, F // 50)]
    blk0 = int(F * 0.6)
    for b in range(nbl):
        re = torch.randn((ncorr, T, F), generator=g, device=device)
        im = torch.randn((ncorr, T, F), generator=g, device=device)
        re[:, :, bad_chan] += 8.0
        re[:, bad_time, :] += 6.0
        re[:, :, blk0:blk0 + 20] += 3.0
        n = ncorr * T * F
        nspike = max(1, n // 10000)
        idx = torch.randint(0, n, (nspike,), generator=g, device=device)
        re.view(-1)[idx] += 50.0
        nnan = max(1, n // 100000)
        idx = torch.randint(0, n, (nnan,), generator=g, device=device)
        re.view(-1)[idx] = float("nan")
        vis[b] = torch.complex(re, im)
        del re, im
    flags[:, :, :, pre_chan] = True
    return vis, flags


def synth_host_windows(nwin, T, F, seed=1234):
    rs = np.random.RandomState(seed)
    vis = np.empty((nwin, 1, T, F), np.complex64)
    vis.real = rs.standard_normal((nwin, 1, T, F)).astype(np.float32)
    vis.imag = rs.standard_normal((nwin, 1, T, F)).astype(np.float32)
    vis.real[..., ::97] += 8.0
    vis.real[:, :, ::211, :] += 6.0
    flags = np.zeros(vis.shape, np.bool_)
    flags[..., ::50] = True
    return vis, flags


def cpu_baseline(kw, T, F, seconds_budget=25.0, chain=None, fixed_windows=None):
    """Oracle (C restatement of the reference CPU path, OpenMP over windows) on a bounded sample of the same
    workload.  `chain`: also run the cheap steps and uvcontsub (numpy restatements) in front.  `fixed_windows`: take
    that many windows (capped at the usable cores) without the one-window calibration run."""
    from oracle import oracle
    oracle.set_modes(oracle.POW_SQMUL, oracle.INTERP_F64)
    nproc = os.cpu_count() or 1
    try:
        nproc = len(os.sched_getaffinity(0))     # the cores this process may run on
    except (AttributeError, OSError):
        pass
    # ... and the CPU time it may use: a container's cgroup quota (cpu.max: "<quota> <period>" in microseconds) can be
    # far below the host's core count -- threads beyond it only take turns (256 threads on a 16-core share: 10 x slower)
    quota = None
    for path, v2 in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            txt = open(path).read().split()
            if v2 and txt[0] != "max":
                quota = float(txt[0]) / float(txt[1])
            elif not v2 and float(txt[0]) > 0:
                quota = float(txt[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = nproc if quota is None else max(1, min(nproc, int(quota + 0.5)))
    cores = int(os.environ.get("TRI_BENCH_CPU_THREADS", usable))

    def run(vis, flags, threads):
        t0 = time.time()
        if chain is not None:
            f = oracle.flag_nans_and_zeros(vis, flags)
            f = oracle.uvcontsub_flagger(vis, f, **chain)
            out = oracle.sum_threshold_flagger(vis, f, n_threads=threads, **kw)
            out |= f
        else:
            out = oracle.sum_threshold_flagger(vis, flags, n_threads=threads, **kw)
        return time.time() - t0, out

    # calibrate on one window with one thread, then size the sample: every thread gets the same number of
    # windows, the whole sample stays within ~seconds_budget of wall time and 128 windows (4.3 GB of vis)
    if fixed_windows:
        nwin = max(1, min(int(fixed_windows), cores))
        cores = nwin
    else:
        vis, flags = synth_host_windows(1, T, F)
        t1, _ = run(vis, flags, 1)
        per_thread = max(1, min(2, int(seconds_budget / max(t1, 1e-3))))
        nwin = cores * per_thread
        while nwin > 128 and per_thread > 1:
            per_thread -= 1
            nwin = cores * per_thread
        nwin = min(nwin, 256)
    vis, flags = synth_host_windows(nwin, T, F)
    dt, out = run(vis, flags, cores)
    what = "chain (flag_nans_zeros, uvcontsub, sum_threshold)" if chain is not None else "same kwargs"
    res = dict(value=round(nwin * T * F / dt / 1e6, 3), unit="Mvis/s", cores=cores, nproc=nproc,
               cpu_quota=None if quota is None else round(quota, 2), kind="port",
               sample="%d windows of %dx%d (1 corr), %s, %.1f s on %d threads (host shows %d cores, cgroup CPU quota %s); "
                      "C restatement of the reference numba path (oracle/), OpenMP over windows"
                      % (nwin, T, F, what, dt, cores, nproc, "none" if quota is None else "%.1f cores" % quota))
    return res, vis, flags, out


def parity_check(torch, tricolour_amd, device, kw, vis, flags, expected, chain=None):
    """The HIP path on the very windows the cpu_baseline leg fed the oracle (outside the timed region; the
    oracle is the checker, never the thing measured).  Bit-exact is the bar for sum_threshold_flagger; the chain
    goes through uvcontsub's FFT, whose summation order differs from numpy's (agreement rate)."""
    dv = torch.from_numpy(vis).to(device)
    df = torch.from_numpy(flags).to(device)
    if chain is not None:
        from tricolour_amd import flagging
        f = flagging.flag_nans_and_zeros(dv, df)
        f = flagging.uvcontsub_flagger(dv, f, **chain)
        got = tricolour_amd.sum_threshold_flagger(dv, f, **kw) | f
    else:
        got = tricolour_amd.sum_threshold_flagger(dv, df, **kw)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    bad = int((got != expected).sum())
    ok = bad == 0 if chain is None else bad <= 1e-3 * vis.size
    return dict(windows=int(vis.shape[0] * vis.shape[1]), vis=int(vis.size), mismatches=bad, parity_ok=bool(ok),
                flagged_oracle=int(expected.sum()), flagged_hip=int(got.sum()),
                expected="bit-exact" if chain is None else ">= 99.9 % agreement (uvcontsub FFT order, SURVEY 8f-2)",
                checker="oracle/ (C restatement of the reference numba path), same windows and kwargs as cpu_baseline")


def _pmc_traffic(name, samples, device_kernels):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of the same launch (FETCH_SIZE x2 per the gfx950
    note + WRITE_SIZE); bench.py cannot collect counters itself.  A file only counts when it was recorded for the SAME
    device kernels (the symbols this leg just launched), the same launch size and the same build of the kernel sources
    (`lib_sha16`); anything else gives traffic = null instead of a stale number (ADVICE r2 / VERDICT r3)."""
    from tricolour_amd import _lib
    sha = _lib.source_hash()
    for fn in ("r04_pmc_%s.json" % name,):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except Exception:
            continue
        if (pmc.get("samples_per_launch") == samples and sorted(pmc.get("device_kernels", [])) == sorted(device_kernels)
                and pmc.get("lib_sha16") == sha):
            return pmc["hbm_bytes_per_launch"], "profiles/" + fn
    return None, None


def _kernel_family(sym):
    return sym.split("<", 1)[0].strip()


def roofline_sumthreshold(torch, device, T, F, kw, nwin):
    """Times the fused SumThreshold column kernel alone, in the launch geometry
    of the step's time-axis pass (line = time, coalesced columns = channels, all
    `nwin` windows of the slab in one launch), with HIP events on its stream."""
    from tricolour_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device=device)
    g.manual_seed(7)
    data = torch.randn((nwin, T, F), generator=g, device=device)
    data[:, :, ::101] += 9.0
    mad = torch.full((nwin, F), 0.6745, dtype=torch.float64, device=device)
    out = torch.empty((nwin, T, F), dtype=torch.uint8, device=device)
    wins = kw.get("windows_time", [1, 2, 4, 8])
    warr = (C.c_int64 * len(wins))(*[int(w) for w in wins])
    ms = C.c_float(0)
    stream = torch.cuda.current_stream(device).cuda_stream
    names = {}
    for reps in (2, 8):
        _lib.kernel_log_begin()
        _lib.check(lib.tri_bench_sumthreshold(data.data_ptr(), mad.data_ptr(), out.data_ptr(),
                                              nwin, T, F, warr, len(wins),
                                              float(kw.get("outlier_nsigma", 4.5)),
                                              float(kw.get("rho", 1.3)),
                                              int(os.environ.get("TRI_BENCH_ST_VARIANT", "0")), reps,
                                              C.byref(ms), stream))
        names = _lib.kernel_log_end()
    samples = nwin * T * F
    achieved = samples * ST_BYTES_PER_SAMPLE / (ms.value * 1e-3) / 1e9
    devk = sorted(names)
    traffic, src = _pmc_traffic("sumthreshold", samples, devk)
    return dict(bound="hbm", kernel="fused SumThreshold (all windows in one pass), time axis: " + ", ".join(devk),
                device_kernels=devk, launches_per_measurement=1,
                timed="the kernel alone through the library's measurement hook tri_bench_sumthreshold, in the step's launch "
                      "geometry, HIP events on the launch stream (not inside the step)",
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=src,
                algorithmic_bytes_per_launch=samples * ST_BYTES_PER_SAMPLE,
                bytes_per_sample=ST_BYTES_PER_SAMPLE, samples_per_launch=samples,
                ms_per_launch=round(ms.value, 4))


def roofline_reject(torch, device, T, F, kw, nwin):
    """Times one rejection step of the background loop (block median of |data - background| + threshold comparison,
    flagging.py:553-574) in the step's geometry: all `nwin` windows, `freq_chunks` blocks per window, through the library's
    measurement hook (k_mr_predict, two rounds of k_mr_pass + k_mr_finish, the redo kernel).  7 B per sample in the pass
    (4 B residual + 1 B flag read, 1 B flag + 1 B transposed flag written) + the lists and patches of the other kernels."""
    from tricolour_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device=device)
    g.manual_seed(11)
    resid = torch.randn((nwin, F, T), generator=g, device=device).abs_()
    fin = (torch.rand((nwin, F, T), generator=g, device=device) < 0.05).view(torch.uint8)
    fout = torch.empty((nwin, F, T), dtype=torch.uint8, device=device)
    ft4 = torch.empty((nwin, T // 4, F, 4), dtype=torch.uint8, device=device)
    nchunk = int(kw.get("freq_chunks", 10))
    ends = [int(x) for x in np.linspace(0, F, nchunk + 1)]      # (flagging.py:936: np.linspace(0, nfreq, freq_chunks + 1, dtype=int))
    med = torch.empty((nwin, nchunk), dtype=torch.float64, device=device)
    earr = (C.c_int64 * len(ends))(*ends)
    ms = C.c_float(0)
    stream = torch.cuda.current_stream(device).cuda_stream
    names = {}
    for reps in (2, 6):
        _lib.kernel_log_begin()
        rc = lib.tri_bench_reject(resid.data_ptr(), fin.data_ptr(), fout.data_ptr(), ft4.data_ptr(), med.data_ptr(), nwin, F, T,
                                  earr, len(ends), float(kw.get("background_reject", 2.0)), reps, C.byref(ms), stream)
        names = _lib.kernel_log_end()
        if rc == _lib.TRI_EUNSUPPORTED:
            return None
        _lib.check(rc)
    samples = nwin * T * F
    bps = 7
    achieved = samples * bps / (ms.value * 1e-3) / 1e9
    devk = sorted(names)
    per_meas = sum(names.values()) // 6 if names else 1
    traffic, src = _pmc_traffic("reject", samples, devk)
    return dict(bound="hbm", kernel="rejection step of the background loop (block median + threshold in one pass): " + ", ".join(devk),
                device_kernels=devk, launches_per_measurement=max(per_meas, 1),
                timed="the whole step (six launches) through the library's measurement hook tri_bench_reject, in the step's launch geometry, "
                      "HIP events on the launch stream",
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=traffic, traffic_source=src, algorithmic_bytes_per_launch=samples * bps, bytes_per_sample=bps,
                samples_per_launch=samples, ms_per_launch=round(ms.value, 4))


def roofline_boxfilter(torch, device, T, F, kw, nwin):
    """Times the two stages of one masked box filter (flagging.py:469-513) in the step's launch geometry at the
    radii of the parameter set: the time-axis stage at its largest radius (first background iteration) and the
    fused frequency-axis stage + masked division at its last radius (the final pass of every background)."""
    from tricolour_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device=device)
    g.manual_seed(9)
    data = torch.randn((nwin, T, F), generator=g, device=device).abs_()
    wimg = torch.rand((nwin, 2, T, F), generator=g, device=device)    # stage 1 input: weight, weight * data per window
    f4 = (torch.rand((nwin, T // 4, F), generator=g, device=device) < 0.05).to(torch.int32) * 0x01010101
    ow = torch.empty((nwin, T, F), device=device)
    oo = torch.empty((nwin, T, F), device=device)
    ms = C.c_float(0)
    stream = torch.cuda.current_stream(device).cuda_stream
    nit = int(kw.get("background_iterations", 1))
    swt, swf = float(kw.get("spike_width_time", 12.5)), float(kw.get("spike_width_freq", 10.0))
    cases = [(0, box_radius(max(nit, 1) * swt), "time-axis stage (k_boxq / k_boxt / k_colfilter_lds), first background iteration"),
             (1, box_radius(swf), "frequency-axis stage fused with the masked division (k_boxf), last background pass")]
    if nit > 1:
        cases.insert(1, (1, box_radius(nit * swf), "frequency-axis stage fused with the masked division (k_boxqf), first background iteration"))
    samples = nwin * T * F
    out = []
    for ci, (stage, rad, what) in enumerate(cases):
        if rad <= 0:
            continue
        bps = BOX_TIME_BYTES_PER_SAMPLE if stage == 0 else BOX_BYTES_PER_SAMPLE
        src = f4 if stage == 0 else wimg
        names = {}
        try:
            for reps in (1, 4):
                _lib.kernel_log_begin()
                _lib.check(lib.tri_bench_boxfilter(data.data_ptr(), src.data_ptr(), ow.data_ptr(), oo.data_ptr(),
                                                   nwin, T, F, rad, stage, 0, reps, C.byref(ms), stream))
                names = _lib.kernel_log_end()
        except (NotImplementedError, ValueError) as e:
            _lib.kernel_log_end()
            out.append(dict(bound="hbm", kernel="box filter %s, r = %d" % (what, rad), error=str(e)))
            continue
        achieved = samples * bps / (ms.value * 1e-3) / 1e9
        devk = sorted(names)
        traffic, tsrc = _pmc_traffic("boxfilter_s%d_r%d" % (stage, rad), samples, devk)
        out.append(dict(bound="hbm", kernel="box filter %s, r = %d: %s" % (what, rad, ", ".join(devk)),
                        device_kernels=devk, launches_per_measurement=len(devk),
                        timed="the stage alone through the library's measurement hook tri_bench_boxfilter (the flagger's own "
                              "route for that radius), in the step's launch geometry, HIP events on the launch stream",
                        achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=tsrc,
                        algorithmic_bytes_per_launch=samples * bps,
                        bytes_per_sample=bps, samples_per_launch=samples,
                        ms_per_launch=round(ms.value, 4)))
    return out


def pick_dominant(kernels, step_launches):
    """The roofline object is the measured kernel whose FAMILY takes the largest share of the timed step:
    share = (launches of that family in one step, from the library's kernel log) / (launches per measurement)
    x (measured ms).  Returns (entry, shares)."""
    fam_count = {}
    for sym, n in step_launches.items():
        fam_count[_kernel_family(sym)] = fam_count.get(_kernel_family(sym), 0) + n
    best, shares = None, []
    for e in kernels:
        if "frac" not in e:
            continue
        fams = sorted(set(_kernel_family(k) for k in e.get("device_kernels", [])))
        launches = sum(fam_count.get(f, 0) for f in fams)
        est = launches / max(e.get("launches_per_measurement", 1), 1) * e["ms_per_launch"]
        e["step_family_launches"] = launches
        e["step_share_ms_estimate"] = round(est, 2)
        shares.append((est, e))
        if best is None or est > best[0]:
            best = (est, e)
    return (best[1] if best else None), shares


# ---------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------
def chain_setup(nbl, F):
    """Host-side inputs of the cheap strategy steps: 64 antennas, the first `nbl` baselines in
    (ant2, ant1) order (packing.py:54-56), a static mask hitting 300 channels."""
    nant = 64
    a1, a2 = np.triu_indices(nant, 0)
    ubl = np.stack([np.arange(nbl), a1[:nbl], a2[:nbl]], axis=1)
    ants = np.random.RandomState(0).uniform(-4000, 4000, size=(nant, 3))
    cf = np.linspace(0.856e9, 1.712e9, F)
    cw = np.full(F, cf[1] - cf[0])
    masks = [cf[np.random.RandomState(1).choice(F, 300, replace=False)][:, None] + 10.0]
    return dict(ubl=ubl, ant_pos=ants, chan_freq=cf, chan_width=cw, masked_channels=masks)


def chain_strategies(kw):
    return [dict(task="flag_nans_zeros"),
            dict(task="apply_static_mask", kwargs=dict(accumulation_mode="or", uvrange="")),
            dict(task="flag_autos"),
            dict(task="uvcontsub_flagger", kwargs=dict(UVCONTSUB_KW)),
            dict(task="sum_threshold", kwargs=kw)]


def ska_stream(torch, tricolour_amd, device, kw, nbl, ncorr, T, F, slabs, warmup, kstreams=2):
    """Streams `slabs` window slabs (nbl x ncorr windows of T x F each) from two pinned host buffers through
    `kstreams` + 1 device buffer sets: stream A copies slab i + 1 host->device while the kernel streams flag slabs i
    (and, with two kernel streams, i - 1) and stream C copies finished flags back.  TWO kernel streams (round 4): the
    1-D spectrum path of these windows is 65536 sequential steps on two workgroups -- 10 % of a slab's time with 254 CUs
    idle -- so consecutive slabs run as two concurrent calls (each its own workspace) and one slab's spectrum kernels
    run under the other's 2-D kernels, the way two dask threads would issue them.  Returns (seconds for `slabs` slabs
    with transfers, seconds for the same slabs device-resident, flagged fraction)."""
    from tricolour_amd import flagging
    shape = (nbl, ncorr, T, F)
    NB = kstreams + 1
    if kstreams > 1:
        flagging.set_num_threads(kstreams)              # every concurrent call sizes its workspace against 1 / kstreams of the device
    # host side: two pinned (vis, flags, out) buffer sets filled once with synthetic data
    vis_d, flags_d = synth_slab(torch, nbl, ncorr, T, F, device, 4321)
    hv = [torch.empty(shape, dtype=torch.complex64).pin_memory() for _ in range(2)]
    hf = [torch.empty(shape, dtype=torch.bool).pin_memory() for _ in range(2)]
    ho = [torch.empty(shape, dtype=torch.bool).pin_memory() for _ in range(2)]
    for k in range(2):
        hv[k].copy_(vis_d)
        hf[k].copy_(flags_d)
    dv = [vis_d] + [torch.empty_like(vis_d) for _ in range(NB - 1)]
    df = [flags_d] + [torch.empty_like(flags_d) for _ in range(NB - 1)]
    douts = [None] * NB
    s_in, s_out = torch.cuda.Stream(device), torch.cuda.Stream(device)
    s_k = [torch.cuda.Stream(device) for _ in range(kstreams)]
    ev_in = [torch.cuda.Event() for _ in range(NB)]     # slab landed in dv/df[b]
    ev_k = [torch.cuda.Event() for _ in range(NB)]      # kernels of buffer b done (input reusable, output ready)
    ev_out = [torch.cuda.Event() for _ in range(2)]     # host output buffer h copied into

    def h2d(i):
        b = i % NB
        with torch.cuda.stream(s_in):
            if i >= NB:
                s_in.wait_event(ev_k[b])                # the kernels that read dv[b] (slab i - NB) are done
            dv[b].copy_(hv[i & 1], non_blocking=True)
            df[b].copy_(hf[i & 1], non_blocking=True)
            ev_in[b].record(s_in)

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h2d(0)
        for i in range(n):
            b = i % NB
            if i + 1 < n:
                h2d(i + 1)
            ks = s_k[i % kstreams]
            with torch.cuda.stream(ks):
                ks.wait_event(ev_in[b])
                douts[b] = tricolour_amd.sum_threshold_flagger(dv[b], df[b], **kw)
                douts[b].record_stream(s_out)
                ev_k[b].record(ks)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_k[b])
                ho[i & 1].copy_(douts[b], non_blocking=True)     # (s_out is one stream: copies into a host buffer are ordered)
                ev_out[i & 1].record(s_out)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(max(warmup, kstreams))
    t_stream = run(slabs)
    # device-resident rate over the same number of slabs (same concurrency, no transfers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [None] * kstreams
    for i in range(slabs):
        with torch.cuda.stream(s_k[i % kstreams]):
            outs[i % kstreams] = tricolour_amd.sum_threshold_flagger(dv[i % NB], df[i % NB], **kw)
    torch.cuda.synchronize()
    t_res = time.perf_counter() - t0
    flagged = float(outs[0].float().mean().item())
    if kstreams > 1:
        flagging.set_num_threads(1)
    return t_stream, t_res, flagged


def scatter_leg(torch, dist, tricolour_amd, device, rank, world, kw, ncorr, T, F, bl_per_rank, backend, root_budget_gb=2.0):
    """Rank 0 holds world x bl_per_rank baselines in (pinned) HOST memory -- the full MeerKAT-64 set of configs[2] is 304 GB
    and cannot sit on one 288 GB GPU -- and streams them out in rounds within `root_budget_gb` of device staging memory
    (tricolour_amd.distributed.scatter_windows_streamed: batched point-to-point sends, RCCL: each peer's piece on its own
    xGMI link, the next round staged while one is on the links); every rank flags its slab, the uint8 flags are gathered
    back.  Separately timed; never part of `value`."""
    from tricolour_amd import distributed as D
    nbl = world * bl_per_rank
    shape = (nbl, ncorr, T, F)
    comm_dev = device if backend == "nccl" else torch.device("cpu")
    hv = hf = None
    if rank == 0:
        vis, flags = synth_slab(torch, nbl, ncorr, T, F, device, 99)
        try:
            hv = torch.empty(shape, dtype=torch.complex64).pin_memory()
            hf = torch.empty(shape, dtype=torch.bool).pin_memory()
        except RuntimeError:
            hv, hf = torch.empty(shape, dtype=torch.complex64), torch.empty(shape, dtype=torch.bool)
        hv.copy_(vis)
        hf.copy_(flags)
        del vis, flags
        torch.cuda.empty_cache()

    def reader(b0, b1):
        return hv[b0:b1], hf[b0:b1]

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    budget = int(root_budget_gb * (1 << 30))
    # untimed pass over one baseline per rank: the point-to-point connections are set up on first use
    wshape = (world, ncorr, T, F)
    wv, wf = D.scatter_windows_streamed(reader if rank == 0 else None, wshape, budget, src=0, device=comm_dev)
    D.gather_flags(wf.view(torch.uint8), wshape, dst=0)
    del wv, wf
    sync()
    stats = {}
    t0 = time.perf_counter()
    v, f = D.scatter_windows_streamed(reader if rank == 0 else None, shape, budget, src=0, device=comm_dev, stats=stats)
    sync()
    t1 = time.perf_counter()
    out = tricolour_amd.sum_threshold_flagger(v.to(device), f.to(device), **kw)
    sync()
    t2 = time.perf_counter()
    full = D.gather_flags(out if backend == "nccl" else out.cpu(), shape, dst=0)
    sync()
    t3 = time.perf_counter()
    # every rank's flag count must arrive at the root unchanged
    cnt = torch.tensor([float(out.sum().item())], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(cnt)
    res = None
    if rank == 0:
        sent = (nbl - bl_per_rank) * ncorr * T * F      # samples that left / re-entered the root
        ok = abs(float(full.sum().item()) - float(cnt.item())) < 0.5
        res = dict(backend="rccl" if backend == "nccl" else backend, baselines_per_rank=bl_per_rank,
                   scatter="streamed from pinned host memory in %d rounds of %d baselines per peer, root staging budget %.1f GiB (peak %.2f GiB)"
                           % (stats.get("rounds", 0), stats.get("baselines_per_round_and_peer", 0), root_budget_gb, stats.get("peak_root_bytes", 0) / 2**30),
                   scatter_gb_s=round(sent * 9 / (t1 - t0) / 1e9, 2), gather_gb_s=round(sent / (t3 - t2) / 1e9, 2),
                   scatter_s=round(t1 - t0, 4), flag_s=round(t2 - t1, 4), gather_s=round(t3 - t2, 4),
                   flag_count_matches=bool(ok))
    del hv, hf, v, f, out, full
    return res


def other_workload_legs(torch, tricolour_amd, flagging, device, kw, pname):
    """Short legs of the two other single-GPU configurations inside the default run, each with its
    own parity_check: `chain` = BASELINE configs[3] (1 step of 2 scans x 42 bl), `ska` = configs[4] geometry (4 slabs
    of 32 bl x 2 corr x 512 x 65536 streamed through pinned buffers + the same slabs device-resident)."""
    from tricolour_amd.strategies import apply_strategies
    legs = {}
    # ---- chain
    nbl, ncorr, T, F, nscans = 42, 4, 1024, 4096, 2
    scans = [synth_slab(torch, nbl, ncorr, T, F, device, 4321 + 17 * s) for s in range(nscans)]
    setup = chain_setup(nbl, F)
    strategies = chain_strategies(kw)

    def step():
        out = None
        for v, f in scans:
            out = apply_strategies(strategies, f, v, **setup)
        return out
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nvis = nscans * nbl * ncorr * T * F
    leg = dict(value=round(nvis / dt / 1e6, 2), unit="Mvis/s", ms_per_step=round(dt * 1e3, 3), steps=1, warmup=1,
               workload="multi-scan chain (BASELINE configs[3]): %d scans x (%d bl x %d corr x %d x %d), each flag_nans_zeros -> "
                        "apply_static_mask(or) -> flag_autos -> uvcontsub_flagger -> sum_threshold (kwargs=%s), device-resident"
                        % (nscans, nbl, ncorr, T, F, pname),
               flagged_fraction=round(float(out.float().mean().item()), 4))
    del scans, out
    flagging.release_workspace()
    torch.cuda.empty_cache()
    cb, hv, hf, hexp = cpu_baseline(kw, T, F, chain=UVCONTSUB_KW, fixed_windows=16)
    leg["cpu_baseline"] = cb
    leg["parity_check"] = parity_check(torch, tricolour_amd, device, kw, hv, hf, hexp, chain=UVCONTSUB_KW)
    del hv, hf, hexp
    flagging.release_workspace()
    torch.cuda.empty_cache()
    legs["chain"] = leg
    # ---- ska
    nbl, ncorr, T, F, slabs = 32, 2, 512, 65536, 4
    t_stream, t_res, flagged = ska_stream(torch, tricolour_amd, device, kw, nbl, ncorr, T, F, slabs, 1)
    nvis = nbl * ncorr * T * F * slabs
    leg = dict(value=round(nvis / t_res / 1e6, 2), unit="Mvis/s", ms_per_step=round(t_res / slabs * 1e3, 3), steps=slabs, warmup=1,
               workload="SKA-Mid window geometry (BASELINE configs[4], the 1-GPU half): slabs of %d bl x %d corr x %d x %d, "
                        "sum_threshold_flagger kwargs=%s; value = device-resident rate" % (nbl, ncorr, T, F, pname),
               pcie_inclusive=dict(value=round(nvis / t_stream / 1e6, 2), unit="Mvis/s", seconds=round(t_stream, 3),
                                   what="same slabs streamed host->device->host through two pinned buffers on three streams"),
               flagged_fraction=round(flagged, 4))
    flagging.release_workspace()
    torch.cuda.empty_cache()
    # one 512 x 65536 window at the full parameter set costs the oracle a minute per core: the parity sample of this
    # short leg takes ONE major iteration (all five background radii); the -m gpu suite holds the two-iteration case
    pkw = dict(kw, num_major_iterations=1)
    cb, hv, hf, hexp = cpu_baseline(pkw, T, F, fixed_windows=8)
    cb["sample"] += "; num_major_iterations=1 for this sample"
    leg["cpu_baseline"] = cb
    leg["parity_check"] = parity_check(torch, tricolour_amd, device, pkw, hv, hf, hexp)
    leg["parity_check"]["kwargs"] = "%s with num_major_iterations=1" % pname
    del hv, hf, hexp
    flagging.release_workspace()
    torch.cuda.empty_cache()
    legs["ska"] = leg
    return legs


def self_launch(args):
    """`bench.py --gpus N` outside a torchrun environment: start the N ranks as child processes (before any
    GPU call in this process) and relay rank 0's line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TRI_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["slab", "chain", "ska"], default="slab")
    ap.add_argument("--bl", type=int, default=None,
                    help="baselines per rank (slab: 252 of the 2016-baseline configuration; chain: 84 per scan; "
                         "ska: 32 per streamed slab)")
    ap.add_argument("--corr", type=int, default=None)
    ap.add_argument("--time", type=int, default=None)
    ap.add_argument("--chan", type=int, default=None)
    ap.add_argument("--scans", type=int, default=3, help="chain: scans per step")
    ap.add_argument("--ska-streams", type=int, default=2, choices=[1, 2],
                    help="ska: concurrent calls (kernel streams); 1 = one slab at a time as in round 3")
    ap.add_argument("--params", choices=sorted(PARAM_SETS), default=None)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: map every rank to cuda:0 (with --backend gloo)")
    ap.add_argument("--scatter-bl", type=int, default=8, help="baselines per rank in the N > 1 scatter / gather leg")
    ap.add_argument("--no-scatter", action="store_true")
    ap.add_argument("--scatter-timeout", type=int, default=240, help="seconds the N > 1 scatter / gather leg may take")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-other-params", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short chain / ska legs of the default (slab, N = 1) run")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the per-kernel roofline legs (what the rocprofv3 --pmc passes of profiles/ wrap)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: every rank joins a gloo group, rank 0 prints the rank count")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1,
                    help="launcher rehearsal: this rank leaves with exit code 3 the way a failed scatter leg does, while its "
                         "peers wait in a collective -- the launcher must tear the group down and report a non-zero code")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)                      # never returns
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with `python bench.py --gpus N` (self-launching) or "
                  "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    if args.dry_run:
        import torch
        import torch.distributed as dist
        n = 1
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
            if args.dry_run_fail_rank >= 0:
                if rank == args.dry_run_fail_rank:
                    if rank == 0:
                        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_joined": n, "scatter": {"error": "rehearsed failure"}}))
                        sys.stdout.flush()
                    os._exit(3)
                dist.barrier()                 # (never completes: the launcher ends this rank)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_joined": n}))
        return

    wl = args.workload
    ncorr = args.corr or (2 if wl == "ska" else 4)
    T = args.time or (512 if wl == "ska" else 1024)
    F = args.chan or (65536 if wl == "ska" else 4096)
    nbl = args.bl or int(os.environ.get("TRI_BENCH_BL", {"slab": "252", "chain": "84", "ska": "32"}[wl]))
    pname = args.params or "stage1"          # stage 1 is the headline parameter set
    steps = args.steps if args.steps is not None else {"slab": 3, "chain": 1, "ska": 64}[wl]
    warmup = args.warmup if args.warmup is not None else {"slab": 2, "chain": 1, "ska": 2}[wl]
    kw = PARAM_SETS[pname]

    import torch
    from tricolour_amd import _lib
    if world == 1 and _lib.needs_build():
        _lib.build()          # in-tree hipcc build (normally done by __graft_entry__.build())
    import tricolour_amd
    from tricolour_amd import flagging

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev_index = 0 if args.share_gpu else local_rank
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.roofline_only:
        nwin = min(nbl * ncorr, 1008 if wl != "ska" else 64)
        rk = [roofline_sumthreshold(torch, device, T, F, kw, nwin)] + roofline_boxfilter(torch, device, T, F, kw, nwin)
        rj = roofline_reject(torch, device, T, F, kw, nwin)
        print(json.dumps({"roofline_kernels": rk + ([rj] if rj else []), "lib_sha16": _lib.source_hash()}))
        return

    extra = {}
    nvis_step = nbl * ncorr * T * F
    if wl == "slab":
        vis, flags = synth_slab(torch, nbl, ncorr, T, F, device, 1234 + rank)
        step = lambda: tricolour_amd.sum_threshold_flagger(vis, flags, **kw)
        workload = ("MeerKAT-64 slab: %d of 2016 bl x %d corr x %d time x %d chan per GPU (BASELINE configs[1] "
                    "processed as HBM-resident baseline slabs), sum_threshold_flagger kwargs=%s" % (nbl, ncorr, T, F, pname)) \
            if (ncorr, T, F) == (4, 1024, 4096) else \
            ("custom window set: %d bl x %d corr x %d time x %d chan per GPU, sum_threshold_flagger kwargs=%s"
             % (nbl, ncorr, T, F, pname))
    elif wl == "chain":
        from tricolour_amd.strategies import apply_strategies
        scans = [synth_slab(torch, nbl, ncorr, T, F, device, 1234 + 17 * s + rank) for s in range(args.scans)]
        setup = chain_setup(nbl, F)
        strategies = chain_strategies(kw)

        def step():
            out = None
            for v, f in scans:
                out = apply_strategies(strategies, f, v, **setup)
            return out
        nvis_step *= args.scans
        workload = ("multi-scan chain (BASELINE configs[3]): %d scans x (%d bl x %d corr x %d time x %d chan) per GPU, "
                    "each flag_nans_zeros -> apply_static_mask(or) -> flag_autos -> uvcontsub_flagger -> sum_threshold "
                    "(kwargs=%s), device-resident" % (args.scans, nbl, ncorr, T, F, pname))
    else:
        workload = ("SKA-Mid window geometry (BASELINE configs[4]): slabs of %d bl x %d corr x %d time x %d chan "
                    "per GPU, sum_threshold_flagger kwargs=%s; value = device-resident rate over %d slabs"
                    % (nbl, ncorr, T, F, pname, steps))
    torch.cuda.synchronize()

    out = None
    if wl == "ska":
        barrier()
        t_stream, dt, flagged = ska_stream(torch, tricolour_amd, device, kw, nbl, ncorr, T, F, steps, warmup, args.ska_streams)
        barrier()
        extra["pcie_inclusive"] = dict(value=round(nvis_step * steps * world / t_stream / 1e6, 2), unit="Mvis/s",
                                       what="same slabs streamed host->device->host through two pinned buffers, "
                                            "H2D / kernels / D2H overlapped on three streams (10 B/vis over PCIe)",
                                       seconds=round(t_stream, 3))
    else:
        step_launches = {}
        for w in range(warmup):
            if rank == 0 and w == warmup - 1:
                _lib.kernel_log_begin()            # untimed: which kernels one step launches, and how often
            out = step()
            if rank == 0 and w == warmup - 1:
                step_launches = _lib.kernel_log_end()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        flagged = float(out.float().mean().item()) if out is not None else float("nan")
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_vis = nvis_step * world * steps
    value = total_vis / dt / 1e6

    res = None
    if rank == 0:
        res = {
            "metric": "Mvis/s flagged (bl x time x chan x corr)",
            "value": round(value, 2),
            "unit": "Mvis/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(dt / max(steps, 1) * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 data / f64 accumulators / u8 flags",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "baselines_per_gpu": nbl,
                "params": pname,
                "sharding": "baselines across ranks, no data-path collective",
                "flagged_fraction": round(flagged, 4),
            },
        }
        res.update(extra)
    # the same slab under the other shipped parameter sets (extra keys, N = 1, slab workload)
    if wl == "slab" and world == 1 and not args.no_other_params:
        others = {}
        for other in sorted(PARAM_SETS):
            if other == pname:
                continue
            okw = PARAM_SETS[other]
            tricolour_amd.sum_threshold_flagger(vis, flags, **okw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                tricolour_amd.sum_threshold_flagger(vis, flags, **okw)
            torch.cuda.synchronize()
            odt = (time.perf_counter() - t0) / 2
            others[other] = dict(value=round(nvis_step / odt / 1e6, 2), unit="Mvis/s", ms_per_step=round(odt * 1e3, 3), steps=2)
        res["other_params"] = others
    if wl == "slab":
        del vis, flags
    elif wl == "chain":
        del scans
    out = None
    flagging.release_workspace()
    torch.cuda.empty_cache()
    if rank == 0:
        nwin = min(nbl * ncorr, 1008 if wl != "ska" else 64)
        if not args.no_roofline:
            kernels = [roofline_sumthreshold(torch, device, T, F, kw, nwin)] + \
                roofline_boxfilter(torch, device, T, F, kw, nwin)
            rj = roofline_reject(torch, device, T, F, kw, nwin)
            if rj:
                kernels.append(rj)
            # "roofline" = ONE object: the kernel with the largest share of the timed step -- the box-filter
            # stage that runs at the most radii of the parameter set (the fused frequency-axis stage at its
            # largest radius when there are several background iterations, else the time-axis stage);
            # every measured kernel, the SumThreshold kernel among them, is under "roofline_kernels"
            if wl == "ska" or not step_launches:
                # (no per-step log on this path: count the launches of one untimed call)
                pv, pf = synth_slab(torch, min(nbl, 2), ncorr, T, F, device, 5)
                _lib.kernel_log_begin()
                tricolour_amd.sum_threshold_flagger(pv, pf, **kw)
                step_launches = _lib.kernel_log_end()
                del pv, pf
                flagging.release_workspace()
            dom, _ = pick_dominant(kernels, step_launches)
            if dom is not None:
                res["roofline"] = dict(dom)
            res["roofline_kernels"] = kernels
            res["lib_sha16"] = _lib.source_hash()
        if world == 1 and not args.no_cpu_baseline:
            chain_kw = UVCONTSUB_KW if wl == "chain" else None
            res["cpu_baseline"], hv, hf, hexp = cpu_baseline(kw, T, F, chain=chain_kw)
            if not args.no_parity_check:
                res["parity_check"] = parity_check(torch, tricolour_amd, device, kw, hv, hf, hexp, chain=chain_kw)
                flagging.release_workspace()
            del hv, hf, hexp
        torch.cuda.empty_cache()
        if wl == "slab" and world == 1 and not args.no_other_workloads and not args.no_cpu_baseline and (ncorr, T, F) == (4, 1024, 4096):
            try:
                res["other_workloads"] = other_workload_legs(torch, tricolour_amd, flagging, device, kw, pname)
            except Exception as e:            # the headline line must survive a failing extra leg -- but not silently
                res["other_workloads"] = dict(error="%s: %s" % (type(e).__name__, e))
        checks = [res.get("parity_check")] + [l.get("parity_check") for l in res.get("other_workloads", {}).values() if isinstance(l, dict)]
        checks = [c for c in checks if c]
        if checks:
            res["parity_ok"] = all(c.get("parity_ok", False) for c in checks) and "error" not in res.get("other_workloads", {})
    leg_failed = False
    # N > 1: the scatter -> flag -> gather leg comes last, under a watchdog -- a stuck point-to-point transfer
    # must not take the measured line with it
    if dist is not None and not args.no_scatter:
        import threading

        def give_up():
            if rank == 0:
                res["scatter"] = dict(error="no completion within %d s" % args.scatter_timeout)
                print(json.dumps(res))
                sys.stdout.flush()
            os._exit(3)                   # a stuck leg is a failure: the launcher and the driver must see it
        timer = threading.Timer(args.scatter_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            sc = scatter_leg(torch, dist, tricolour_amd, device, rank, world, kw, ncorr, T, F, args.scatter_bl, args.backend)
        except Exception as e:        # the headline line must survive a failing rehearsal leg
            sc = dict(error="%s: %s" % (type(e).__name__, e))
            leg_failed = True
        timer.cancel()
        if rank == 0:
            res["scatter"] = sc
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
        if res.get("parity_ok") is False and dist is None:
            sys.exit(4)                    # a parity failure must not look like a green run; the line is out
    if leg_failed:
        # peers may be stuck in the failed leg's transfers: no further collective; abort the group so they
        # fail fast instead of waiting for the collective timeout, and leave with a non-zero code
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        os._exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
