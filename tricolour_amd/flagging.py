"""Host-side mirror of ``tricolour.flagging.sum_threshold_flagger``.

Same name, keyword arguments, defaults, return value and error behaviour as
the reference (``tricolour/flagging.py:1076-1196``); the numerics run in the
HIP extension through the C ABI (``include/tricolour_amd.h``).  PyTorch is
only used to own device memory and the HIP stream.

* ``numpy`` in -> ``numpy`` out (H2D, run, D2H);
* ``torch`` ROCm tensors in -> ``torch.bool`` tensor out on the same device,
  zero-copy, enqueued on the current stream.

There is no CPU path: without a GPU (or without the built extension) the
call raises ``RuntimeError``.
"""
import ctypes as C
import collections
import os
import threading
import time

import numpy as np

from tricolour_amd import _lib

_tls = threading.local()
# host-array calls take turns on the PCIe link (TRICOLOUR_AMD_LINK_TURNS=0: free-for-all)
_H2D_TURN = threading.Lock()
_D2H_TURN = _H2D_TURN
_LINK_TURNS = os.environ.get("TRICOLOUR_AMD_LINK_TURNS", "1") != "0"
_TRACE = [] if os.environ.get("TRICOLOUR_AMD_TRACE") == "1" else None   # (debug) per-call phase times
# numpy blocks are cut into up to this many pieces along the baseline axis and pipelined inside the call
# (copy of piece i + 1 under the kernels of piece i); TRICOLOUR_AMD_PIPELINE=1 keeps a block in one piece
# (2 pieces measured best for one calling thread: 128 -> 102 ms per 16-baseline block; 4 pieces: 110 ms, the launches get
# too small.  When other threads have calls in flight their blocks already overlap each other's copies: no cutting then.)
_PIPELINE_PIECES = int(os.environ.get("TRICOLOUR_AMD_PIPELINE", "2") or 1)
_PIPELINE_MIN_WINDOWS = 16         # (bl, corr) windows per piece at least: smaller launches no longer fill the device
_INFLIGHT = [0]                    # numpy-block calls currently inside the library (all threads)
_INFLIGHT_LOCK = threading.Lock()
_LAST_CALLER = [None, 0.0]         # (thread id, time) of the latest numpy-block call
# TRICOLOUR_AMD_PINNED_RESULTS=1: numpy results are views of pinned host tensors from torch's caching host allocator --
# the flags then land in the result array at the link rate, without the 20 ms CPU copy into freshly mapped pages per
# 16-baseline block.  Opt-in: every result the caller keeps alive pins its memory.
_PINNED_RESULTS = os.environ.get("TRICOLOUR_AMD_PINNED_RESULTS", "0") == "1"


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    _lib.lib()  # fail loudly if the extension is missing
    if not torch.cuda.is_available():
        raise RuntimeError("tricolour_amd needs a ROCm GPU (MI355X / gfx950); "
                           "no device is visible and there is no CPU fallback")
    return torch


def prepare_params(ntime, nchan, outlier_nsigma=4.5,
                   windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                   background_reject=2.0, background_iterations=1,
                   spike_width_time=12.5, spike_width_freq=10.0,
                   time_extend=3, freq_extend=3,
                   freq_chunks=10, average_freq=1,
                   flag_all_time_frac=0.6, flag_all_freq_frac=0.8,
                   rho=1.3, num_major_iterations=5):
    """The plain-Python preparation of ``flagging.py:1156-1179`` (window
    de-duplication and clipping, frequency chunk ends), done by the library's
    ``tri_prepare_params``.  Returns a ``TriParams`` (keeps its chunk buffer
    alive) -- usable without a GPU."""
    lib = _lib.lib()
    wt = (C.c_double * max(len(windows_time), 1))(*[float(w) for w in windows_time])
    wf = (C.c_double * max(len(windows_freq), 1))(*[float(w) for w in windows_freq])
    freq_chunks = int(freq_chunks)
    buf = (C.c_int64 * (max(freq_chunks, 0) + 1))()
    p = _lib.TriParams()
    _lib.check(lib.tri_prepare_params(
        int(ntime), int(nchan), float(outlier_nsigma),
        wt, len(windows_time), wf, len(windows_freq),
        float(background_reject), int(background_iterations),
        float(spike_width_time), float(spike_width_freq),
        int(time_extend), int(freq_extend), freq_chunks, int(average_freq),
        float(flag_all_time_frac), float(flag_all_freq_frac), float(rho),
        int(num_major_iterations), buf, len(buf), C.byref(p)))
    p._chunk_buf = buf
    return p


_WS_CACHE_MAX = 3       # cached workspaces per thread and device (least recently used goes first)


def _workspace(torch, device, nbytes):
    """Per-thread, per-device, per-STREAM workspace tensor (grown on demand): calls from the
    threads of a dask ThreadPool never share scratch memory, and neither do two pipelines
    that one thread has in flight on different streams.  The cache is a small LRU: a thread
    that rotates through many streams keeps at most _WS_CACHE_MAX workspaces per device alive
    (a stream handle can also be recycled by the runtime after its stream is destroyed, so an
    old entry must not live forever); before a workspace GROWS, the thread's other workspaces
    on that device are dropped -- the new one was sized against the memory they hold.  Callers
    that create and destroy streams freely should call release_workspace() when done."""
    cache = getattr(_tls, "ws", None)
    if cache is None:
        cache = _tls.ws = collections.OrderedDict()
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = cache.get(key)
    if t is not None and t.numel() >= nbytes:
        cache.move_to_end(key)
        return t
    if t is not None:
        # this stream's workspace must GROW: it was sized against the memory the thread's other workspaces on the
        # device hold, so those go first (a miss for a new stream leaves them alone -- two pipelined pieces on two
        # side streams keep one workspace each)
        for k in [k for k in cache if k[:2] == key[:2]]:
            del cache[k]
        t = None
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    cache[key] = t
    cache.move_to_end(key)
    mine = [k for k in cache if k[:2] == key[:2]]
    for k in mine[:-_WS_CACHE_MAX]:
        del cache[k]
    return t


def release_workspace():
    """Drops this thread's cached device workspaces."""
    _tls.ws = collections.OrderedDict()


def set_num_threads(n):
    """Declares how many host threads will call into the library concurrently (dask's
    ThreadPool size): every thread then sizes its workspace against 1/n of the device
    instead of against whatever the other threads have left free.  Also settable through
    the environment variable TRICOLOUR_AMD_THREADS."""
    global _declared_threads
    _declared_threads = max(1, int(n))


_declared_threads = None


def _workspace_budget(torch, device, share=1.0):
    """Bytes a call may take for its workspace.  `share`: fraction of the device's free memory this call may claim
    (the in-call pipeline runs two pieces on two streams: each budgets half)."""
    env = os.environ.get("TRICOLOUR_AMD_WORKSPACE_GB")
    if env:
        return int(float(env) * (1 << 30) * share)
    free, total = torch.cuda.mem_get_info(device)
    # what this thread already holds FOR THIS STREAM can be reused or released: it counts as available.  The workspaces
    # of its other streams do not -- they may be in flight, and their blocks return to another stream's pool
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    mine = getattr(_tls, "ws", {}).get(key)
    have = mine.numel() if mine is not None else 0
    nthreads = _declared_threads or int(os.environ.get("TRICOLOUR_AMD_THREADS", "0") or 0)
    if nthreads > 1:
        # a fixed share of the device -- but never more than what is actually free right now (the caller's own
        # slabs, other processes): the batch shrinks instead of the allocation failing
        return max(min(int(0.6 * total / nthreads * share), int(0.9 * free * share) + have), 0)
    return int(0.6 * (free * share + have))


def _pick_batch(lib, p, n_cp, T, F, budget):
    one = lib.tri_workspace_bytes(1, T, F, C.byref(p))
    if one == 0:
        # let the flagger produce the precise error message
        return 1, 256
    if one > budget:
        raise MemoryError("tricolour_amd: one (%d x %d) window needs %.1f MiB of "
                          "workspace, more than the %.1f MiB budget"
                          % (T, F, one / 2**20, budget / 2**20))
    lo, hi = 1, max(1, min(n_cp, 4096))
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if lib.tri_workspace_bytes(mid, T, F, C.byref(p)) <= budget:
            lo = mid
        else:
            hi = mid - 1
    return lo, lib.tri_workspace_bytes(lo, T, F, C.byref(p))


def _host_call_stream(torch, vis, flags):
    """Calls that hand over HOST arrays (the dask graph does) run on a stream of
    their own, one per calling thread: the H2D copies, kernels and the D2H copy
    of blocks processed by different threads then overlap instead of queueing
    on the default stream.  Device tensors stay on the caller's current stream."""
    for a in (vis, flags):
        if torch.is_tensor(a) and a.is_cuda:
            return None
    device = torch.device("cuda", torch.cuda.current_device())
    streams = getattr(_tls, "streams", None)
    if streams is None:
        streams = _tls.streams = {}
    key = device.index
    if key not in streams:
        streams[key] = torch.cuda.Stream(device)
    return streams[key]


_D2H_STAGE_MAX = 1 << 30      # larger results (blocks of > 64 baselines of 1024 x 4096) go straight to their array


def _d2h_stage(torch, nbytes):
    """This thread's pinned staging buffer for results (grown on demand, reused by every later call);
    None when the result is too large to keep pinned memory of its size around, or pinning fails."""
    if nbytes > _D2H_STAGE_MAX:
        return None
    st = getattr(_tls, "d2h_stage", None)
    if st is None or st.numel() < nbytes:
        _tls.d2h_stage = None
        try:
            st = _tls.d2h_stage = torch.empty(int(nbytes), dtype=torch.uint8).pin_memory()
        except RuntimeError:
            return None
    return st


def _as_device_inputs(torch, vis, flags):
    """Returns (vis_tensor, flags_u8_tensor, vis_dtype_code, from_numpy, device)."""
    from_numpy = isinstance(vis, np.ndarray) or isinstance(flags, np.ndarray)
    device = None
    if torch.is_tensor(vis) and vis.is_cuda:
        device = vis.device
    elif torch.is_tensor(flags) and flags.is_cuda:
        device = flags.device
    else:
        device = torch.device("cuda", torch.cuda.current_device())

    def to_t(a):
        if isinstance(a, np.ndarray):
            # pageable host memory: the driver's own staged copy runs at the link rate on this platform
            # (56 GB/s measured, scripts/host_copy_rates.py) -- faster than copying into pinned buffers first
            return torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)
        if torch.is_tensor(a):
            return a.to(device)
        return torch.as_tensor(np.asarray(a)).to(device)

    if from_numpy and _LINK_TURNS:
        # One block on the link at a time.  A single copy already saturates it, so queueing the copies
        # loses nothing -- but it staggers the calling threads: while one block's kernels run, the next
        # thread's block is on the link, instead of all threads copying together and then computing
        # together (measured: 2.2 -> see profiles/r02_host_path.txt).
        t0 = time.time()
        with _H2D_TURN:
            t1 = time.time()
            v = to_t(vis)
            f = to_t(flags)
            torch.cuda.current_stream(device).synchronize()
        if _TRACE is not None:
            _TRACE.append((threading.get_ident(), "h2d", t0, t1, time.time()))
    else:
        v = to_t(vis)
        f = to_t(flags)
    if v.dtype == torch.complex64:
        code = _lib.TRI_VIS_C64
    elif v.dtype == torch.float32:
        code = _lib.TRI_VIS_F32
    elif v.dtype == torch.float64:
        # |x| of a float64 enters the reference's float32 accumulator through
        # one round-to-nearest (flagging.py:856-859), which is what the cast
        # does when no channel averaging follows
        code = -64
    elif v.dtype == torch.complex128:
        code = -128
    else:
        raise TypeError("tricolour_amd.sum_threshold_flagger: visibilities must be real or "
                        "complex floating point (got %s)" % v.dtype)
    if f.dtype == torch.bool:
        f8 = f.contiguous().view(torch.uint8)
    elif f.dtype in (torch.uint8, torch.int8):
        f8 = f.contiguous().view(torch.uint8)
    else:
        f8 = (f != 0).view(torch.uint8)   # flagging.py:833-835: non-zero = flagged
    return v.contiguous(), f8, code, from_numpy, device


def sum_threshold_flagger(vis, flags, outlier_nsigma=4.5,
                          windows_time=[1, 2, 4, 8], windows_freq=[1, 2, 4, 8],
                          background_reject=2.0, background_iterations=1,
                          spike_width_time=12.5, spike_width_freq=10.0,
                          time_extend=3, freq_extend=3,
                          freq_chunks=10, average_freq=1,
                          flag_all_time_frac=0.6, flag_all_freq_frac=0.8,
                          rho=1.3, num_major_iterations=5, _debug=None):
    """
    Flagger that uses the SumThreshold method (Offringa, A., MNRAS, 405,
    155-167, 2010) to detect spikes in both frequency and time axes --
    MI355X implementation of ``tricolour.flagging.sum_threshold_flagger``
    (reference ``tricolour/flagging.py:1076-1196``), same arguments:

    vis : (bl, corr, time, chan) complex64 (or float32 amplitudes)
    flags : same shape, bool (any integer type: non-zero = flagged)

    Returns the flags found by the LAST major iteration (not OR-ed with the
    input flags), bool, same 4-D shape; inputs are never modified.
    """
    lib = _lib.lib()
    if tuple(vis.shape) != tuple(flags.shape):
        raise ValueError("shape mismatch")           # flagging.py:840-841
    if len(vis.shape) != 4:
        raise ValueError("not enough values to unpack (expected 4, got %d)"
                         % len(vis.shape))            # flagging.py:1156
    nbl, ncorr, ntime, nchan = (int(s) for s in vis.shape)
    p = prepare_params(ntime, nchan, outlier_nsigma, windows_time, windows_freq,
                       background_reject, background_iterations, spike_width_time,
                       spike_width_freq, time_extend, freq_extend, freq_chunks,
                       average_freq, flag_all_time_frac, flag_all_freq_frac, rho,
                       num_major_iterations)
    torch = _require_gpu()
    if _debug is None and isinstance(vis, np.ndarray) and isinstance(flags, np.ndarray):
        me, now = threading.get_ident(), time.time()
        with _INFLIGHT_LOCK:
            _INFLIGHT[0] += 1
            # alone: nobody else inside, and no other thread was here within the last two seconds (a pool of
            # threads taking blocks one after the other overlaps their copies by itself)
            alone = _INFLIGHT[0] == 1 and (_LAST_CALLER[0] in (None, me) or now - _LAST_CALLER[1] > 2.0)
            _LAST_CALLER[0], _LAST_CALLER[1] = me, now
        try:
            if (alone and _PIPELINE_PIECES > 1 and nbl >= 2 and nbl * ncorr >= 2 * _PIPELINE_MIN_WINDOWS
                    and ntime > 0 and nchan > 0):
                return _flag_numpy_pipelined(torch, lib, p, vis, flags, average_freq)
            return _flag_host_call(torch, lib, p, vis, flags, average_freq, _debug)
        finally:
            with _INFLIGHT_LOCK:
                _INFLIGHT[0] -= 1
                _LAST_CALLER[1] = time.time()
    return _flag_host_call(torch, lib, p, vis, flags, average_freq, _debug)


def _flag_host_call(torch, lib, p, vis, flags, average_freq, _debug):
    side = _host_call_stream(torch, vis, flags)
    if side is not None:
        caller = torch.cuda.current_stream(side.device)
        side.wait_stream(caller)          # anything the caller queued (e.g. a CPU tensor being produced) comes first
        with torch.cuda.stream(side):
            out = _flag_on_current_stream(torch, lib, p, vis, flags, average_freq, _debug)
        if torch.is_tensor(out) and out.is_cuda:
            # host inputs that were not numpy (CPU tensors, lists) get a device tensor back: hand it
            # over to the caller's stream properly
            caller.wait_stream(side)
            out.record_stream(caller)
        return out
    return _flag_on_current_stream(torch, lib, p, vis, flags, average_freq, _debug)


def _flag_numpy_pipelined(torch, lib, p, vis, flags, average_freq):
    """One numpy block, pipelined INSIDE the call (the dask graph hands over one block per call and thread,
    dask_wrappers.py:23-46 of the reference): the block is cut into pieces along the baseline axis, piece i + 1 is
    on the PCIe link while piece i's kernels run (two per-thread side streams, each with its own workspace),
    and piece i's flags come back -- and are copied into the result array by the CPU -- under piece i + 1's
    kernels.  Windows are independent (flagging.py:765-774), so the pieces' results are the block's results."""
    nbl, ncorr, ntime, nchan = (int(x) for x in vis.shape)
    device = torch.device("cuda", torch.cuda.current_device())
    pieces = max(2, min(_PIPELINE_PIECES, nbl, (nbl * ncorr) // _PIPELINE_MIN_WINDOWS))
    bounds = [nbl * k // pieces for k in range(pieces + 1)]
    streams = getattr(_tls, "pipe_streams", None)
    if streams is None:
        streams = _tls.pipe_streams = {}
    if device.index not in streams:
        streams[device.index] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    pair = streams[device.index]
    caller = torch.cuda.current_stream(device)
    res = _new_result(torch, (nbl, ncorr, ntime, nchan))
    pending = None

    def drain(item):
        out, st, b0, b1 = item
        with torch.cuda.stream(st):
            _result_to_numpy(torch, out, device, res[b0:b1])
    for k in range(pieces):
        b0, b1 = bounds[k], bounds[k + 1]
        if b1 <= b0:
            continue
        st = pair[k & 1]
        st.wait_stream(caller)
        with torch.cuda.stream(st):
            out, _ = _flag_device(torch, lib, p, vis[b0:b1], flags[b0:b1], average_freq, None, share=0.5)
        if pending is not None:
            drain(pending)          # (its kernels were queued before this piece's copy: they ran under it)
        pending = (out, st, b0, b1)
    if pending is not None:
        drain(pending)
    return res


def _flag_on_current_stream(torch, lib, p, vis, flags, average_freq, _debug):
    out, from_numpy = _flag_device(torch, lib, p, vis, flags, average_freq, _debug)
    if from_numpy:
        return _result_to_numpy(torch, out, out.device)
    return out


def _flag_device(torch, lib, p, vis, flags, average_freq, _debug, share=1.0):
    """Inputs to the device (if they are not there), the flagger on the current stream; returns the device
    bool tensor and whether the caller handed over numpy arrays."""
    nbl, ncorr, ntime, nchan = (int(s) for s in vis.shape)
    v, f8, code, from_numpy, device = _as_device_inputs(torch, vis, flags)
    nan_part = None
    if code in (-64, -128):
        # float64 / complex128 input (the reference accepts "real or complex", flagging.py:830-835):
        # its amplitude is np.abs in float64 -- for complex128 numba's hypot(re, im) -- and enters the
        # float32 accumulator `value` through one round-to-nearest (flagging.py:856-859).  Without
        # channel averaging that rounded amplitude IS the averaged sample, so it can be formed up front.
        # With channel averaging the float64 amplitude itself enters the accumulator -- the sum is formed in float64 and
        # rounded to float32 at every step (flagging.py:858-859) -- so it goes to the library as float64 (TRI_VIS_F64); a
        # visibility with a NaN part arrives as NaN (the final isnan(in_data), flagging.py:777-781).
        wide = int(average_freq) != 1
        if code == -128:
            re, im = v.real, v.imag
            # torch.hypot is within 1 ulp of float64 of libm's hypot: far inside the float32 rounding that follows;
            # (inf, nan) -> inf like C99 hypot
            amp = torch.hypot(re, im)
            amp = torch.where(torch.isinf(re) | torch.isinf(im), torch.full_like(amp, float("inf")), amp)
            # A NaN part always flags the sample in the end (flagging.py:777-781), also next to an infinite one -- but the
            # AMPLITUDE of (inf, nan) is +inf (np.abs = hypot), and that is what _average_freq accumulates while the sample
            # is unflagged (flagging.py:856-861).  So the amplitude stays +inf (as on the complex64 route: k_amplitude4
            # keeps amplitude and NaN bit apart) and the NaN marks are OR-ed into the result below.
            nan_part = torch.isnan(re) | torch.isnan(im)
            if not bool((nan_part & torch.isinf(amp)).any()):
                nan_part = None                       # every NaN part already shows as a NaN amplitude
        else:
            amp = v.abs()
        if wide:
            v, code = amp.contiguous(), _lib.TRI_VIS_F64
        else:
            v, code = amp.to(torch.float32), _lib.TRI_VIS_F32
    n_cp = nbl * ncorr
    with torch.cuda.device(device):
        out = torch.empty((nbl, ncorr, ntime, nchan), dtype=torch.uint8, device=device)
        if n_cp > 0 and ntime > 0 and nchan > 0:
            budget = _workspace_budget(torch, device, share)
            _batch, nbytes = _pick_batch(lib, p, n_cp, ntime, nchan, budget)
            ws = _workspace(torch, device, nbytes)
            stream = torch.cuda.current_stream(device).cuda_stream
            if _debug is None:
                rc = lib.tri_sum_threshold_flagger(v.data_ptr(), code, f8.data_ptr(),
                                                   out.data_ptr(), n_cp, ntime, nchan,
                                                   C.byref(p), ws.data_ptr(), ws.numel(),
                                                   stream)
            else:
                fa = (nchan + int(average_freq) - 1) // int(average_freq)
                n = ntime * fa
                dbg_f = torch.zeros(fa + 2 * n, dtype=torch.float32, device=device)
                dbg_u = torch.zeros(fa + 2 * n, dtype=torch.uint8, device=device)
                rc = lib.tri_sum_threshold_flagger_debug(
                    v.data_ptr(), code, f8.data_ptr(), out.data_ptr(), n_cp, ntime,
                    nchan, C.byref(p), ws.data_ptr(), ws.numel(), stream,
                    dbg_f.data_ptr(), dbg_u.data_ptr())
                if rc == 0:
                    torch.cuda.synchronize(device)
                    df, du = dbg_f.cpu().numpy(), dbg_u.cpu().numpy()
                    _debug.update(
                        spec_resid=df[:fa].copy(),
                        background=df[fa:fa + n].reshape(fa, ntime).T.copy(),
                        residual=df[fa + n:].reshape(ntime, fa).copy(),
                        spec_flags=du[:fa].astype(bool),
                        time_flags=du[fa:fa + n].reshape(ntime, fa).astype(bool),
                        freq_flags=du[fa + n:].reshape(ntime, fa).astype(bool))
            _lib.check(rc)
            if nan_part is not None and int(p.num_major_iterations) > 0:
                out |= nan_part.view(torch.uint8)     # flagging.py:777-781 for (inf, nan) samples of complex128 input
        elif ntime <= 0 or nchan <= 0:
            _lib.check(lib.tri_sum_threshold_flagger(
                v.data_ptr(), code, f8.data_ptr(), out.data_ptr(), n_cp, ntime, nchan,
                C.byref(p), None, 0, None))
    return out.view(torch.bool), from_numpy


def _new_result(torch, shape):
    """The numpy bool array a host call returns: ordinary memory, or (TRICOLOUR_AMD_PINNED_RESULTS=1) a view of a pinned
    tensor from torch's caching host allocator (the array keeps the tensor alive; its memory returns to the pool with it)."""
    if _PINNED_RESULTS:
        try:
            return torch.empty(shape, dtype=torch.bool, pin_memory=True).numpy()
        except RuntimeError:
            pass
    return np.empty(shape, np.bool_)


def _dest_is_pinned(torch, dest):
    try:
        return _PINNED_RESULTS and dest is not None and torch.from_numpy(dest).is_pinned()
    except Exception:
        return False


def _result_to_numpy(torch, out, device, dest=None):
    """Device flags of a numpy call -> a fresh numpy bool array, or the C-contiguous array `dest` (on the current stream)."""
    shape = tuple(int(x) for x in out.shape)
    if _PINNED_RESULTS:
        if dest is None:
            dest = _new_result(torch, shape)
        if _dest_is_pinned(torch, dest):
            torch.cuda.current_stream(device).synchronize()
            with _D2H_TURN:
                torch.from_numpy(dest).copy_(out, non_blocking=True)
                torch.cuda.current_stream(device).synchronize()
            return dest
    if _LINK_TURNS:
        t0 = time.time()
        torch.cuda.current_stream(device).synchronize()     # kernels done before queueing for the link
        t1 = time.time()
        # The result goes to a fresh numpy array, whose pages the kernel has to zero and map first:
        # copied into directly, that happens page by page INSIDE the device-to-host copy (5-10 GB/s, and
        # it drags down a concurrent host-to-device copy of another thread).  So: device -> this thread's
        # pinned staging buffer at the link rate (5 ms for a 16-baseline block), link released, then a
        # plain CPU copy into the fresh array.
        stage = _d2h_stage(torch, out.numel())
        if stage is None:
            with _D2H_TURN:
                got = out.cpu().numpy()
            if dest is None:
                return got
            np.copyto(dest, got)
            return dest
        with _D2H_TURN:
            t2 = time.time()
            stage[:out.numel()].copy_(out.view(torch.uint8).reshape(-1), non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
        t3 = time.time()
        res = np.empty(shape, np.bool_) if dest is None else dest
        np.copyto(res.reshape(-1).view(np.uint8), stage[:out.numel()].numpy())
        if _TRACE is not None:
            _TRACE.append((threading.get_ident(), "kernels+d2h", t0, t1, t2, t3, time.time()))
        return res
    got = out.cpu().numpy()
    if dest is None:
        return got
    np.copyto(dest, got)
    return dest


# ---------------------------------------------------------------------------
# The cheap strategy steps around sum_threshold, same
# signatures as the reference so a default.yaml chain can stay on the device.
# ---------------------------------------------------------------------------
def _flags_u8(torch, f, device):
    if isinstance(f, np.ndarray):
        f = torch.from_numpy(np.ascontiguousarray(f))
    f = f.to(device)
    if f.dtype in (torch.bool, torch.uint8, torch.int8):
        return f.contiguous().view(torch.uint8)
    return (f != 0).view(torch.uint8)


def flag_nans_and_zeros(vis_windows, flag_windows):
    """Flag nan and zero visibilities -- ``tricolour.flagging.flag_nans_and_zeros``
    (flagging.py:29-62): ``(vis == 0) | isnan(vis) | (flags != 0)``; output has
    the dtype of ``flag_windows`` (np.zeros_like, flagging.py:51)."""
    if tuple(vis_windows.shape) != tuple(flag_windows.shape):
        raise ValueError("vis_windows.shape != flag_windows.shape")   # flagging.py:46-47
    torch = _require_gpu()
    v, f8, code, from_numpy, device = _as_device_inputs(torch, vis_windows, flag_windows)
    if code == -64:
        raise TypeError("tricolour_amd.flag_nans_and_zeros: visibilities must be complex64 or float32")
    f8 = _flags_u8(torch, flag_windows, device)
    out = torch.empty(f8.shape, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().tri_flag_nans_and_zeros(
            v.data_ptr(), code, f8.data_ptr(), out.data_ptr(), out.numel(),
            torch.cuda.current_stream(device).cuda_stream))
    return _like_flags(torch, out, flag_windows, from_numpy)


def _like_flags(torch, out_u8, like, from_numpy):
    """uint8 0/1 result -> the container / dtype of the reference's output."""
    if from_numpy:
        o = out_u8.cpu().numpy()
        dt = like.dtype if isinstance(like, np.ndarray) else np.bool_
        return o.view(np.bool_) if dt == np.bool_ else o.astype(dt)
    if torch.is_tensor(like) and like.dtype != torch.bool:
        return out_u8.to(like.dtype)
    return out_u8.view(torch.bool)


def _apply_bl_chan(torch, flags, bl_sel, chan_masks, mode):
    from_numpy = isinstance(flags, np.ndarray)
    device = flags.device if (torch.is_tensor(flags) and flags.is_cuda) else \
        torch.device("cuda", torch.cuda.current_device())
    f8 = _flags_u8(torch, flags, device)
    nbl, ncorr, ntime, nchan = (int(x) for x in f8.shape)
    out = f8.clone()                                   # flagging.py:90, 151: flags.copy()
    sel = torch.from_numpy(np.ascontiguousarray(bl_sel, np.uint8)).to(device)
    lib = _lib.lib()
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        for cm in chan_masks:
            m = torch.from_numpy(np.ascontiguousarray(cm, np.uint8)).to(device)
            _lib.check(lib.tri_apply_baseline_channel_mask(
                out.data_ptr(), out.data_ptr(), sel.data_ptr(), m.data_ptr(), mode,
                nbl, ncorr, ntime, nchan, stream))
    return _like_flags(torch, out, flags, from_numpy)


def flag_autos(flags, ubl):
    """Flags auto-correlations -- ``tricolour.flagging.flag_autos``
    (flagging.py:65-95).  ``ubl`` arrives wrapped in a list, as dask's
    ``bl-comp`` contraction delivers it (flagging.py:84)."""
    ubl = np.asarray(ubl[0])
    if flags.shape[0] != ubl.shape[0]:
        raise ValueError("flag and ubl shape mismatch %s != %s" % (flags.shape[2], ubl.shape[0]))
    torch = _require_gpu()
    bl_sel = ubl[:, 1] == ubl[:, 2]
    return _apply_bl_chan(torch, flags, bl_sel, [np.ones(int(flags.shape[3]), np.uint8)], 0)


def static_mask_selection(ubl, antspos, masks, chan_freqs, chan_widths, uvrange=""):
    """Host half of ``apply_static_mask`` (flagging.py:131-160): the baseline
    selection from the uv-range and one boolean channel mask per static mask."""
    from tricolour_amd.util import casa_style_range
    uvrange = casa_style_range(uvrange)
    ubl = np.asarray(ubl)
    antspos = np.asarray(antspos)
    spw_chanlb = chan_freqs - chan_widths * 0.5
    spw_chanub = chan_freqs + chan_widths * 0.5
    bl_length = antspos[ubl[:, 1]] - antspos[ubl[:, 2]]
    d2 = 0.5 * np.sum(bl_length**2, axis=1)
    luvrange = 0.0 if uvrange is None else min(uvrange[0], uvrange[1])
    uuvrange = np.inf if uvrange is None else max(uvrange[0], uvrange[1])
    bl_sel = np.logical_and(d2 >= luvrange**2, d2 <= uuvrange**2)
    chan_masks = []
    for mask in masks:
        mask = np.asarray(mask)
        if mask.ndim != 2 and mask.shape[1] != 1:
            raise ValueError("masks.shape != (dim, 1)")
        lower_mask = mask[:, :] >= spw_chanlb[None, :]
        upper_mask = mask[:, :] < spw_chanub[None, :]
        chan_masks.append(np.logical_and(lower_mask, upper_mask).sum(axis=0) > 0)
    return bl_sel, chan_masks


def apply_static_mask(flag, ubl, antspos, masks, chan_freqs, chan_widths,
                      accumulation_mode="or", uvrange=""):
    """Applies static masks -- ``tricolour.flagging.apply_static_mask``
    (flagging.py:98-172): channels whose band contains a masked frequency are
    flagged ("or") or replace the flags ("override") on the baselines inside
    ``uvrange``."""
    ubl = np.asarray(ubl)
    if flag.shape[0] != ubl.shape[0]:
        raise ValueError("flag and ubl shape mismatch %s != %s" % (flag.shape[1], ubl.shape[0]))
    if accumulation_mode not in ("or", "override"):
        if len(masks) > 0:
            raise ValueError("Invalid accumulation_mode '%s'. Should be 'or' or 'override'"
                             % accumulation_mode)
    bl_sel, chan_masks = static_mask_selection(ubl, antspos, masks, chan_freqs, chan_widths, uvrange)
    torch = _require_gpu()
    return _apply_bl_chan(torch, flag, bl_sel, chan_masks, 0 if accumulation_mode == "or" else 1)


def uvcontsub_flagger(vis, flags, major_cycles=5, or_original_from_cycle=1,
                      taylor_degrees=20, sigma=5):
    """Iteratively fits a low-order Fourier model to the time-averaged
    spectrum, subtracts it and clips at ``sigma`` x MAD-of-MAD --
    ``tricolour.flagging.uvcontsub_flagger`` (flagging.py:989-1073), same
    arguments.  The reference is plain NumPy (float32 FFT / residuals under
    NumPy >= 2, float64 before); this follows the float32 semantics, so flags
    agree with it to within threshold-boundary cases, not bit for bit."""
    if tuple(vis.shape) != tuple(flags.shape):
        raise ValueError("vis and flags must have the same shape")       # flagging.py:1018-1019
    torch = _require_gpu()
    v, f8, code, from_numpy, device = _as_device_inputs(torch, vis, flags)
    if code != _lib.TRI_VIS_C64:
        raise TypeError("tricolour_amd.uvcontsub_flagger: visibilities must be complex64")
    f8 = _flags_u8(torch, flags, device)
    nbl, ncorr, ntime, nchan = (int(s) for s in v.shape)
    lib = _lib.lib()
    out = torch.empty(f8.shape, dtype=torch.uint8, device=device)
    n_cp = nbl * ncorr
    with torch.cuda.device(device):
        if n_cp > 0 and ntime > 0 and nchan > 0:
            budget = _workspace_budget(torch, device)
            batch = max(1, min(n_cp, 4096))
            while batch > 1 and lib.tri_uvcontsub_workspace_bytes(batch, ntime, nchan) > budget:
                batch = (batch + 1) // 2
            nbytes = lib.tri_uvcontsub_workspace_bytes(batch, ntime, nchan)
            ws = _workspace(torch, device, nbytes)
            _lib.check(lib.tri_uvcontsub_flagger(
                v.data_ptr(), f8.data_ptr(), out.data_ptr(), n_cp, ntime, nchan,
                int(major_cycles), int(or_original_from_cycle), int(taylor_degrees),
                float(sigma), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream(device).cuda_stream))
    return _like_flags(torch, out, flags, from_numpy)
