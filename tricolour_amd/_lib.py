"""ctypes binding of the C ABI in ``include/tricolour_amd.h``.

The shared library is built in-tree (``tricolour_amd/libtricolour_amd.so``)
by :func:`build` with ``hipcc --offload-arch=gfx950``; it is the ONLY compute
path of this package -- there is no CPU or PyTorch fallback, and loading
fails loudly when the library is missing.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRICOLOUR_AMD_LIB: developer override, loads another build of the same sources
# (kernel A/B experiments, scripts/build_variants.sh)
LIB_PATH = os.environ.get("TRICOLOUR_AMD_LIB") or os.path.join(_HERE, "libtricolour_amd.so")
SOURCES = [os.path.join(_HERE, "csrc", "tricolour_amd.hip")]   # one translation unit
DEPENDS = [os.path.join(_HERE, "csrc", f) for f in (
    "tri_common.hpp", "kernels_elementwise.hpp", "kernels_median.hpp", "kernels_reject.hpp", "kernels_reject_tile.hpp", "kernels_boxfilter.hpp", "kernels_boxline.hpp", "kernels_boxpipe.hpp",
    "kernels_boxweight.hpp", "kernels_boxexact.hpp",
    "kernels_sumthreshold.hpp")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "tricolour_amd.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
               # IEEE evaluation order: no FMA contraction, no fast-math;
               # correctly rounded float32 division / sqrt
               "-ffp-contract=off", "-fno-fast-math",
               "-fhip-fp32-correctly-rounded-divide-sqrt",
               # the kernels are long unrolled blocks of independent dependency
               # chains: schedule for ILP rather than for occupancy (+6 % on the
               # fused SumThreshold kernel, neutral elsewhere)
               "-mllvm", "-amdgpu-sched-strategy=max-ilp"]

TRI_OK, TRI_EINVAL, TRI_EUNSUPPORTED, TRI_EWORKSPACE, TRI_EHIP = range(5)
TRI_VIS_C64, TRI_VIS_F32, TRI_VIS_C128, TRI_VIS_F64 = 0, 1, 2, 3
TRI_MAX_WINDOWS = 16


class TriParams(C.Structure):
    _fields_ = [
        ("outlier_nsigma", C.c_double),
        ("n_windows_time", C.c_int64),
        ("windows_time", C.c_int64 * TRI_MAX_WINDOWS),
        ("n_windows_freq", C.c_int64),
        ("windows_freq", C.c_int64 * TRI_MAX_WINDOWS),
        ("background_reject", C.c_double),
        ("background_iterations", C.c_int64),
        ("spike_width_time", C.c_double),
        ("spike_width_freq", C.c_double),
        ("time_extend", C.c_int64),
        ("freq_extend", C.c_int64),
        ("n_chunk_ends", C.c_int64),
        ("chunk_ends", C.POINTER(C.c_int64)),
        ("average_freq", C.c_int64),
        ("flag_all_time_frac", C.c_double),
        ("flag_all_freq_frac", C.c_double),
        ("rho", C.c_double),
        ("num_major_iterations", C.c_int64),
    ]


def source_hash():
    """sha256 (16 hex digits) of the kernel sources and build flags: identifies the build a measurement belongs to
    (bench.py ties the committed PMC traffic files to it)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in SOURCES + DEPENDS + [HEADER]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_log_begin():
    lib().tri_kernel_log(0, None, 0)


def kernel_log_end():
    """{kernel symbol: launches} of this thread since kernel_log_begin()."""
    buf = C.create_string_buffer(1 << 16)
    lib().tri_kernel_log(1, buf, len(buf))
    out = {}
    for item in buf.value.decode().split(";"):
        if "=" in item:
            k, v = item.rsplit("=", 1)
            out[k] = int(v)
    return out


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES + DEPENDS + [HEADER])


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950 (cross-compiles without a GPU)."""
    if not (force or needs_build()):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None

_SIGNATURES = {
    "tri_prepare_params": (C.c_int, [C.c_int64, C.c_int64, C.c_double,
                                     C.POINTER(C.c_double), C.c_int64,
                                     C.POINTER(C.c_double), C.c_int64,
                                     C.c_double, C.c_int64, C.c_double, C.c_double,
                                     C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_double, C.c_double, C.c_double, C.c_int64,
                                     C.POINTER(C.c_int64), C.c_int64,
                                     C.POINTER(TriParams)]),
    "tri_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64,
                                         C.POINTER(TriParams)]),
    "tri_sum_threshold_flagger": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_int64, C.c_int64,
                                            C.POINTER(TriParams), C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "tri_sum_threshold_flagger_debug": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_int64, C.c_int64,
                                                  C.c_int64, C.POINTER(TriParams),
                                                  C.c_void_p, C.c_size_t, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "tri_pack_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "tri_fill_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tri_unpack_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "tri_flag_nans_and_zeros": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tri_apply_baseline_channel_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "tri_stokes_intensity": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "tri_window_counts": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "tri_uvcontsub_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "tri_uvcontsub_flagger": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "tri_last_error": (C.c_char_p, []),
    "tri_version": (C.c_int, []),
    "tri_bench_sumthreshold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                         C.c_int64, C.c_double, C.c_double, C.c_int,
                                         C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "tri_bench_boxfilter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_float), C.c_void_p]),
    "tri_bench_reject": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_double, C.c_int,
                                   C.POINTER(C.c_float), C.c_void_p]),
    "tri_boxx_last_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tri_kernel_log": (C.c_int, [C.c_int, C.c_char_p, C.c_int64]),
    "tri_medrej_stats": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "tri_test_box_divide": (C.c_int, [C.c_int64, C.POINTER(C.c_uint64), C.c_void_p]),
    "tri_test_median": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_int, C.c_void_p]),
    "tri_abs_c64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """Loads the extension; raises if it is missing (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "tricolour_amd: HIP extension %s is missing -- run "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch owns the process's HIP runtime (its wheel bundles
        # libamdhip64.so.7).  Import it first so that the extension's
        # NEEDED libamdhip64.so.7 binds to that same runtime instance --
        # device pointers and streams are only meaningful within one.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc == TRI_OK:
        return
    msg = lib().tri_last_error().decode("utf-8", "replace")
    if rc == TRI_EINVAL:
        raise ValueError(msg)
    if rc == TRI_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == TRI_EWORKSPACE:
        raise MemoryError(msg)
    raise RuntimeError(msg)
