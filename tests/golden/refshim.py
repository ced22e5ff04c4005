"""Harness-side loader for the *un-jitted* reference (runs ONLY in the build
container, never on the GPU box; nothing from /root/reference is copied).

Follows SURVEY.md Appendix A: an identity-decorator ``numba`` stand-in (the
reference anticipates un-jitted use, flagging.py:213), a stub ``tricolour``
package object so that ``tricolour/__init__.py`` (donfig + package metadata)
is skipped, and ``_as_min_dtype`` widened to int64 (value preserving; needed
because un-jitted NumPy negates uint8 0-d arrays modulo 256).
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _install_numba_shim():
    if "numba" in sys.modules and not getattr(sys.modules["numba"], "_tri_shim", False):
        raise RuntimeError("a real numba is already imported")
    numba = types.ModuleType("numba")
    numba._tri_shim = True

    def _jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    numba.jit = numba.njit = _jit
    ext = types.ModuleType("numba.extending")
    ext.overload = lambda *a, **k: (lambda f: f)
    ext.register_jitable = _jit
    tps = types.ModuleType("numba.types")

    class Boolean:  # only referenced inside the (never executed) overload body
        pass

    class Integer:
        pass

    tps.Boolean, tps.Integer = Boolean, Integer
    numba.extending, numba.types = ext, tps
    sys.modules["numba"] = numba
    sys.modules["numba.extending"] = ext
    sys.modules["numba.types"] = tps


def load_reference_flagging():
    """Returns the reference ``tricolour.flagging`` module, un-jitted."""
    _install_numba_shim()
    if "tricolour" not in sys.modules:
        pkg = types.ModuleType("tricolour")
        pkg.__path__ = [REFERENCE_ROOT + "/tricolour"]
        sys.modules["tricolour"] = pkg
    sys.dont_write_bytecode = True
    import tricolour.flagging as fl  # noqa: E402
    fl._as_min_dtype = lambda v: np.array(v, np.int64)
    return fl
