"""Stokes-derived intensities: drop-in for ``tricolour.stokes`` (reference
``tricolour/stokes.py``), the input transform of the ``polarisation`` and
``total_power`` flagging strategies (``apps/tricolour/app.py:413-432``).

``stokes_corr_map`` is host logic; ``polarised_intensity`` and
``unpolarised_intensity`` run one elementwise HIP kernel (``tri_stokes_intensity``)
with numba's typing of the reference loops: terms in complex128, float64 sums,
result cast back to the visibility dtype.
"""
import ctypes as C

import numpy as np

from tricolour_amd import _lib

# Correlation / Stokes enumeration of Measurement Set 2.0 (casacore Stokes.h)
STOKES_TYPES = dict(I=1, Q=2, U=3, V=4, RR=5, RL=6, LR=7, LL=8, XX=9, XY=10, YX=11, YY=12)

# stokes = alpha * (s1 * corr1 + s2 * corr2); linear feeds first, circular feeds
# second, so that circular wins when a dataset somehow carries both
# (stokes.py:29-34, 64-72: the last applicable rule is kept)
_RULES = (
    ("I", "XX", "YY", 0.5 + 0.0j, 1, 1), ("I", "RR", "LL", 0.5 + 0.0j, 1, 1),
    ("Q", "XX", "YY", 0.5 + 0.0j, 1, -1), ("Q", "RL", "LR", 0.5 + 0.0j, 1, 1),
    ("U", "XY", "YX", 0.5 + 0.0j, 1, 1), ("U", "RL", "LR", 0.0 - 0.5j, 1, -1),
    ("V", "XY", "YX", 0.0 - 0.5j, 1, -1), ("V", "RR", "LL", 0.5 + 0.0j, 1, -1),
)


def stokes_corr_map(corr_types):
    """``{stokes: (c1, c2, a, s1, s2)}`` for the Stokes parameters that the
    correlations ``corr_types`` (casacore codes, in dataset order) can form:
    ``stokes = a * (s1 * vis[:, :, c1] + s2 * vis[:, :, c2])``."""
    corr_types = list(corr_types)
    position = {code: corr_types.index(code) for code in set(corr_types)}
    found = {}
    for stokes, first, second, alpha, s1, s2 in _RULES:
        k1, k2 = STOKES_TYPES[first], STOKES_TYPES[second]
        if k1 in position and k2 in position:
            found[stokes] = (position[k1], position[k2], alpha, s1, s2)
    # dictionary order of the reference: I, Q, U, V
    return {s: found[s] for s in "IQUV" if s in found}


def _term_tables(terms):
    terms = tuple(terms)
    idx = np.zeros((max(len(terms), 1), 4), np.int32)
    alpha = np.zeros((max(len(terms), 1), 2), np.float64)
    for k, (c1, c2, a, s1, s2) in enumerate(terms):
        idx[k] = (c1, c2, s1, s2)
        alpha[k] = (complex(a).real, complex(a).imag)
    return idx, alpha, len(terms)


def _intensity(vis, stokes_pol, stokes_unpol, mode):
    import torch
    from_numpy = not torch.is_tensor(vis)
    v = torch.from_numpy(np.ascontiguousarray(vis)) if from_numpy else vis
    if v.dim() != 3:
        raise ValueError("vis must have shape (row, chan, corr)")
    if v.dtype == torch.complex64:
        code = _lib.TRI_VIS_C64
    elif v.dtype == torch.complex128:
        code = _lib.TRI_VIS_C128
    else:
        raise TypeError("tricolour_amd.stokes: visibilities must be complex64 or complex128 (got %s)" % v.dtype)
    if not v.is_cuda:
        v = v.cuda()
    v = v.contiguous()
    nrow, nchan, ncorr = (int(s) for s in v.shape)
    out = torch.empty((nrow, nchan, 1), dtype=v.dtype, device=v.device)
    pidx, palpha, npol = _term_tables(stokes_pol)
    uidx, ualpha, nunpol = _term_tables(stokes_unpol)
    stream = torch.cuda.current_stream(v.device).cuda_stream
    _lib.check(_lib.lib().tri_stokes_intensity(
        v.data_ptr(), code, nrow * nchan, ncorr,
        pidx.ctypes.data_as(C.c_void_p), palpha.ctypes.data_as(C.c_void_p), npol,
        uidx.ctypes.data_as(C.c_void_p), ualpha.ctypes.data_as(C.c_void_p), nunpol,
        mode, out.data_ptr(), C.c_void_p(stream)))
    return out.cpu().numpy() if from_numpy else out


def polarised_intensity(vis, stokes_pol):
    r""":math:`\sqrt{Q^2 + U^2 + V^2}` (every term entered as :math:`|.|^2`) of
    ``(row, chan, corr)`` visibilities, shape ``(row, chan, 1)``, dtype of
    ``vis`` (stokes.py:157-209).  ``stokes_pol``: the ``(c1, c2, a, s1, s2)``
    tuples of :func:`stokes_corr_map`."""
    return _intensity(vis, stokes_pol, (), 0)


def unpolarised_intensity(vis, stokes_unpol, stokes_pol):
    r""":math:`|I| - \sqrt{Q^2 + U^2 + V^2}` (stokes.py:79-153)."""
    if not len(stokes_unpol) == 1:
        raise ValueError("There should be exactly one entry for unpolarised stokes (stokes_unpol)")
    if not len(stokes_pol) > 0:
        raise ValueError("No entries for polarised stokes (stokes_pol)")
    return _intensity(vis, stokes_pol, stokes_unpol, 1)
