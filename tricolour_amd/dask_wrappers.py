"""Graph-level drop-ins for ``tricolour.dask_wrappers.sum_threshold_flagger``,
``uvcontsub_flagger`` (reference ``tricolour/dask_wrappers.py:23-66``) and the
Stokes intensity transforms (``:117-147``), plus the cheap strategy steps
``flag_nans_and_zeros``, ``apply_static_mask`` and ``flag_autos`` (``:69-115``).

Same call signatures and the same result as the reference wrappers: a dask
array with the chunking of ``vis`` and the dtype of ``flag`` whose graph holds
one task per baseline chunk.  Each task hands its numpy window block to the
HIP path (:mod:`tricolour_amd.flagging`: H2D, kernels, D2H), so the wrappers
slot into the existing graph over Measurement Sets unchanged.

dask is imported lazily: it is only needed when a graph is built.
"""
import numpy as np

from tricolour_amd.flagging import sum_threshold_flagger as amd_sum_threshold_flagger
from tricolour_amd.flagging import uvcontsub_flagger as amd_uvcontsub_flagger
from tricolour_amd.flagging import flag_nans_and_zeros as amd_flag_nans_and_zeros
from tricolour_amd.flagging import apply_static_mask as amd_apply_static_mask
from tricolour_amd.flagging import flag_autos as amd_flag_autos
from tricolour_amd.packing import _WINDOW_SCHEMA
from tricolour_amd.stokes import polarised_intensity as amd_polarised_intensity
from tricolour_amd.stokes import unpolarised_intensity as amd_unpolarised_intensity


def _window_blockwise(per_block, layer_name, vis, flag, kwargs):
    """One blockwise layer over the (bl, corr, time, chan) window schema.

    The low-level ``dask.blockwise.blockwise`` is used (as the reference does)
    because window blocks are chunked along ``bl`` only and differ in size, so
    the block counts have to be given explicitly."""
    import dask.array as da
    from dask.blockwise import blockwise
    from dask.highlevelgraph import HighLevelGraph

    block_counts = {vis.name: vis.numblocks, flag.name: flag.numblocks}
    layer = blockwise(per_block, layer_name, _WINDOW_SCHEMA,
                      vis.name, _WINDOW_SCHEMA, flag.name, _WINDOW_SCHEMA,
                      numblocks=block_counts, **kwargs)
    graph = HighLevelGraph.from_collections(layer_name, layer, dependencies=(vis, flag))
    return da.Array(graph, layer_name, chunks=vis.chunks, dtype=flag.dtype)


def sum_threshold_flagger(vis, flag, **kwargs):
    """Dask wrapper for :func:`tricolour_amd.flagging.sum_threshold_flagger`;
    the layer is named ``sum-threshold-flagger-<tokenize(vis, flag, kwargs)>``
    exactly as in the reference."""
    import dask.array as da
    token = da.core.tokenize(vis, flag, kwargs)
    return _window_blockwise(amd_sum_threshold_flagger, 'sum-threshold-flagger-' + token,
                             vis, flag, kwargs)


def uvcontsub_flagger(vis, flag, **kwargs):
    """Dask wrapper for :func:`tricolour_amd.flagging.uvcontsub_flagger`; the
    layer is named ``uvcontsub-flagger-<tokenize(vis, flag, **kwargs)>`` (the
    reference passes the keyword arguments to ``tokenize`` unpacked here)."""
    import dask.array as da
    token = da.core.tokenize(vis, flag, **kwargs)
    return _window_blockwise(amd_uvcontsub_flagger, 'uvcontsub-flagger-' + token,
                             vis, flag, kwargs)


_ROW_SCHEMA = ("row", "chan", "corr")


def _single_corr_blockwise(per_block, vis, **terms):
    """(row, chan, corr) -> (row, chan, 1), block by block."""
    import dask.array as da
    return da.blockwise(lambda block, **kw: per_block(block, **kw), _ROW_SCHEMA, vis, _ROW_SCHEMA,
                        adjust_chunks={"corr": 1}, dtype=vis.dtype, **terms)


def polarised_intensity(vis, stokes_pol):
    """Dask wrapper for :func:`tricolour_amd.stokes.polarised_intensity`."""
    return _single_corr_blockwise(amd_polarised_intensity, vis, stokes_pol=stokes_pol)


def unpolarised_intensity(vis, stokes_unpol, stokes_pol):
    """Dask wrapper for :func:`tricolour_amd.stokes.unpolarised_intensity`."""
    return _single_corr_blockwise(amd_unpolarised_intensity, vis, stokes_unpol=stokes_unpol, stokes_pol=stokes_pol)


_UBL_SCHEMA = ("bl", "bl-comp")


def flag_nans_and_zeros(vis_windows, flag_windows):
    """Dask wrapper for :func:`tricolour_amd.flagging.flag_nans_and_zeros`."""
    import dask.array as da
    return da.blockwise(lambda v, f: amd_flag_nans_and_zeros(v, f), _WINDOW_SCHEMA,
                        vis_windows, _WINDOW_SCHEMA, flag_windows, _WINDOW_SCHEMA,
                        meta=np.empty((0, 0, 0, 0), dtype=flag_windows.dtype))


def apply_static_mask(flag, ubl, antspos, masks, spw_chanlabels, spw_chanwidths, **kwargs):
    """Dask wrapper for :func:`tricolour_amd.flagging.apply_static_mask`; ``ubl``
    is chunked like the baselines of ``flag``, everything else is passed whole."""
    import dask.array as da

    def per_block(f, u, pos, m, labels, widths, **kw):
        # "bl-comp" is contracted: dask hands the baseline rows as a one-element list
        return amd_apply_static_mask(f, u[0], pos, m, labels, widths, **kw)

    return da.blockwise(per_block, _WINDOW_SCHEMA, flag, _WINDOW_SCHEMA, ubl, _UBL_SCHEMA, antspos, None, masks, None,
                        spw_chanlabels, None, spw_chanwidths, None,
                        meta=np.empty((0, 0, 0, 0), dtype=flag.dtype), **kwargs)


def flag_autos(flag, ubl, **kwargs):
    """Dask wrapper for :func:`tricolour_amd.flagging.flag_autos` (which, like the
    reference's, unwraps the one-element list dask makes of the ``bl-comp`` axis)."""
    import dask.array as da
    return da.blockwise(lambda f, u: amd_flag_autos(f, u), _WINDOW_SCHEMA, flag, _WINDOW_SCHEMA,
                        ubl, _UBL_SCHEMA, meta=np.empty((0, 0, 0, 0), dtype=flag.dtype))
