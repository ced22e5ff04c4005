"""Graph-level drop-in for ``tricolour.dask_wrappers.sum_threshold_flagger``
(reference ``tricolour/dask_wrappers.py:23-46``): same signature, same
``blockwise`` layer over the window schema, same return value; the per-block
callable is :func:`tricolour_amd.flagging.sum_threshold_flagger` (numpy block
in -> H2D -> HIP kernels -> D2H -> numpy block out), so it slots into the
existing graph over Measurement Sets unchanged.

dask is imported lazily: it is needed only when the graph is built.
"""
from tricolour_amd.flagging import sum_threshold_flagger as amd_sum_threshold_flagger
from tricolour_amd.flagging import uvcontsub_flagger as amd_uvcontsub_flagger
from tricolour_amd.packing import _WINDOW_SCHEMA


def sum_threshold_flagger(vis, flag, **kwargs):
    """
    Dask wrapper for :func:`tricolour_amd.flagging.sum_threshold_flagger`
    """
    import dask.array as da
    import dask.blockwise as db
    from dask.highlevelgraph import HighLevelGraph

    # dask.blockwise.blockwise rather than dask.array.blockwise, as in the
    # reference: blocks are chunked along "bl" only and differ in size
    token = da.core.tokenize(vis, flag, kwargs)
    name = 'sum-threshold-flagger-' + token

    layers = db.blockwise(amd_sum_threshold_flagger, name, _WINDOW_SCHEMA,
                          vis.name, _WINDOW_SCHEMA,
                          flag.name, _WINDOW_SCHEMA,
                          numblocks={
                              vis.name: vis.numblocks,
                              flag.name: flag.numblocks,
                          },
                          **kwargs)

    graph = HighLevelGraph.from_collections(name, layers, (vis, flag))
    return da.Array(graph, name, vis.chunks, dtype=flag.dtype)


def uvcontsub_flagger(vis, flag, **kwargs):
    """
    Dask wrapper for :func:`tricolour_amd.flagging.uvcontsub_flagger`
    (reference ``tricolour/dask_wrappers.py:49-66``)
    """
    import dask.array as da
    import dask.blockwise as db
    from dask.highlevelgraph import HighLevelGraph

    name = 'uvcontsub-flagger-' + da.core.tokenize(vis, flag, **kwargs)

    layers = db.blockwise(amd_uvcontsub_flagger, name, _WINDOW_SCHEMA,
                          vis.name, _WINDOW_SCHEMA,
                          flag.name, _WINDOW_SCHEMA,
                          numblocks={
                              vis.name: vis.numblocks,
                              flag.name: flag.numblocks,
                          },
                          **kwargs)

    graph = HighLevelGraph.from_collections(name, layers, (vis, flag))
    return da.Array(graph, name, vis.chunks, dtype=flag.dtype)
