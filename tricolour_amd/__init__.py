"""tricolour_amd -- MI355X-native SumThreshold RFI flagger.

Drop-in for one hot path of ratt-ru/tricolour: the per-baseline
(time x chan) window pipeline behind
``tricolour.dask_wrappers.sum_threshold_flagger`` /
``tricolour.flagging.sum_threshold_flagger`` and the window pack / unpack
either side of it.  Host Python holds visibility / flag chunks as
PyTorch-ROCm tensors and calls hand-written HIP kernels through the C ABI of
``include/tricolour_amd.h``.
"""
__version__ = "0.1.0"

from tricolour_amd.flagging import sum_threshold_flagger  # noqa: F401,E402
