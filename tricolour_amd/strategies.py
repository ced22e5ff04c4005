"""Device-resident strategy chain: the combination rules of
``StrategyExecutor.apply_strategies`` (reference
``tricolour/apps/tricolour/strat_executor.py:29-83``) applied to torch tensors
that stay in HBM between steps."""
from tricolour_amd import flagging


def apply_strategies(strategies, flag_windows, vis_windows, ubl=None, ant_pos=None,
                     chan_freq=None, chan_width=None, masked_channels=None):
    """Runs the ordered ``strategies`` (dicts with ``task`` and ``kwargs``, as
    parsed from the YAML) on (bl, corr, time, chan) tensors and returns the
    final flags."""
    import torch
    original = flag_windows.clone() if torch.is_tensor(flag_windows) else flag_windows.copy()
    lor = torch.logical_or if torch.is_tensor(flag_windows) else (lambda a, b: a | b)
    for strategy in strategies:
        try:
            task = strategy['task']
        except KeyError:
            raise ValueError("strategy has no 'task': %s" % strategy)
        kw = strategy.get('kwargs') or {}
        if task == "sum_threshold":
            new_flags = flagging.sum_threshold_flagger(vis_windows, flag_windows, **kw)
            flag_windows = lor(new_flags, flag_windows)          # strat_executor.py:43
        elif task == "uvcontsub_flagger":
            # discards the previous flags by design (strat_executor.py:44-51)
            flag_windows = flagging.uvcontsub_flagger(vis_windows, flag_windows, **kw)
        elif task == "flag_autos":
            new_flags = flagging.flag_autos(flag_windows, [ubl])
            flag_windows = lor(new_flags, flag_windows)          # :54
        elif task == "combine_with_input_flags":
            flag_windows = lor(flag_windows, original)           # :59
        elif task == "unflag":
            flag_windows = flag_windows * 0 if not torch.is_tensor(flag_windows) else torch.zeros_like(flag_windows)
        elif task == "flag_nans_zeros":
            flag_windows = flagging.flag_nans_and_zeros(vis_windows, flag_windows)   # :63
        elif task == "apply_static_mask":
            new_flags = flagging.apply_static_mask(flag_windows, ubl, ant_pos, masked_channels,
                                                   chan_freq, chan_width, **kw)
            if kw["accumulation_mode"].strip() == "or":          # :75-78
                flag_windows = lor(new_flags, flag_windows)
            else:
                flag_windows = new_flags
        else:
            raise ValueError("Task '%s' does not name a valid task", task)
    return flag_windows
