"""The cheap strategy steps around sum_threshold:
flag_nans_and_zeros, flag_autos, apply_static_mask and the StrategyExecutor
combination rules.  CPU part: the oracle's numpy restatements against the
expectations of the reference's own tests
(tricolour/tests/test_flagging_additional.py:53-192).  GPU part: the HIP
kernels against the oracle."""
import numpy as np
import pytest

WSRT = np.array([
    [3828763.10544699, 442449.10566454, 5064923.00777],
    [3828746.54957258, 442592.13950824, 5064923.00792],
    [3828729.99081359, 442735.17696417, 5064923.00829],
    [3828713.43109885, 442878.2118934, 5064923.00436],
    [3828696.86994428, 443021.24917264, 5064923.00397],
    [3828680.31391933, 443164.28596862, 5064923.00035],
    [3828663.75159173, 443307.32138056, 5064923.00204],
    [3828647.19342757, 443450.35604638, 5064923.0023],
    [3828630.63486201, 443593.39226634, 5064922.99755],
    [3828614.07606798, 443736.42941621, 5064923.],
    [3828609.94224429, 443772.19450029, 5064922.99868],
    [3828601.66208572, 443843.71178407, 5064922.99963],
    [3828460.92418735, 445059.52053929, 5064922.99071],
    [3828452.64716351, 445131.03744105, 5064922.98793]], dtype=np.float64)


def _ubl():
    a1, a2 = np.triu_indices(WSRT.shape[0], 0)
    u = np.unique(np.stack([a1, a2], axis=1), axis=1)
    return np.concatenate([np.arange(u.shape[0])[:, None], u], axis=1)


def _mask_setup():
    nchan = 16
    chan_freqs = np.linspace(.856e9, 2 * .856e9, nchan, dtype=np.float64)
    chan_widths = np.zeros_like(chan_freqs)
    chan_widths[0:-1] = np.diff(chan_freqs)
    chan_widths[-1] = chan_widths[0]
    mask_one = np.asarray([chan_freqs[2] + 128., chan_freqs[10]])[:, None]
    mask_two = np.asarray([chan_freqs[4] - 64, chan_freqs[11] + 64, chan_freqs[5] - 128])[:, None]
    return chan_freqs, chan_widths, mask_one, mask_two


def test_oracle_flag_nans_and_zeros(oracle):
    ubl = _ubl()
    shape = (ubl.shape[0], 4, 10, 16)
    rs = np.random.RandomState(0)
    vis = rs.random_sample(shape) + 1j * rs.random_sample(shape)
    vis[4, 2, 4, 5] = 0
    vis[0, 1, 2, 7] = np.nan + np.nan * 1j
    out = oracle.flag_nans_and_zeros(vis, np.zeros(shape, np.uint8))
    assert out[4, 2, 4, 5] == 1 and out[0, 1, 2, 7] == 1 and out.sum() == 2
    flags = rs.randint(0, 2, shape).astype(np.uint8)
    np.testing.assert_array_equal(oracle.flag_nans_and_zeros(vis, flags),
                                  flags | (vis == 0) | np.isnan(vis))


def test_oracle_flag_autos_and_static_mask(oracle):
    ubl = _ubl()
    flags = np.ones((ubl.shape[0], 4, 10, 16), np.uint8)
    sel = ubl[:, 1] == ubl[:, 2]
    flags[sel] = 0
    assert np.all(oracle.flag_autos(flags, [ubl])[sel] == 1)

    cf, cw, m1, m2 = _mask_setup()
    z = np.zeros((ubl.shape[0], 4, 10, 16), np.uint8)
    out = oracle.apply_static_mask(z, ubl, WSRT, [m1], cf, cw, "or")
    cs = np.zeros(16, bool); cs[[2, 10]] = True
    assert np.all(out[..., cs] == 1) and np.all(out[..., ~cs] == 0)
    out = oracle.apply_static_mask(z, ubl, WSRT, [m1, m2], cf, cw, "or")
    cs[[4, 11, 5]] = True
    assert np.all(out[..., cs] == 1) and np.all(out[..., ~cs] == 0)
    out = oracle.apply_static_mask(z, ubl, WSRT, [m1, m2], cf, cw, "override")
    cs = np.zeros(16, bool); cs[[4, 11, 5]] = True
    assert np.all(out[..., cs] == 1) and np.all(out[..., ~cs] == 0)
    out = oracle.apply_static_mask(z, ubl, WSRT, [m1, m2], cf, cw, "or", (1e3, 2e4))
    d2 = 0.5 * ((WSRT[ubl[:, 1]] - WSRT[ubl[:, 2]])**2).sum(axis=1)
    bl = (d2 > 1e6) & (d2 < 4e8)
    cs = np.zeros(16, bool); cs[[2, 10, 4, 11, 5]] = True
    assert np.all(out[np.ix_(bl, range(4), range(10), cs)] == 1)
    assert np.all(out[np.ix_(~bl, range(4), range(10), ~cs)] == 0)


def test_casa_style_range():
    from tricolour_amd.util import casa_style_range
    assert casa_style_range("") == (0, np.inf) and casa_style_range("*") == (0, np.inf)
    assert casa_style_range("0~550") == [0.0, 550.0]
    assert casa_style_range("0~250m") == [0.0, 250.0]
    with pytest.raises(ValueError):
        casa_style_range("abc")
    with pytest.raises(ValueError):
        casa_style_range(5)


@pytest.mark.gpu
def test_gpu_strategy_steps(gpu, oracle):
    import torch
    from tricolour_amd import flagging
    ubl = _ubl()
    shape = (ubl.shape[0], 4, 10, 16)
    rs = np.random.RandomState(1)
    vis = (rs.random_sample(shape) + 1j * rs.random_sample(shape)).astype(np.complex64)
    vis[4, 2, 4, 5] = 0
    vis[0, 1, 2, 7] = np.nan + np.nan * 1j
    vis[3, 0, 1, 1] = complex(0.0, np.nan)
    vis[3, 0, 1, 2] = complex(0.0, 1e-30)
    flags = rs.randint(0, 2, shape).astype(np.uint8)
    got = flagging.flag_nans_and_zeros(vis, flags)
    assert got.dtype == np.uint8
    np.testing.assert_array_equal(got, oracle.flag_nans_and_zeros(vis, flags))
    gotb = flagging.flag_nans_and_zeros(torch.from_numpy(vis).cuda(), torch.from_numpy(flags.astype(bool)).cuda())
    assert gotb.dtype == torch.bool
    np.testing.assert_array_equal(gotb.cpu().numpy(), oracle.flag_nans_and_zeros(vis, flags).astype(bool))

    f0 = rs.randint(0, 2, shape).astype(bool)
    np.testing.assert_array_equal(flagging.flag_autos(f0, [ubl]), oracle.flag_autos(f0, [ubl]))

    cf, cw, m1, m2 = _mask_setup()
    for mode in ("or", "override"):
        for uv, uvp in (("", (0, np.inf)), ("1000.0~20000.0", (1e3, 2e4))):
            got = flagging.apply_static_mask(f0, ubl, WSRT, [m1, m2], cf, cw, accumulation_mode=mode, uvrange=uv)
            exp = oracle.apply_static_mask(f0, ubl, WSRT, [m1, m2], cf, cw, mode, uvp)
            np.testing.assert_array_equal(got, exp)
    assert np.array_equal(f0, f0.copy())   # inputs untouched


@pytest.mark.gpu
def test_gpu_strategy_chain(gpu, oracle):
    """A default.yaml-like chain (without uvcontsub) with the executor's
    combination rules, device-resident, against the same chain in numpy."""
    import torch
    from tricolour_amd.strategies import apply_strategies
    a1, a2 = np.triu_indices(4, 0)
    ubl = np.stack([np.arange(a1.size), a1, a2], axis=1)
    ants = WSRT[:4]
    shape = (ubl.shape[0], 2, 48, 64)
    rs = np.random.RandomState(3)
    vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 20] *= 8
    vis[2, 1, 5, :] = 0
    vis[1, 0, 7, 9] = np.nan
    flags = rs.uniform(size=shape) < 0.02
    cf = np.linspace(1e9, 1.1e9, 64)
    cw = np.full(64, cf[1] - cf[0])
    masks = [np.array([cf[30], cf[31] + 10.0])[:, None]]
    st_kw = dict(outlier_nsigma=10, background_iterations=2, num_major_iterations=2, freq_chunks=4)
    strategies = [
        dict(task="flag_nans_zeros"),
        dict(task="apply_static_mask", kwargs=dict(accumulation_mode="or", uvrange="")),
        dict(task="sum_threshold", kwargs=st_kw),
        dict(task="flag_autos"),
        dict(task="combine_with_input_flags"),
    ]
    got = apply_strategies(strategies, torch.from_numpy(flags).cuda(), torch.from_numpy(vis).cuda(),
                           ubl=ubl, ant_pos=ants, chan_freq=cf, chan_width=cw, masked_channels=masks)
    f = oracle.flag_nans_and_zeros(vis, flags)
    f = oracle.apply_static_mask(f, ubl, ants, masks, cf, cw, "or") | f
    f = oracle.sum_threshold_flagger(vis, f, **st_kw) | f
    f = oracle.flag_autos(f, [ubl]) | f
    f = f | flags
    np.testing.assert_array_equal(got.cpu().numpy(), f)
