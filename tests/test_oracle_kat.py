"""Known-answer vectors of the reference's own unit tests
(tricolour/tests/test_flagging.py) replayed against the oracle's restatement
of each helper.  Only the vectors (data) are restated here."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter, gaussian_filter1d


# --- _average_freq (test_flagging.py:36-130) --------------------------------
@pytest.fixture
def small():
    data = np.arange(30, dtype=np.float32).reshape(1, 5, 6).repeat(2, axis=0)
    flags = np.zeros(data.shape, np.bool_)
    flags[0, 3, :] = 1
    flags[0, :, 4] = 1
    flags[:, 2, 0] = 1
    flags[:, 2, 5] = 1
    return data, flags


def test_average_freq_one(oracle, small):
    data, flags = small
    ad, af = oracle.average_freq(data, flags, 1)
    exp = data.copy()
    exp[flags] = 0
    assert ad.dtype == np.float32 and af.dtype == np.bool_
    np.testing.assert_array_equal(exp, ad)
    np.testing.assert_array_equal(flags, af)


def test_average_freq_divides(oracle, small):
    data, flags = small
    exp_d = np.array([[[0.5, 2.5, 5.0], [6.5, 8.5, 11.0], [13.0, 14.5, 0.0], [0.0, 0.0, 0.0],
                       [24.5, 26.5, 29.0]],
                      [[0.5, 2.5, 4.5], [6.5, 8.5, 10.5], [13.0, 14.5, 16.0], [18.5, 20.5, 22.5],
                       [24.5, 26.5, 28.5]]], np.float32)
    exp_f = np.array([[[False] * 3, [False] * 3, [False, False, True], [True] * 3, [False] * 3],
                      [[False] * 3] * 5])
    ad, af = oracle.average_freq(data, flags, 2)
    np.testing.assert_array_equal(exp_d, ad)
    np.testing.assert_array_equal(exp_f, af)


def test_average_freq_uneven(oracle, small):
    data, flags = small
    exp_d = np.array([[[1.5, 5.0], [7.5, 11.0], [14.0, 0.0], [0.0, 0.0], [25.5, 29.0]],
                      [[1.5, 4.5], [7.5, 10.5], [14.0, 16.0], [19.5, 22.5], [25.5, 28.5]]],
                     np.float32)
    exp_f = np.array([[[False, False], [False, False], [False, True], [True, True], [False, False]],
                      [[False, False]] * 5], np.bool_)
    ad, af = oracle.average_freq(data, flags, 4)
    np.testing.assert_array_equal(exp_d, ad)
    np.testing.assert_array_equal(exp_f, af)


# --- _time_median (test_flagging.py:133-151) --------------------------------
def test_time_median(oracle):
    data = np.array([[2.0, 1.0, 2.0, 5.0], [3.0, 1.0, 8.0, 6.0], [4.0, 1.0, 4.0, 7.0],
                     [5.0, 1.0, 5.0, 6.5], [1.5, 1.0, 1.5, 5.5]], np.float32)
    flags = np.array([[0, 1, 0, 1], [0, 1, 1, 0], [0, 1, 0, 1], [0, 1, 0, 1], [0, 1, 0, 1]], np.bool_)
    od, of = oracle.time_median(data, flags)
    np.testing.assert_array_equal(np.array([[3.0, 0.0, 3.0, 6.0]], np.float32), od)
    np.testing.assert_array_equal(np.array([[0, 1, 0, 0]], np.bool_), of)


# --- medians (test_flagging.py:154-179) -------------------------------------
def test_median_abs(oracle):
    data = np.array([[-2.0, -6.0, 4.5], [1.5, 3.3, 0.5]], np.float32)
    flags = np.array([[0, 0, 0], [0, 1, 0]], np.uint8)
    assert oracle.median_abs(data, flags) == 2.0
    assert np.isnan(oracle.median_abs(data, np.ones_like(flags)))
    np.testing.assert_array_equal(np.array([[1.75, 6.0, 2.5]]), oracle.median_abs_axis0(data, flags))
    flags[:, 1] = True
    np.testing.assert_array_equal(np.array([[1.75, np.nan, 2.5]]), oracle.median_abs_axis0(data, flags))


# --- _linearly_interpolate_nans (test_flagging.py:182-224) -------------------
Y = np.array([np.nan, np.nan, 4.0, np.nan, np.nan, 10.0, np.nan, -2.0, np.nan, np.nan])
YE = np.array([4.0, 4.0, 4.0, 6.0, 8.0, 10.0, 4.0, -2.0, -2.0, -2.0])


def test_interpolate_nans(oracle):
    y = Y.astype(np.float32)
    oracle.linearly_interpolate_nans1d(y)
    np.testing.assert_allclose(YE.astype(np.float32), y, rtol=1e-6)
    y = YE.astype(np.float32)
    oracle.linearly_interpolate_nans1d(y)
    np.testing.assert_array_equal(YE.astype(np.float32), y)
    y = np.full(10, np.nan, np.float32)
    oracle.linearly_interpolate_nans1d(y)
    np.testing.assert_array_equal(np.zeros(10, np.float32), y)


# --- _box_gaussian_filter (test_flagging.py:228-289) -------------------------
def test_box_one_pass(oracle):
    a = np.array([50.0, 10.0, 60.0, -70.0, 30.0, 20.0, -15.0], np.float32)
    b = oracle.box_gaussian_filter1d(a, 2, 1)
    np.testing.assert_equal(np.array([24.0, 10.0, 16.0, 10.0, 5.0, -7.0, 7.0], np.float32), b)


def test_box_width(oracle):
    a = np.zeros((1, 200), np.float32)
    a[:, 100] = 1.0
    b = oracle.box_gaussian_filter(a, (0.0, 10.0))
    x = np.arange(a.size) - a.size // 2
    np.testing.assert_allclose(1.0, np.sum(b), rtol=1e-5)
    np.testing.assert_allclose(0.0, np.sum(x * b), atol=1e-5)
    np.testing.assert_allclose(np.sqrt(np.sum(x * x * b)), 10.0, atol=1)


def test_box_2d_and_axes(oracle):
    rs = np.random.RandomState(seed=1)
    data = rs.uniform(size=(77, 53)).astype(np.float32)
    expected = gaussian_filter(data, (8, 2.3), mode='constant')
    np.testing.assert_allclose(expected, oracle.box_gaussian_filter(data, (8, 2.3)), rtol=1e-1)
    out0 = oracle.box_gaussian_filter(data, (8.0, 0.0))
    out1 = oracle.box_gaussian_filter(np.ascontiguousarray(data.T), (0.0, 8.0)).T
    np.testing.assert_array_equal(out0, out1)   # test_flagging.py:268-277


def test_box_edge(oracle):
    rs = np.random.RandomState(seed=1)
    data = np.zeros((1, 200), np.float32)
    data[:, 80:120] = rs.uniform(size=(1, 40))
    fdata = oracle.box_gaussian_filter(data, (0.0, 3.0))
    fcore = oracle.box_gaussian_filter(np.ascontiguousarray(data[:, 80:120]), (0.0, 3.0))
    np.testing.assert_allclose(fdata[:, 80:120], fcore, rtol=1e-5)


def test_box_denominator_is_square_and_multiply(oracle):
    """D1: numba's float32(d)**4 is fl32(fl32(d*d)**2), which differs from
    powf for d in {65, 69, 71, 73, 75, 795}."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    differ = []
    for r in range(0, 700):
        d = np.float32(2 * r + 1)
        sq = np.float32(np.float32(d * d) * np.float32(d * d))
        got = oracle.lib().tro_box_denominator(ctypes.c_int64(r), ctypes.c_int64(4))
        assert np.float32(got) == sq
        if np.float32(libm.powf(float(d), 4.0)) != sq:
            differ.append(int(d))
    assert differ == [65, 69, 71, 73, 75, 795]


# --- masked filter (test_flagging.py:292-332) --------------------------------
def _masked_expected(data, flags, sigma, truncate):
    weight = 1.0 - flags
    d = data * weight
    for i, (s, t) in enumerate(zip(sigma, truncate)):
        weight = gaussian_filter1d(weight, s, axis=i, mode='constant', truncate=t)
        d = gaussian_filter1d(d, s, axis=i, mode='constant', truncate=t)
    with np.errstate(invalid='ignore', divide='ignore'):
        d /= weight
    return d


def test_masked_filter(oracle):
    rs = np.random.RandomState(seed=1)
    data = rs.uniform(size=(77, 53)).astype(np.float32)
    flags = rs.uniform(size=(77, 53)) >= 0.5
    exp = _masked_expected(data, flags, (5, 2.3), (4.0, 4.0))
    np.testing.assert_allclose(exp, oracle.masked_gaussian_filter(data, flags, (5, 2.3)), rtol=1e-1)
    flags[:] = False
    flags[30:70, 10:40] = True
    sigma = (3, 3.3)
    radius = [int(0.5 * np.sqrt(12.0 * s**2 / 4 + 1)) for s in sigma]
    trunc = [4 * r / s for r, s in zip(radius, sigma)]
    exp = _masked_expected(data, flags, sigma, trunc)
    act = oracle.masked_gaussian_filter(data, flags, sigma)
    np.testing.assert_allclose(exp, act, rtol=1e-1)
    assert 0 < np.sum(np.isnan(exp))


# --- _get_background2d (test_flagging.py:335-421) ----------------------------
def _bg(oracle, data, flags=None, iterations=1, spike_width=(10.0, 10.0), reject=2.0):
    if flags is None:
        flags = np.zeros(data.shape, np.uint8)
    sw = np.array(spike_width, np.float32)
    return oracle.get_background2d(data, flags, iterations, sw, reject, np.array([0, data.shape[1]]))


def test_background2d(oracle):
    shape = (95, 86)
    data = np.ones(shape, np.float32) * 7.5
    np.testing.assert_allclose(data, _bg(oracle, data), rtol=1e-5)
    flags = np.ones(shape, np.uint8)
    np.testing.assert_array_equal(np.zeros(shape, np.float32), _bg(oracle, data, flags))
    d2 = data.copy()
    d2[::3] = 20.0
    f2 = np.zeros(shape, np.uint8)
    f2[::3] = True
    np.testing.assert_allclose(data, _bg(oracle, d2, f2), rtol=1e-5)


def test_background2d_interpolate(oracle):
    shape = (95, 86)
    data = np.ones(shape, np.float32) * 7.5
    flags = np.zeros(shape, np.uint8)
    data[:, 70:] = 3.0
    flags[:, 30:70] = True
    rs = np.random.RandomState(seed=1)
    data[:50, :] += rs.uniform(-0.001, 0.001, data[0:50].shape)
    bg = _bg(oracle, data, flags, spike_width=(2.5, 2.5), reject=5.0)
    exp = np.zeros_like(data)
    exp[:, :37] = 7.5
    exp[:, 63:] = 3.0
    exp[:, 37:63] = np.linspace(7.5, 3.0, 26)
    np.testing.assert_allclose(exp[56:], bg[56:], rtol=1e-4)
    np.testing.assert_allclose(exp[:56], bg[:56], rtol=1e-2)


def test_background2d_iterations(oracle):
    shape = (95, 86)
    expected = np.ones(shape, np.float32) * 7.5
    rs = np.random.RandomState(seed=1)
    data = expected + (rs.standard_normal(shape) * 0.1).astype(np.float32)
    data[20:50, 30:80] += 15
    np.testing.assert_allclose(expected, _bg(oracle, data, iterations=3), rtol=1e-2)


# --- _sum_threshold (test_flagging.py:424-501) -------------------------------
WINDOWS = np.array([1, 2, 4, 8])


def test_sum_threshold_all_flagged(oracle):
    data = np.arange(30, dtype=np.float32).reshape(5, 6)
    flags = np.ones(data.shape, np.bool_)
    out = oracle.sum_threshold(data, flags, 0, np.array([1, 2, 4]), 4.5, 1.3)
    np.testing.assert_array_equal(np.zeros_like(flags), out)


@pytest.mark.parametrize("axis", [0, 1])
def test_sum_threshold_basic(oracle, axis):
    rs = np.random.RandomState(seed=1)
    data = rs.standard_normal((100, 90)).astype(np.float32) * 3.0
    rfi = np.zeros_like(data)
    rfi[10, 20] = 100.0
    rfi[80, 80] = -100.0
    rfi[:, 40] = rs.uniform(80.0, 120.0, size=(100,))
    rfi[:, 2] = -rfi[:, 40]
    rfi[:, 60:67] = rs.uniform(15.0, 20.0, size=(100, 7))
    rfi[:, 10:17] = -rfi[:, 60:67]
    in_flags = np.zeros(data.shape, np.bool_)
    expected = rfi != 0
    data += rfi
    if axis == 0:
        data, in_flags = data.T.copy(), in_flags.T.copy()
    out = oracle.sum_threshold(data, in_flags, axis, WINDOWS, 4.5, 1.3)
    if axis == 0:
        out = out.T
    assert np.sum(expected != out) / data.size < 0.01
    for region in (np.s_[8:13, 18:23], np.s_[78:83, 78:83]):
        np.testing.assert_equal(expected[region], out[region])


def test_sum_threshold_existing(oracle):
    rs = np.random.RandomState(seed=1)
    data = rs.standard_normal((100, 90)).astype(np.float32) * 3.0
    in_flags = np.zeros(data.shape, np.bool_)
    data[:48] += 1000.0
    in_flags[:48] = True
    data[70, 0], data[70, 1], data[70, 2], data[70, 3] = 12.5, -12.5, 20.0, -20.0
    out = oracle.sum_threshold(data, in_flags, 0, WINDOWS, 5, 1.3)
    np.testing.assert_array_equal([False, False, True, True], out[70, :4])


# --- whole flagger behaviour (test_flagging.py:504-649, function path) -------
def test_flagger_all_flagged(oracle):
    data = np.zeros((4, 1, 100, 80), np.float32)
    flags = np.ones(data.shape, np.bool_)
    out = oracle.sum_threshold_flagger(data, flags)
    np.testing.assert_array_equal(np.zeros_like(flags), out)


def test_flagger_variable_noise(oracle):
    rs = np.random.RandomState(seed=1)
    shape = (1, 234, 345)
    noise = rs.standard_normal(shape)
    noise *= np.arange(shape[2])[np.newaxis, np.newaxis, :] / shape[2]
    noise = noise.astype(np.float32)
    noise[:, 100, 17] = 1.0
    noise[:, 200, 170] = 1.0
    data = np.abs(np.ones(shape, np.float32) * 11 + noise)[None]
    out = oracle.sum_threshold_flagger(data, np.zeros(data.shape, np.bool_), num_major_iterations=1)
    assert out[0, 0, 100, 17]
    assert not out[0, 0, 200, 170]
