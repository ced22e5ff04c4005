"""uvcontsub_flagger.  The reference routine is plain NumPy
and its own tests do not cover it; G11 holds its output (run under this
image's NumPy).  The oracle restatement must reproduce G11 exactly; the HIP
path follows the same float32 semantics but cannot reproduce NumPy's FFT
rounding, so it is held to a flag-agreement rate."""
import numpy as np
import pytest

from conftest import load_golden


def _cases():
    d, _ = load_golden("G11_uvcontsub.npz")
    out = []
    for name in ("a", "b"):
        kw = {k[len("kw_%s_" % name):]: d[k].tolist() for k in d.files if k.startswith("kw_%s_" % name)}
        out.append((d["vis"], d["flags"], kw, d["out_" + name]))
    return out, str(d["numpy_version"])


def test_oracle_reproduces_reference():
    from oracle import oracle
    cases, npver = _cases()
    if npver.split(".")[0] != np.__version__.split(".")[0]:
        pytest.skip("fixture generated with NumPy %s" % npver)
    for vis, flags, kw, exp in cases:
        got = oracle.uvcontsub_flagger(vis, flags, **kw)
        assert np.array_equal(got, exp)
        assert got[1, 1].all()            # fully flagged product untouched


@pytest.mark.gpu
def test_gpu_uvcontsub_agreement(gpu):
    from oracle import oracle
    from tricolour_amd import flagging
    cases, _ = _cases()
    for vis, flags, kw, exp in cases:
        got = flagging.uvcontsub_flagger(vis, flags, **kw)
        assert got.shape == exp.shape and got.dtype == np.bool_
        agree = (got == exp).mean()
        assert agree >= 0.999, "flag agreement %.5f" % agree
        assert got[1, 1].all()
    # a larger random block against the oracle
    rs = np.random.RandomState(3)
    shape = (3, 2, 64, 512)
    x = np.linspace(0, 1, shape[3])
    vis = ((2 + np.cos(7 * x))[None, None, None, :] + 0.3 * rs.standard_normal(shape)
           + 1j * 0.3 * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 100] += 5
    vis[2, 0, 10] += 3
    flags = rs.uniform(size=shape) < 0.02
    kw = dict(major_cycles=5, or_original_from_cycle=1, taylor_degrees=20, sigma=10.0)
    exp = oracle.uvcontsub_flagger(vis, flags, **kw)
    got = flagging.uvcontsub_flagger(vis, flags, **kw)
    assert (got == exp).mean() >= 0.9995


@pytest.mark.gpu
def test_gpu_config4_chain(gpu):
    """BASELINE config 4 in miniature: static mask -> flag_autos -> uvcontsub
    (replaces the running flags) -> sum_threshold, device-resident, against the
    same chain evaluated with the oracle."""
    import torch
    from oracle import oracle
    from tricolour_amd.strategies import apply_strategies
    a1, a2 = np.triu_indices(3, 0)
    ubl = np.stack([np.arange(a1.size), a1, a2], axis=1)
    ants = np.array([[0., 0., 0.], [100., 0., 0.], [0., 300., 0.]])
    shape = (ubl.shape[0], 2, 48, 128)
    rs = np.random.RandomState(4)
    x = np.linspace(0, 1, shape[3])
    vis = ((3 + np.sin(5 * x))[None, None, None, :] + 0.3 * rs.standard_normal(shape)
           + 1j * 0.3 * rs.standard_normal(shape)).astype(np.complex64)
    vis[..., 33] += 6
    flags = rs.uniform(size=shape) < 0.02
    cf = np.linspace(1e9, 1.1e9, shape[3])
    cw = np.full(shape[3], cf[1] - cf[0])
    masks = [np.array([cf[60]])[:, None]]
    uv_kw = dict(major_cycles=3, or_original_from_cycle=1, taylor_degrees=20, sigma=15.0)
    st_kw = dict(outlier_nsigma=10, background_iterations=2, num_major_iterations=1)
    strategies = [dict(task="apply_static_mask", kwargs=dict(accumulation_mode="or", uvrange="")),
                  dict(task="flag_autos"),
                  dict(task="uvcontsub_flagger", kwargs=uv_kw),
                  dict(task="sum_threshold", kwargs=st_kw)]
    got = apply_strategies(strategies, torch.from_numpy(flags).cuda(), torch.from_numpy(vis).cuda(),
                           ubl=ubl, ant_pos=ants, chan_freq=cf, chan_width=cw, masked_channels=masks)
    f = oracle.apply_static_mask(flags, ubl, ants, masks, cf, cw, "or") | flags
    f = oracle.flag_autos(f, [ubl]) | f
    f = oracle.uvcontsub_flagger(vis, f, **uv_kw)
    f = oracle.sum_threshold_flagger(vis, f, **st_kw) | f
    assert (got.cpu().numpy() == f).mean() >= 0.999


UV_AB_SCRIPT = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
from tricolour_amd import flagging
rs = np.random.RandomState(8)
shape = (3, 2, 64, 512)
x = np.linspace(0, 1, shape[3])
vis = ((2 + np.cos(7 * x))[None, None, None, :] + 0.3 * rs.standard_normal(shape)
       + 1j * 0.3 * rs.standard_normal(shape)).astype(np.complex64)
vis[..., 100] += 5
vis[2, 0, 10] += 3
vis[0, 1, 5, 7] = np.nan
flags = rs.uniform(size=shape) < 0.02
flags[1, 1] = True
out = flagging.uvcontsub_flagger(vis, flags, major_cycles=5, or_original_from_cycle=1, taylor_degrees=20, sigma=10.0)
print("DIGEST", hashlib.sha1(np.packbits(out).tobytes()).hexdigest(), int(out.sum()))
'''


@pytest.mark.gpu
def test_gpu_uvcontsub_vector_kernels_match_scalar(gpu):
    """The four-samples-per-thread residual / apply kernels and the scalar ones (TRI_UV_SCALAR=1, read once per
    process) give the same flags bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    digests = []
    for scalar in ("0", "1"):
        env = dict(os.environ)
        env["TRI_UV_SCALAR"] = scalar
        p = subprocess.run([sys.executable, "-c", UV_AB_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        digests.append([l for l in p.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1], digests
