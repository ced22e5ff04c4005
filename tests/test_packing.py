"""pack / unpack: the oracle's restatement of packing.py:243-278, 369-415
against the reference's own outputs (G8, G9 captured through its dask graph),
the host row map, and -- on a GPU -- the HIP scatter / gather kernels."""
import numpy as np
import pytest

from conftest import load_golden


def test_oracle_pack_unpack_vs_reference(oracle):
    d, _ = load_golden("G8_packing.npz")
    vw, fw = oracle.pack_data(d["time_inv"], d["ubl"], d["ant1"], d["ant2"], d["data"], d["flag"],
                              int(d["ntime"]))
    ref_v = d["vis_windows"]
    assert np.array_equal(np.isnan(vw.real), np.isnan(ref_v.real))
    ok = ~np.isnan(ref_v.real)
    assert np.array_equal(vw[ok], ref_v[ok])
    assert np.array_equal(fw, d["flag_windows"].astype(bool))
    up = oracle.unpack_data(d["time_inv"], d["ubl"], d["ant1"], d["ant2"], d["out_windows"])
    assert np.array_equal(up, d["unpacked"].astype(bool))


def test_unique_baselines_and_row_map():
    from tricolour_amd import packing
    d, _ = load_golden("G8_packing.npz")
    ubl = packing.unique_baselines(d["ant1"], d["ant2"])
    assert np.array_equal(ubl, d["ubl"])
    row_bl, row_bl_pack, row_time = packing.row_map(d["ant1"], d["ant2"], ubl, d["time_inv"], int(d["ntime"]))
    assert np.array_equal(ubl[row_bl, 1], d["ant1"]) and np.array_equal(ubl[row_bl, 2], d["ant2"])
    assert np.array_equal(row_bl, row_bl_pack)      # no duplicate cells in an MS
    assert np.array_equal(row_time, d["time_inv"])
    # a baseline chunk: rows of other baselines are unmapped
    sub = ubl[5:9]
    rb, _, _ = packing.row_map(d["ant1"], d["ant2"], sub, d["time_inv"], int(d["ntime"]))
    inside = np.isin(d["ant1"].astype(np.int64) | (d["ant2"].astype(np.int64) << 32),
                     sub[:, 1].astype(np.int64) | (sub[:, 2].astype(np.int64) << 32))
    assert np.array_equal(rb >= 0, inside)
    # duplicates: the last row wins, as in the reference's serial loop
    a1 = np.array([0, 0, 1], np.int32)
    a2 = np.array([1, 1, 1], np.int32)
    u = packing.unique_baselines(a1, a2)
    _, pack, _ = packing.row_map(a1, a2, u, np.array([0, 0, 0]), 1)
    assert list(pack >= 0) == [False, True, True]


def test_config1_plumbing_capture(oracle):
    """G9 = BASELINE config 1 through the reference's own dask graph
    (pack_data -> dask_wrappers.sum_threshold_flagger -> unpack_data)."""
    d, kw = load_golden("G9_config1_plumbing.npz")
    vw, fw = oracle.pack_data(d["time_inv"], d["ubl"], d["ant1"], d["ant2"], d["data"], d["flag"],
                              int(d["ntime"]))
    assert np.array_equal(vw, d["vis_windows"]) and np.array_equal(fw, d["flag_windows"].astype(bool))
    oracle.set_modes(oracle.POW_POWF, oracle.INTERP_F32)
    try:
        out = oracle.sum_threshold_flagger(vw, fw, **kw)
    finally:
        oracle.set_modes(oracle.POW_SQMUL, oracle.INTERP_F64)
    assert np.array_equal(out, d["out_windows"].astype(bool))
    up = oracle.unpack_data(d["time_inv"], d["ubl"], d["ant1"], d["ant2"], out)
    assert np.array_equal(up, d["unpacked"].astype(bool))


@pytest.mark.gpu
def test_gpu_pack_flag_unpack(gpu, oracle):
    import torch
    from tricolour_amd import packing
    for name in ("G8_packing.npz", "G9_config1_plumbing.npz"):
        d, kw = load_golden(name)
        vw, fw = packing.pack_data(d["time_inv"], d["ubl"], d["ant1"], d["ant2"], d["data"],
                                   d["flag"], int(d["ntime"]))
        ref_v = d["vis_windows"]
        got_v = vw.cpu().numpy()
        assert np.array_equal(np.isnan(got_v.real), np.isnan(ref_v.real))
        ok = ~np.isnan(ref_v.real)
        assert np.array_equal(got_v[ok], ref_v[ok])
        assert np.array_equal(fw.cpu().numpy(), d["flag_windows"].astype(bool))
        if kw:
            out = gpu.sum_threshold_flagger(vw, fw, **kw)
            assert np.array_equal(out.cpu().numpy(), d["out_windows"].astype(bool))
        else:
            out = torch.from_numpy(d["out_windows"].astype(bool)).cuda()
        up = packing.unpack_data(d["ant1"], d["ant2"], d["time_inv"], d["ubl"], out)
        assert np.array_equal(up.cpu().numpy(), d["unpacked"].astype(bool))
        # app.py:479-480: flag entire visibility if any correlation is flagged
        eq = packing.unpack_data(d["ant1"], d["ant2"], d["time_inv"], d["ubl"], out, equalize_corr=True)
        ref = d["unpacked"].astype(bool)
        exp_eq = np.broadcast_to(ref.sum(axis=2, keepdims=True) > 0, ref.shape)
        assert np.array_equal(eq.cpu().numpy(), exp_eq)
        # unpack of a baseline chunk: rows outside stay 0 (packing.py:396-398)
        sub = d["ubl"][2:4].copy()
        up2 = packing.unpack_data(d["ant1"], d["ant2"], d["time_inv"], sub, out[2:4])
        exp2 = oracle.unpack_data(d["time_inv"], sub, d["ant1"], d["ant2"], out[2:4].cpu().numpy())
        assert np.array_equal(up2.cpu().numpy(), exp2)
