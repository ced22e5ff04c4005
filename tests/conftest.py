import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), numba-canonical modes."""
    from oracle import oracle as O
    O.build()
    O.set_modes(O.POW_SQMUL, O.INTERP_F64)
    return O


@pytest.fixture(scope="session")
def gpu():
    if not has_gpu():
        pytest.skip("no GPU visible")
    import tricolour_amd
    return tricolour_amd


def load_golden(name):
    import numpy as np
    d = np.load(os.path.join(GOLDEN, name))
    kw = {k[3:]: d[k].tolist() for k in d.files if k.startswith("kw_")}
    return d, kw


GOLDEN_CASES = ["G1_defaults.npz", "G2_stage1.npz", "G2b_radius32.npz", "G3_broad.npz",
                "G4_preflagged.npz", "G5_complex_nan.npz", "G6_all_flagged.npz",
                "G7_clipping.npz", "G10_average2.npz", "G14_wide_average3.npz", "G14b_complex128_average3.npz"]
