import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference's bundled 640x480 pair and its RTL x-Sobel output (tools/make_golden_fixtures.py)."""
    g = np.load(ROOT / "tests" / "golden" / "ref_pair_640x480.npz")
    return {k: g[k] for k in g.files}


@pytest.fixture(scope="session")
def oracle():
    import sbm_oracle

    sbm_oracle.lib()
    return sbm_oracle


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory `u96-slam_amd/`, imported as u96_slam_amd)."""
    import _pkg

    return _pkg.load()
