import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (builds oracle/liboracle.so on first use)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "oracle_golden.npz"))


@pytest.fixture(scope="session")
def car_xyz9():
    return np.load(os.path.join(GOLDEN, "car_tris.npy"))


@pytest.fixture(scope="session")
def phoenix_xyz9():
    return np.load(os.path.join(GOLDEN, "phoenix_tris.npy"))


@pytest.fixture(scope="session")
def rt():
    import ray_tracer_amd
    return ray_tracer_amd


@pytest.fixture()
def core(rt):
    """A fresh trace-core context on GPU 0; fails (does not skip) when the library cannot run."""
    c = rt.TraceCore(0)
    yield c
    c.close()
