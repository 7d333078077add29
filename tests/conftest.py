import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must FAIL, not skip, when the HIP path is unavailable on a GPU
    # box; on a CPU-only box they are deselected by `-m "not gpu"`.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def gen_pair(seed, s1, s2, shift=0.5):
    """The survey's input convention: x1 then x2 from one default_rng(seed)."""
    rng = np.random.default_rng(seed)
    a = rng.random(s1, dtype=np.float32) - np.float32(shift)
    b = rng.random(s2, dtype=np.float32) - np.float32(shift)
    return a, b
