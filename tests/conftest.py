import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_WCACHE = {}


@pytest.fixture(scope="session")
def weights_for():
    """(config name, seed) -> (cfg, tensors) with caching (small/base take seconds to hash)."""
    from gliclass.c_amd.config import CONFIGS
    from gliclass.c_amd import weights

    def get(name, seed=42):
        if (name, seed) not in _WCACHE:
            _WCACHE[(name, seed)] = (CONFIGS[name], weights.make_weights(CONFIGS[name], seed))
        return _WCACHE[(name, seed)]
    return get
