import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# OpenMP teams (oracle, host layer, torch) sized to the CPUs this job may really use, and sleeping rather than spinning at
# barriers: on a GPU box that shows 128 cores but grants a share of them, 128 spinning threads inside the CFS quota turned a
# 2-second oracle call into minutes (set before any OpenMP runtime is loaded).
import oracle_c  # noqa: E402
os.environ.setdefault("OMP_NUM_THREADS", str(min(oracle_c.effective_cpus(), 64)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


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


@pytest.fixture
def c_generated_weights():
    """(spec, cfg) -> name -> numpy dict of a synthetic model from the C weight source (glc_weights_load, the code
    create_ort_session uses), WITHOUT copying: the 0.4-1.5 B-parameter configs take minutes through the numpy generator.
    The mappings are released when the test ends."""
    import ctypes as C
    import numpy as np
    from gliclass.c_amd import _lib
    from gliclass.c_amd.weights import tensor_specs
    handles = []

    def get(spec, cfg):
        M = _lib.model()
        w = _lib.Weights()
        assert M.glc_weights_load(spec.encode(), C.byref(w)) == 0
        handles.append(w)
        specs = tensor_specs(cfg)
        assert w.n_tensors == len(specs)
        return {name: np.ctypeslib.as_array(w.tensors[i], shape=tuple(shape)) for i, (name, shape, _, _) in enumerate(specs)}
    yield get
    for w in handles:
        _lib.model().glc_weights_free(C.byref(w))
