import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """Kernel-vs-oracle parity first, launcher / subprocess tests last: under ``-x`` a failure of the multi-rank launcher
    (tests/test_bench_launch.py spawns gloo ranks and torch.distributed.run) must never again hide the parity tests."""
    last = ("test_bench_launch.py",)
    items.sort(key=lambda it: 1 if os.path.basename(str(it.fspath)) in last else 0)     # stable: keeps the rest in order


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class _Golden:
        def __init__(self):
            self._cache = {}

        def __call__(self, name):
            if name not in self._cache:
                self._cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
            return self._cache[name]
    return _Golden()


@pytest.fixture
def kernel_policy():
    """``kernel_policy("no_fixed", ...)`` sets the context's kernel-selection policy (fx_ctx_set_policy) for the rest
    of the test; the default policy is restored afterwards."""
    from fiat_amd import runtime
    ctx = runtime.Context.get()
    yield ctx.set_policy
    ctx.set_policy()
