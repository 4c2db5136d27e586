import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "thesis-fmri-reconstruction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "selfcheck: compares the engine with itself (launch modes, ranks), not with the "
                                       "oracle / goldens -- collected after every parity test")


def pytest_collection_modifyitems(config, items):
    """Every HIP-vs-oracle / HIP-vs-golden test runs before any engine-vs-engine comparison, wherever it is defined: a
    failing self-comparison can then never keep ``pytest -x`` from reaching a parity test.  (Stable sort: the order
    inside the two classes is the collection order.)"""
    items.sort(key=lambda it: 1 if it.get_closest_marker("selfcheck") else 0)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def deterministic():
    """The engine's deterministic-reduction mode (fmri_hip.ops.set_deterministic) for the duration of one test."""
    from fmri_hip import ops
    was = ops.set_deterministic(True)
    try:
        yield
    finally:
        ops.set_deterministic(was)
