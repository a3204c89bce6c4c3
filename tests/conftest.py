import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _seeded_torch():
    """Brains built with `Models.X()` take their initial weights from torch's global generator: seed it per test, so that a run's outcome
    (which agents eat, whether a three-agent tester() world survives its 25 ticks) does not depend on the process."""
    import torch
    torch.manual_seed(20260929)
    yield


@pytest.fixture
def hip_option():
    """hip_option(name, value): a process-level option of the HIP library (rl_set_option; handles created afterwards snapshot it),
    put back to what the environment says when the test ends."""
    from reinlife_amd import _lib
    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.set_option(name, value)
    yield set_
    for name in touched:
        _lib.set_option(name, None)
