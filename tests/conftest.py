import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The small-state path (csrc/ekf_small.hip) takes over every handle with n_max <= 79 by default.  The tests written for the
# general kernels deliberately run them at small sizes too (states smaller than a slab, golden N = 20 streams through the fused
# cadence, pass kernels on 64 landmarks ...): they keep doing so -- the library reads this variable when a handle is created --
# and tests/test_gpu_small_state.py switches the small-state path on explicitly (set_option / monkeypatch) for its own cases.
os.environ.setdefault("EKFSLAM_HIP_SMALL_STATE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
