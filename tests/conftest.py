import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# No library defaults are overridden here: every test runs the configuration a user gets (handles with n_max <= 79 take the
# small-state path, csrc/ekf_small.hip).  Tests that are ABOUT the general kernels at small sizes (states smaller than a slab,
# golden N = 20 streams through the fused cadence ...) say so themselves: set_option("small_state", 0) on their handle, or the
# `general_kernels` fixture for code that creates handles out of the test's reach (the drop-in function, the replay driver).


@pytest.fixture(params=["default_path", "general_kernels"])
def both_paths(request, monkeypatch):
    """Runs a small-state test twice: as shipped (n_max <= 79: ONE workgroup per trajectory, P resident in LDS) and on the
    general kernels at the same size.  The value says which: tests assert the path with `path_ran`."""
    if request.param == "general_kernels":
        monkeypatch.setenv("EKFSLAM_HIP_SMALL_STATE", "0")
    else:
        monkeypatch.delenv("EKFSLAM_HIP_SMALL_STATE", raising=False)
    # the handles the binding keeps between calls (drop-in function, 3-state prototype surface) were created under some
    # earlier test's setting: the next call makes new ones
    from slam_duckietown_amd import ekf_bindings as eb
    for h in list(eb._proto.values()) + ([eb._drop.filt] if eb._drop.filt is not None else []):
        h.close()
    eb._proto.clear()
    eb._drop.filt = None
    return request.param


def path_ran(f, which):
    """The handle `f` ran (so far) on the path the `both_paths` parameter `which` names: small-state launches counted where
    the shipped defaults send a handle of this size there (n_max <= 79; <= 131 for banks of at least 128), none otherwise."""
    import slam_duckietown_amd as sd
    small = sd.load_library().ekf_debug_small_launches(f._h)
    takes_small = which == "default_path" and (f.n_max <= 79 or (f.batch >= 128 and f.n_max <= 131))
    return small > 0 if takes_small else small == 0


@pytest.fixture
def general_kernels(monkeypatch):
    """New handles of this test take the general kernels whatever their size (the library reads the variable at creation)."""
    monkeypatch.setenv("EKFSLAM_HIP_SMALL_STATE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
