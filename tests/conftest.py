import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The kernels are built in-tree by __graft_entry__.build(); on a fresh checkout (no libetamd.so yet) build them
    here so that the ABI tests have something to load (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(ROOT, "eigentrajectory_amd", "libetamd.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "eigentrajectory_amd", "csrc"), "-j", str(os.cpu_count() or 4)])


@pytest.fixture(scope="session")
def oracle():
    from oracle import et_oracle
    et_oracle.build()
    return et_oracle


@pytest.fixture
def et_option():
    """et_set_option for the duration of a test (include/eigentraj.h "tuning switches"): et_option("kmeans_packed", 0)."""
    from eigentrajectory_amd import _lib as L
    saved = {}

    def set_(key, value):
        if key not in saved:
            saved[key] = L.get_option(key)
        L.set_option(key, value)

    yield set_
    for key, value in saved.items():
        L.set_option(key, value)
