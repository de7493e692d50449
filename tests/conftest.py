import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): build it once if the tree is fresh
    (hipcc cross-compiles gfx950 without a GPU).  A failed build is not hidden -- the tests that
    load the library then fail with NativeLibraryMissing."""
    lib = os.path.join(ROOT, "centerpoly_amd", "csrc", "libcenterpoly_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "centerpoly_amd", "csrc")], check=False,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    return load
