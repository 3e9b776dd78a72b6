import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The suite binds sdr-iq-visualizer_amd/lib/libsdrk.so; build it if the tree is fresh (hipcc
    # cross-compiles gfx950 without a GPU).  __graft_entry__.build() does the same.
    lib = os.path.join(REPO, "sdr-iq-visualizer_amd", "lib", "libsdrk.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(REPO, "sdr-iq-visualizer_amd", "csrc"), "-j", "8"], check=False)


@pytest.fixture(scope="session")
def golden():
    """name -> NpzFile of the fixtures captured from the reference (oracle/make_golden.py)."""
    return {
        name[:-4]: np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        for name in sorted(os.listdir(GOLDEN))
        if name.endswith(".npz")
    }
