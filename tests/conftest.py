import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_engine():
    """The product library on cuda:0.  Fails (does not skip) when the HIP extension is missing."""
    import fastani_amd
    return fastani_amd.engine(0)


@pytest.fixture(scope="session")
def emu_engine():
    """TEST-ONLY CPU build of the same sources (tests/emu) for host-logic/control-flow checks without a GPU."""
    import ctypes
    import subprocess
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir])
    from fastani_amd.api import Engine
    return Engine(ctypes.CDLL(os.path.join(emu_dir, "libfastani_emu.so")), 0)
