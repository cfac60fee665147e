"""Kernel-level checks of the index build on the CPU stand-in for the HIP runtime (tests/emu): a kernel against the plain
definition of what it computes, on inputs the end-to-end parity cases reach only by chance.  TEST INFRASTRUCTURE ONLY."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def window_links_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kchk") / "window_links_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "tests/emu/include"),
           "-I", os.path.join(ROOT, "fastani_amd/csrc"), "-o", exe, os.path.join(ROOT, "tests/kernel_checks/window_links_check.cpp"),
           os.path.join(ROOT, "tests/emu/hip_emu.cpp"), "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_window_links_against_their_definition(window_links_check, seed):
    """k_index_window_links (four consecutive entries per thread, answers advanced from the previous entry's) = the definition
    of A / B / the same-position bit, for every entry: single-entry contigs, dense and sparse stretches, contigs around the
    workgroup size and its halo, the vector and the scalar load / store paths."""
    r = subprocess.run([window_links_check, str(seed), "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
