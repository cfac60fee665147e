"""CPU-side checks of host logic and kernel control flow: the UNMODIFIED product sources compiled against the fiber-based
HIP stand-in in tests/emu (test infrastructure only), compared with the oracle.  The parity tests proper are the `gpu`
ones in test_gpu_parity.py, which run the same cases through the hipcc-built library on a real MI355X."""
import numpy as np
import pytest

import parity_cases as pc


@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda c: c.__name__)
def test_case(emu_engine, case):
    if case is pc.case_synthetic_cluster:
        case(emu_engine, 45000)
    else:
        case(emu_engine)


def test_device_synth(emu_engine):
    def alloc(nbytes):
        a = np.zeros(nbytes // 4 + 4, dtype=np.uint32)
        return a, a.ctypes.data
    pc.case_device_synth(emu_engine, alloc)


def test_fuzz(emu_engine):
    assert pc.fuzz(emu_engine, seed=7, iterations=12) == 12
