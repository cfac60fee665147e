"""Parity tests proper: the hipcc-built library on a real MI355X, through the C-ABI, against the CPU oracle."""
import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda c: c.__name__)
def test_case(gpu_engine, case):
    case(gpu_engine)


def test_device_synth(gpu_engine):
    import torch

    def alloc(nbytes):
        t = torch.zeros(nbytes // 4 + 4, dtype=torch.int32, device="cuda:0")
        return t, t.data_ptr()
    pc.case_device_synth(gpu_engine, alloc)
