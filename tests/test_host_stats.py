"""Host-side scalars of the product (fastani_amd/csrc/host/stats.hpp through the C-ABI) against the reference's own LUTs
(tests/golden/stats_k16.npz, dumped from skch::Stat by oracle/ref_dump.cpp): every (s, shared) for s <= 400, bit-exact floats."""
import ctypes as C
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_min_hits_and_identity_tables(emu_engine):
    L = emu_engine.lib
    g = np.load(os.path.join(GOLD, "stats_k16.npz"))
    maxS = int(g["maxS"])
    o = 0
    a, b = C.c_float(), C.c_float()
    for s in range(1, maxS + 1):
        assert L.ani_min_hits_relaxed(s, 16, 80.0) == int(g["minHits"][s - 1]), s
        if s > 48 and s % 9:                 # every s up to 48, then every 9th, with a stride over x
            o += s + 1
            continue
        step = 1 if s <= 48 else 5
        for x in list(range(0, s + 1, step)) + [s]:
            assert L.ani_identity(x, s, 16, C.byref(a), C.byref(b)) == 0
            assert np.float32(a.value) == g["identity"][o + x] and np.float32(b.value) == g["upper"][o + x], (s, x)
        o += s + 1


def test_window_sizes(emu_engine):
    for key, w in json.load(open(os.path.join(GOLD, "windows.json"))).items():
        k, L = [int(v) for v in key.split(",")]
        assert emu_engine.lib.ani_recommended_window(k, L) == w
    p = emu_engine.params()
    assert (p.kmerSize, p.windowSize, p.fragLen, p.percentageIdentity) == (16, 24, 3000, 80.0)


def test_keep_threshold_is_monotone():
    """the device filter uses minShared[s] = first `shared` whose upper bound passes; valid only if the bound is monotone"""
    g = np.load(os.path.join(GOLD, "stats_k16.npz"))
    o = 0
    for s in range(1, int(g["maxS"]) + 1):
        ub = g["upper"][o:o + s + 1]
        assert np.all(np.diff(ub) >= 0), s
        o += s + 1
