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


def test_thresholds_vs_scipy_up_to_2048(emu_engine):
    """minimumHits(s) (computeMap.hpp:301) and the keep threshold minShared(s) (:384) for every sketch size the L1 LDS classes accept
    (s <= 2048; the reference's own LUT dump covers s <= 400), with the binomial tail taken from SciPy (regularised incomplete beta,
    like GSL's gsl_cdf_binomial_Q) instead of the product's own summation — the only third-party arithmetic on the path
    (map_stats.hpp:96).  Float types follow map_stats.hpp:44-167 expression by expression."""
    from scipy.stats import binom
    f32 = np.float32
    k, cutoff = 16, f32(80.0)
    q2 = f32((1.0 - float(f32(0.9))) / 2)

    def j2md(j):
        if j == 0:
            return f32(1.0)
        if j == 1:
            return f32(0.0)
        return f32((-1.0 / k) * np.log(2.0 * float(j) / (1 + float(j))))

    def md2j(d):
        return f32(1.0 / (2.0 * np.exp(float(f32(k) * f32(d))) - 1.0))

    def lower_bound(d, s):
        jd = md2j(d)
        x0 = max(int(np.ceil(float(f32(s) * jd))), 1)
        x = x0
        while x <= s:                                    # vectorised search: 64 candidates per SciPy call
            xs = np.arange(x, min(s, x + 63) + 1)
            tails = binom.sf(xs - 1, s, float(jd))
            hit = np.nonzero(tails < float(q2))[0]
            if len(hit):
                x = int(xs[hit[0]]) - 1
                break
            x = int(xs[-1]) + 1
        return j2md(f32(x) / f32(s))

    def upper_identity(x, s):
        d = j2md(f32(1.0 * x / s))
        return f32(100.0 * (1.0 - float(lower_bound(d, s))))

    L = emu_engine.lib
    a, b = C.c_float(), C.c_float()
    for s in list(range(1, 2049)):
        first = int(np.ceil(1.0 * s * float(md2j(f32(1.0 - float(cutoff) / 100.0)))))
        relaxed = first
        for i in range(first, -1, -1):
            if upper_identity(i, s) >= cutoff:
                relaxed = i
            else:
                break
        assert L.ani_min_hits_relaxed(s, k, float(cutoff)) == relaxed, s
        # keep threshold: smallest shared whose upper bound passes; the product's value through ani_identity
        ms = next(x for x in range(0, s + 2) if x > s or upper_identity(x, s) >= cutoff)
        got = s + 1
        for x in range(0, s + 1):
            assert L.ani_identity(x, s, k, C.byref(a), C.byref(b)) == 0
            if np.float32(b.value) >= cutoff:
                got = x
                break
        assert got == ms, (s, got, ms)
