"""Deterministic inputs of the golden fixtures (regenerated from seeds; only the reference's OUTPUTS are committed)."""
import numpy as np

import orc


def _rng(seed, n, alphabet=b"ACGT"):
    r = np.random.default_rng(seed)
    return np.frombuffer(alphabet, dtype=np.uint8)[r.integers(0, len(alphabet), n)]


def messy(seed, n):
    g = orc.synth_genome(seed, 3, n)
    g[1000:1300] = ord("N")
    g[5000] = ord("R")
    g[20000:26000] = np.frombuffer(g[20000:26000].tobytes().lower(), dtype=np.uint8)
    g[26000:26010] = ord("n")
    cut = [0, n // 2, n // 2 + 10, n // 2 + 30, n // 2 + 2000, n]
    return [g[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)]


def repeat_at(na, n):
    """tests/gen_tests_data.py:33-55 of the reference in miniature: runs of `na` A's separated by one T"""
    return np.frombuffer((b"A" * na + b"T") * (n // (na + 1) + 1), dtype=np.uint8)[:n]


def cases():
    c = {}
    g = [orc.synth_genome(7, i, 60000) for i in (0, 3, 11, 19, 20)]
    c["cluster"] = ([[x] for x in g], [[g[0]], [g[2]], [g[4]]], 16, 3000)
    m = messy(5, 80000)
    c["messy"] = ([m, [orc.synth_genome(5, 0, 50000)], [_rng(1, 10)]], [m, [orc.synth_genome(5, 0, 50000)], [b"N" * 4000]], 16, 3000)
    g12 = [orc.synth_genome(3, i, 40000) for i in (0, 5, 12)]
    c["k12"] = ([[x] for x in g12], [[g12[1]]], 12, 3000)
    g1k = [orc.synth_genome(9, i, 30000) for i in (0, 7)]
    c["frag1000"] = ([[x] for x in g1k], [[g1k[0]], [g1k[1]]], 16, 1000)
    c["repeats"] = ([[repeat_at(64, 3500)], [repeat_at(8, 3400)], [repeat_at(128, 3300)]], [[repeat_at(128, 9000)], [repeat_at(64, 6000)], [np.full(6000, ord("A"), dtype=np.uint8)]], 16, 3000)
    unit = _rng(4, 700)
    rep = np.tile(unit, 30)
    r2 = np.random.default_rng(8)
    noise = rep.copy()
    mm = r2.random(len(noise)) < 0.03
    noise[mm] = np.frombuffer(b"ACGT", dtype=np.uint8)[r2.integers(0, 4, int(mm.sum()))]
    flank = _rng(6, 20000)
    ref = np.concatenate([flank[:10000], rep, flank[10000:]])
    qry = np.concatenate([flank[:10000], noise, flank[10000:]])
    c["tandem"] = ([[ref], [qry]], [[qry], [ref]], 16, 3000)
    return c
