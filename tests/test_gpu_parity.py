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


@pytest.mark.parametrize("path", ["general", "classB"])
def test_l2_alternative_paths(gpu_engine, path, monkeypatch):
    """the general on-the-fly L2 kernel and the wide LDS class must agree with the default class-A path (and the oracle)"""
    monkeypatch.setenv("ANI_L2_PATH", path)
    gpu_engine.reset_counters()
    pc.case_synthetic_cluster(gpu_engine, 60000)
    pc.case_tandem_repeats(gpu_engine)
    c = gpu_engine.counters()
    if path == "general":
        assert c["l2FastCandidates"] == 0 and c["l2SlowCandidates"] > 0
    else:
        assert c["l2SlowCandidates"] == 0 and c["l2FastCandidates"] > 0


def test_full_size_properties(gpu_engine):
    """BASELINE-size genomes (5 Mbp), device-resident 2-bit input: size-independent properties + oracle on two pairs."""
    import torch
    import orc
    from fastani_amd.api import DeviceGenomes, Sketch
    n, L = 24, 5_000_000
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    gpu_engine.synth_packed(99, 0, n, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    p = gpu_engine.params()
    sk = Sketch(gpu_engine, p, dg)
    mins = sk.minimizers()
    # sortedness of the position-ordered index and window-position density (2/(w+1) of the positions)
    key = mins["seqId"].astype(np.int64) << 32 | mins["wpos"].astype(np.int64)
    assert np.all(np.diff(key) > 0)
    assert abs(len(mins) / (n * L) - 2.0 / (p.windowSize + 1)) < 0.003
    rows = sk.map_cgi_batch(dg, 0)
    rows2 = sk.map_cgi_batch(dg, 0)
    assert np.array_equal(rows, rows2)                              # idempotent
    F = L // 3000
    byq = {(int(r["qryGenomeId"]), int(r["refGenomeId"])): r for r in rows}
    for g in range(n):
        r = byq[(g, g)]                                             # a genome against itself
        assert r["countSeq"] >= 0.98 * F and r["totalQueryFragments"] == F and r["identity"] > 99.99   # (a few fragments share a 2980-bp bin)
    for q in range(20):                                             # cluster 0: identity falls with divergence, unrelated cluster absent
        assert (q, 21) not in byq or byq[(q, 21)]["countSeq"] < 5
    ids = [float(byq[(0, m)]["identity"]) for m in range(1, 17)]
    assert all(a > b for a, b in zip(ids, ids[1:]))
    # sharding invariance: a sketch over a subset gives the same rows for those references
    sub = DeviceGenomes(buf.data_ptr(), n, L, first=0, count=6)
    sk2 = Sketch(gpu_engine, p, sub)
    r2 = sk2.map_cgi_batch(DeviceGenomes(buf.data_ptr(), n, L, first=3, count=2), 3)
    exp = np.array([byq[(int(x["qryGenomeId"]), int(x["refGenomeId"]))] for x in r2], dtype=rows.dtype)
    assert np.array_equal(r2, exp)
    # oracle on two full-size pairs
    g0, g5 = orc.synth_genome(99, 0, L), orc.synth_genome(99, 5, L)
    osk = orc.Sketch([[g0]], 16, p.windowSize)
    maps, tot = osk.map_genome([g5])
    o = osk.compute_cgi(maps, tot, 5)[0]
    r = byq[(5, 0)]
    assert (int(r["countSeq"]), int(r["totalQueryFragments"])) == (int(o["countSeq"]), int(o["totalQueryFragments"]))
    assert r["identity"] == o["identity"]


def test_fuzz(gpu_engine):
    assert pc.fuzz(gpu_engine, seed=11, iterations=150) == 150


def test_small_batches_and_chunks(monkeypatch):
    """sub-batches of a few fragments and L2 chunks of a few candidates (fragments straddling chunk borders, both chunk buffer
    sets, the side stream) must give the same mappings and rows as one big batch"""
    import fastani_amd
    monkeypatch.setenv("ANI_SUBBATCH_FRAGS", "7")
    monkeypatch.setenv("ANI_L2_CHUNK", "13")
    e = fastani_amd.api.Engine(fastani_amd._lib.load(), 0)
    pc.case_synthetic_cluster(e, 60000)
    pc.case_messy(e)
    pc.case_tandem_repeats(e)
    pc.case_sparse_hits(e)
    e.close()
