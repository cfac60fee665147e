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
    monkeypatch.setenv("ANI_TEST_L2_PATH", path)
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


def test_evolved_full_size(gpu_engine):
    """24 genomes of ~5 Mbp in two families whose members differ by substitutions, indels, inversions, segmental duplications and
    translocations, some as several shuffled contigs, some with a plasmid-like diverged copy (golden_cases.evolved_family):
    the minimizer stream of all 24 and the CGI rows of six queries against all 24, bit-exact against the oracle"""
    import golden_cases
    import orc
    from fastani_amd.api import Sketch
    fam = golden_cases.evolved_family(21, 5_000_000, 12) + golden_cases.evolved_family(22, 5_000_000, 12)
    p = gpu_engine.params()
    sk = Sketch(gpu_engine, p, fam)
    osk = orc.Sketch(fam, 16, p.windowSize)
    assert np.array_equal(sk.minimizers(), osk.minimizers())
    rows = sk.map_cgi_batch(fam, 0)
    for q in (0, 2, 5, 13, 16, 22):
        maps, tot = osk.map_genome(fam[q])
        exp = osk.compute_cgi(maps, tot, q)
        got = rows[rows["qryGenomeId"] == q]
        assert len(exp) >= 12 and np.array_equal(got, exp), q
    # one of them through the per-mapping interface as well (all 11 fields of every mapping record)
    m, tot = sk.map_query(fam[5])
    om, otot = osk.map_genome(fam[5])
    assert tot == otot and np.array_equal(m, om)


def test_fuzz(gpu_engine):
    assert pc.fuzz(gpu_engine, seed=11, iterations=150) == 150


def test_small_batches_and_chunks(monkeypatch):
    """sub-batches of a few fragments and L2 chunks of a few candidates (fragments straddling chunk borders, both chunk buffer
    sets, the side stream) must give the same mappings and rows as one big batch"""
    import fastani_amd
    monkeypatch.setenv("ANI_SUBBATCH_FRAGS", "7")
    monkeypatch.setenv("ANI_TEST_L2_CHUNK", "13")
    e = fastani_amd.api.Engine(fastani_amd._lib.load(), 0)
    pc.case_synthetic_cluster(e, 60000)
    pc.case_messy(e)
    pc.case_tandem_repeats(e)
    pc.case_sparse_hits(e)
    e.close()


def test_sharded_sketch_records_on_device(gpu_engine):
    """the N > 1 staging of bench.py in one process: each "rank" sketches its share of the references into 12-byte records
    with global seqIds, the shards are concatenated on the device (what the RCCL all-gather delivers), the index is built from
    the records; minimizers and ANI rows must equal the directly built sketch and the oracle"""
    import torch
    import orc
    from fastani_amd.api import HostGenomes, Sketch
    e = gpu_engine
    p = e.params()
    ids = [0, 2, 7, 20, 21]
    genomes = [[orc.synth_genome(5, g, 40000)] if g != 7 else [orc.synth_genome(5, g, 25000), orc.synth_genome(5, 8, 15000)] for g in ids]
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    world = 3
    parts, counts = [], []
    for rank in range(world):
        lo, hi = (len(genomes) * rank) // world, (len(genomes) * (rank + 1)) // world
        ptr, n = e.sketch_records(p, HostGenomes(genomes[lo:hi]), int(gcs[lo]))
        t = torch.zeros(max(n, 1) * 3, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        if n:
            e.device_copy(t.data_ptr(), ptr, n * 12)
            e.device_free(ptr)
        parts.append(t[:n * 3]); counts.append(n)
    rec = torch.cat(parts).contiguous()
    torch.cuda.synchronize()
    sk = Sketch(e, p, records=(rec.data_ptr(), int(sum(counts)), contig_len, gcs))
    direct = Sketch(e, p, genomes)
    assert np.array_equal(sk.minimizers(), direct.minimizers())
    rows = sk.map_cgi_batch(genomes, 0)
    assert np.array_equal(rows, direct.map_cgi_batch(genomes, 0))
    osk = orc.Sketch(genomes, 16, p.windowSize)
    exp = []
    for qi, g in enumerate(genomes):
        maps, tot = osk.map_genome(g)
        exp.append(osk.compute_cgi(maps, tot, qi))
    assert np.array_equal(rows, np.concatenate(exp))


def _engine_with(monkeypatch, **env):
    from fastani_amd import _lib, api
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return api.Engine(_lib.load(), 0)


def test_chunked_reference_set(monkeypatch):
    """>= 3 index chunks on small inputs (ANI_MAX_INDEX_MINIMIZERS): same minimizers, mappings and rows as the oracle"""
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=8000)
    pc.case_chunked(e)
    assert pc.fuzz(e, seed=23, iterations=25) == 25
    e.close()


def test_chunked_with_small_batches(monkeypatch):
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=5000, ANI_SUBBATCH_FRAGS=9, ANI_TEST_L2_CHUNK=11)
    pc.case_synthetic_cluster(e, 60000)
    pc.case_sparse_hits(e)
    pc.case_self(e)
    e.close()


def test_chunked_full_size_equals_unchunked(gpu_engine, monkeypatch):
    """24 x 5 Mbp, 5 chunks against one index: identical rows (and the device-resident input path)"""
    import torch
    from fastani_amd.api import DeviceGenomes, Sketch
    n, L = 24, 5_000_000
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    gpu_engine.synth_packed(99, 0, n, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    p = gpu_engine.params()
    sk = Sketch(gpu_engine, p, dg)
    rows = sk.map_cgi_batch(dg, 0)
    st = sk.stats()
    sk.close()
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=2_000_000)
    sk2 = Sketch(e, p, dg)
    assert len(sk2.chunks()) >= 5
    assert sk2.stats() == st
    assert np.array_equal(sk2.map_cgi_batch(dg, 0), rows)
    e.close()


def test_tiny_batches_do_not_inflate_the_candidate_pool(monkeypatch):
    """a context that mapped single fragments with thousands of candidates (low-complexity input) and then gets a large batch: the
    running estimate of candidates per fragment once took "fullest pool stripe x 64 / fragments" of the one-fragment batches at face
    value and failed the 40 000-fragment batch with a bogus 2^31-candidates error"""
    import torch
    from fastani_amd.api import DeviceGenomes, Sketch
    e = _engine_with(monkeypatch)
    pc.case_low_complexity_big(e)
    pc.case_low_complexity(e)
    n, L = 24, 5_000_000
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    e.synth_packed(99, 0, n, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    sk = Sketch(e, e.params(), dg)
    rows = sk.map_cgi_batch(dg, 0)
    assert len(rows) >= n
    sk.close()
    e.close()


def test_repeated_runs_are_identical(monkeypatch):
    """24 x 5 Mbp mapped 25 times with the L1 noise filter forced on for every fragment (its compaction hands the sort a different
    permutation every time): rows, candidate and window counters must not move.  Regression test for a barrier without its LDS
    wait at the back edge of the sort's pass loop (a mis-sorted fragment in about 10^6, 2-5 % of such runs differed)."""
    import torch
    from fastani_amd.api import DeviceGenomes, Sketch
    e = _engine_with(monkeypatch, ANI_TEST_L1_FILTER_MIN=0)
    n, L = 24, 5_000_000
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    e.synth_packed(99, 0, n, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    sk = Sketch(e, e.params(), dg)
    keys = ("seedHits", "l1Candidates", "l2WindowEntries", "l2Steps", "cgiRows")
    e.reset_counters()
    first = sk.map_cgi_batch(dg, 0)
    c0 = {k: e.counters()[k] for k in keys}
    for i in range(25):
        e.reset_counters()
        rows = sk.map_cgi_batch(dg, 0)
        assert {k: e.counters()[k] for k in keys} == c0, i
        assert np.array_equal(rows, first), i
    sk.close()
    e.close()


def test_streamed_reference_set(monkeypatch, tmp_path):
    """reference set streamed chunk by chunk (one resident chunk), then with two resident chunks and tiny query sub-batches"""
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=7000, ANI_MAX_RESIDENT_CHUNKS=1)
    pc.case_streamed(e, tmp_path)
    assert pc.fuzz(e, seed=31, iterations=20) == 20
    e.close()
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=7000, ANI_MAX_RESIDENT_CHUNKS=2, ANI_SUBBATCH_FRAGS=11, ANI_TEST_L2_CHUNK=17)
    pc.case_streamed(e)
    e.close()


def test_streamed_full_size_equals_resident(gpu_engine, monkeypatch):
    """24 x 5 Mbp: 5 chunks, one resident at a time, against the single resident index: identical rows, exact unique count"""
    import torch
    from fastani_amd.api import DeviceGenomes, Sketch
    n, L = 24, 5_000_000
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    gpu_engine.synth_packed(99, 0, n, L, buf.data_ptr())
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    p = gpu_engine.params()
    sk = Sketch(gpu_engine, p, dg)
    rows = sk.map_cgi_batch(dg, 0)
    st = sk.stats()
    sk.close()
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=2_000_000, ANI_MAX_RESIDENT_CHUNKS=1)
    sk2 = Sketch(e, p, dg)
    assert len(sk2.chunks()) >= 5 and sk2.residency()["streaming"]
    assert np.array_equal(sk2.map_cgi_batch(dg, 0), rows)
    assert sk2.stats() == st
    ptr, nrec, frags = e.sketch_records_self(p, dg, 0)
    e.device_free(ptr)
    assert np.array_equal(sk2.map_cgi_fragset(frags, 0), rows)
    frags.close()
    e.close()


def test_fragset_wire_format(gpu_engine):
    import torch

    def alloc(nbytes):
        t = torch.zeros(nbytes // 4 + 4, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        return t, t.data_ptr()
    pc.case_fragset_wire(gpu_engine, alloc)


def test_sketch_file(gpu_engine, tmp_path):
    pc.case_sketch_file(gpu_engine, tmp_path)


def test_sketch_file_chunked(monkeypatch, tmp_path):
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=6000)
    pc.case_sketch_file(e, tmp_path)
    e.close()


def test_big_l1_groups(monkeypatch):
    """the batched global-memory L1 path cut into many small groups (few fragments, few hits per group)"""
    e = _engine_with(monkeypatch, ANI_TEST_L1_BIG_GROUP_HITS=30000, ANI_TEST_L1_BIG_GROUP_FRAGS=3)
    pc.case_species_dense(e, copies=52, n=9000)
    pc.case_low_complexity_big(e)
    e.close()
    # ANI_TEST_L1_LDS_MAX=0: EVERY fragment takes the batched path (the LDS classes are an optimisation of it, not a different algorithm)
    e = _engine_with(monkeypatch, ANI_TEST_L1_LDS_MAX=0, ANI_TEST_L1_BIG_GROUP_HITS=200000)
    e.reset_counters()
    pc.case_synthetic_cluster(e)
    pc.case_messy(e)
    pc.case_tandem_repeats(e)
    pc.case_evolved(e)
    pc.case_sparse_hits(e)
    assert pc.fuzz(e, seed=53, iterations=30) == 30
    assert e.counters()["l1BigFragments"] > 0
    e.close()


def test_limits(gpu_engine):
    pc.case_limits(gpu_engine)


def test_l2_code_overflow_halves_the_chunk(monkeypatch):
    e = _engine_with(monkeypatch, ANI_TEST_L2_CODE_LIMIT=6000, ANI_TEST_L2_CHUNK=64)
    e.reset_counters()
    pc.case_synthetic_cluster(e, 60000)
    assert e.counters()["l2ChunkHalvings"] > 0
    e.close()


@pytest.mark.gpu
def test_candidate_pool_retry_keeps_the_overflow_marker(monkeypatch):
    """a batch that both overflows the L1 candidate pool and holds a fragment beyond the seed-hit limit"""
    e1 = _engine_with(monkeypatch, ANI_TEST_CAND_POOL_MIN=1)
    e2 = _engine_with(monkeypatch, ANI_TEST_CAND_POOL_MIN=1, ANI_TEST_L1_HIT_LIMIT=600)
    pc.case_cand_pool_retry(e1, e2)
    e1.close(); e2.close()


@pytest.mark.gpu
def test_l1_tiny_path_off(monkeypatch):
    """fragments with <= 64 seed hits are finished by one wave (l1.hpp: l1_tiny) — nearly every fragment of the small cases; with
    ANI_TEST_L1_TINY=0 they take the workgroup path like the others: same candidates, same rows"""
    e = _engine_with(monkeypatch, ANI_TEST_L1_TINY=0)
    pc.case_synthetic_cluster(e, 30000)
    pc.case_sparse_hits(e)
    pc.case_tandem_repeats(e)
    pc.case_l1_class_overflow(e)
    e.close()


@pytest.mark.gpu
def test_l1_lds_cap_below_the_wave_class(monkeypatch):
    """with ANI_TEST_L1_LDS_MAX between 1 and 255 a fragment with ldsHitCap < H <= 256 seed hits belongs to the batched path
    alone — k_l1_tiny must use k_l1_probe's class predicate, or the fragment's candidates enter the pool twice"""
    cands = []
    for env in ({}, dict(ANI_TEST_L1_LDS_MAX=100, ANI_TEST_L1_BIG_GROUP_HITS=200000)):
        e = _engine_with(monkeypatch, **env)
        e.reset_counters()
        pc.case_synthetic_cluster(e, 30000)
        pc.case_tandem_repeats(e)
        # distant relatives only (12-15 % divergence): a few dozen seed hits per fragment, the wave kernel's class under either setting
        a0 = pc.rng_genome(31, 24000)
        far = [[a0], [pc.mutate(a0, 0.12, 32)]]
        pp, sk, osk = pc.check_sketch(e, far)
        pc.check_queries(e, pp, sk, osk, [[pc.mutate(a0, 0.15, 33)]])
        c = e.counters()
        cands.append(c["l1Candidates"])
        if env:
            assert c["l1BigFragments"] > 0 and c["l1TinyFragments"] > 0
        e.close()
    assert cands[0] == cands[1], cands            # every candidate emitted once



@pytest.mark.gpu
def test_result_rows_collected_on_host_threads(monkeypatch):
    """the dense result table of a sub-batch is turned into rows by the host pool (engine_map.hip: collect_rows: rows per query,
    then every query's rows at their place); ANI_TEST_HOST_PAR_MIN_WORK=0 sends small tables through the pool too — same rows, same order,
    also with more threads than queries and with several kept sets per call"""
    import torch
    def alloc(nbytes):
        t = torch.zeros(nbytes // 4 + 4, dtype=torch.int32, device="cuda:0")
        return t, t.data_ptr()
    for threads in (3, 64):
        e = _engine_with(monkeypatch, ANI_TEST_HOST_PAR_MIN_WORK=0, ANI_HOST_THREADS=threads)
        pc.case_self(e, combos=((16, 3000),))
        pc.case_fragset_wire(e, alloc)
        pc.case_species_dense(e)
        e.close()
    monkeypatch.delenv("ANI_TEST_HOST_PAR_MIN_WORK"); monkeypatch.delenv("ANI_HOST_THREADS")



@pytest.mark.gpu
def test_same_hash_links_rerun(monkeypatch):
    """the list of same-hash links (index.hpp: DupLinks) is sized from a guess; a repetitive reference (tandem repeats, one k-mer on
    thousands of contigs) holds more near-duplicate pairs than that and the links kernel runs again with room for all"""
    e = _engine_with(monkeypatch, ANI_TEST_DUP_PAIR_CAP=3)
    pc.case_tandem_repeats(e)
    pc.case_low_complexity(e)
    pc.case_gap_counter_overflow(e)
    e.close()


@pytest.mark.gpu
def test_l2_without_the_side_stream(monkeypatch):
    """since round 4 the L2 simulation of a chunk runs on the side stream beside the next chunk's ranges / codes kernels; ANI_TEST_L2_OVERLAP=0
    is the serial order of rounds 1-3: same results (tiny L2 chunks: many hand-overs between the two streams)"""
    for ov in ("0", "1"):
        e = _engine_with(monkeypatch, ANI_TEST_L2_OVERLAP=ov, ANI_TEST_L2_CHUNK=97)
        pc.case_synthetic_cluster(e, 60000)
        pc.case_tandem_repeats(e)
        e.close()


@pytest.mark.gpu
def test_merged_fragment_sets_on_a_streamed_reference_set(monkeypatch):
    """a merged set (ani_fragset_unpack_merged: non-consecutive query ids) mapped against a reference set that is streamed chunk by
    chunk: the rows of a sub-batch come chunk by chunk and are put back into (query, reference) order through the id table"""
    import torch

    def alloc(nbytes):
        t = torch.zeros(nbytes // 4 + 4, dtype=torch.int32, device="cuda:0")
        return t, t.data_ptr()
    e = _engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=4000, ANI_MAX_RESIDENT_CHUNKS=1, ANI_SUBBATCH_FRAGS=9)
    pc.case_fragset_wire(e, alloc)
    e.close()
