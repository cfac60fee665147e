"""CPU-side checks of host logic and kernel control flow: the UNMODIFIED product sources compiled against the fiber-based
HIP stand-in in tests/emu (test infrastructure only), compared with the oracle.  The parity tests proper are the `gpu`
ones in test_gpu_parity.py, which run the same cases through the hipcc-built library on a real MI355X."""
import numpy as np
import pytest

import parity_cases as pc


@pytest.mark.parametrize("case", pc.ALL_CASES, ids=lambda c: c.__name__)
def test_case(emu_engine, case):
    # (the CPU stand-in runs every lane as a fiber: the largest case is cut here, the GPU suite runs it at its full size)
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


def test_small_batches_and_chunks(monkeypatch):
    """sub-batches of a few fragments and L2 chunks of a few candidates (fragments straddling chunk borders, both chunk buffer
    sets, the side stream) must give the same mappings and rows as one big batch"""
    import ctypes
    import os
    from fastani_amd.api import Engine
    monkeypatch.setenv("ANI_SUBBATCH_FRAGS", "7")
    monkeypatch.setenv("ANI_TEST_L2_CHUNK", "13")
    e = Engine(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libfastani_emu.so")), 0)
    pc.case_synthetic_cluster(e, 30000)
    pc.case_messy(e)
    pc.case_tandem_repeats(e)
    e.close()


def _emu_engine_with(monkeypatch, **env):
    import ctypes
    import os
    from fastani_amd.api import Engine
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return Engine(ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libfastani_emu.so")), 0)


def test_chunked_reference_set(monkeypatch):
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=8000)
    pc.case_chunked(e)
    assert pc.fuzz(e, seed=23, iterations=3) == 3
    e.close()


def test_chunked_with_small_batches(monkeypatch):
    """index chunks x query sub-batches x L2 chunks, all tiny"""
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=5000, ANI_SUBBATCH_FRAGS=9, ANI_TEST_L2_CHUNK=11)
    pc.case_synthetic_cluster(e, 30000)
    pc.case_sparse_hits(e)
    pc.case_self(e, combos=((16, 3000), (16, 3050)))
    e.close()


def test_streamed_reference_set(monkeypatch, tmp_path):
    """reference set streamed chunk by chunk (one resident chunk), then with two resident chunks and tiny query sub-batches"""
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=4000, ANI_MAX_RESIDENT_CHUNKS=1)
    pc.case_streamed(e, tmp_path, n=18000)
    e.close()
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=4000, ANI_MAX_RESIDENT_CHUNKS=2, ANI_SUBBATCH_FRAGS=11, ANI_TEST_L2_CHUNK=17)
    pc.case_streamed(e, n=12000, light=True)
    e.close()


def test_fragset_wire_format(emu_engine):
    def alloc(nbytes):
        a = np.zeros(nbytes // 4 + 4, dtype=np.uint32)
        return a, a.ctypes.data
    pc.case_fragset_wire(emu_engine, alloc)


def test_sketch_file(emu_engine, tmp_path):
    pc.case_sketch_file(emu_engine, tmp_path)


def test_sketch_file_chunked(monkeypatch, tmp_path):
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=6000)
    pc.case_sketch_file(e, tmp_path)
    e.close()


def test_big_l1_groups(monkeypatch):
    """the batched global-memory L1 path cut into many small groups (few fragments, few hits per group)"""
    e = _emu_engine_with(monkeypatch, ANI_TEST_L1_BIG_GROUP_HITS=30000, ANI_TEST_L1_BIG_GROUP_FRAGS=3)
    pc.case_species_dense(e, copies=52, n=9000)
    pc.case_low_complexity_big(e)
    e.close()
    # ANI_TEST_L1_LDS_MAX=0: EVERY fragment takes the batched path (the LDS classes are an optimisation of it, not a different algorithm)
    e = _emu_engine_with(monkeypatch, ANI_TEST_L1_LDS_MAX=0, ANI_TEST_L1_BIG_GROUP_HITS=200000)
    e.reset_counters()
    pc.case_synthetic_cluster(e, 24000)
    pc.case_tandem_repeats(e)
    assert e.counters()["l1BigFragments"] > 0
    e.close()


def test_limits(emu_engine):
    pc.case_limits(emu_engine)


def test_l2_code_overflow_halves_the_chunk(monkeypatch):
    """a chunk whose 16-bit code stream would pass the offset limit is cut in half and redone (engine_map.hip, L2 chunk loop);
    the limit is lowered through a test knob so that small inputs reach the branch"""
    e = _emu_engine_with(monkeypatch, ANI_TEST_L2_CODE_LIMIT=6000, ANI_TEST_L2_CHUNK=64)
    e.reset_counters()
    pc.case_synthetic_cluster(e, 30000)
    assert e.counters()["l2ChunkHalvings"] > 0
    e.close()


def test_candidate_pool_retry_keeps_the_overflow_marker(monkeypatch):
    """a batch that both overflows the L1 candidate pool and holds a fragment beyond the seed-hit limit"""
    e1 = _emu_engine_with(monkeypatch, ANI_TEST_CAND_POOL_MIN=1)
    e2 = _emu_engine_with(monkeypatch, ANI_TEST_CAND_POOL_MIN=1, ANI_TEST_L1_HIT_LIMIT=600)
    pc.case_cand_pool_retry(e1, e2)
    e1.close(); e2.close()


def test_l1_tiny_path_off(monkeypatch):
    """fragments with <= 64 seed hits are finished by one wave (l1.hpp: l1_tiny) — nearly every fragment of the small cases; with
    ANI_TEST_L1_TINY=0 they take the workgroup path like the others: same candidates, same rows"""
    e = _emu_engine_with(monkeypatch, ANI_TEST_L1_TINY=0)
    pc.case_synthetic_cluster(e, 30000)
    pc.case_sparse_hits(e)
    pc.case_tandem_repeats(e)
    pc.case_l1_class_overflow(e)
    e.close()


def test_l1_lds_cap_below_the_wave_class(monkeypatch):
    """with ANI_TEST_L1_LDS_MAX between 1 and 255 a fragment with ldsHitCap < H <= 256 seed hits belongs to the batched path
    alone — k_l1_tiny must use k_l1_probe's class predicate, or the fragment's candidates enter the pool twice"""
    cands = []
    for env in ({}, dict(ANI_TEST_L1_LDS_MAX=100, ANI_TEST_L1_BIG_GROUP_HITS=200000)):
        e = _emu_engine_with(monkeypatch, **env)
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


def test_sort_is_repeated_once_when_a_look_back_gives_up(monkeypatch):
    """a look-back that gives up (spin bound; a scheduling surprise, not a property of the data) repeats the sort once
    before the call fails — with ANI_TEST_SORT_FAIL_EVERY=2 every sort of the process reports that once and passes when repeated (the
    index build's side-stream form, the fragment order, the batched L1 path's hit sort, the same-hash links)"""
    e = _emu_engine_with(monkeypatch, ANI_TEST_SORT_FAIL_EVERY=2, ANI_TEST_L1_LDS_MAX=0)
    pc.case_synthetic_cluster(e, 24000)
    pc.case_tandem_repeats(e)
    e.close()
    e = _emu_engine_with(monkeypatch, ANI_TEST_SORT_FAIL_EVERY=1)          # never passes: the error surfaces
    from fastani_amd.api import AniError
    with pytest.raises(AniError):
        pc.case_synthetic_cluster(e, 24000)
    e.close()




def test_result_rows_collected_on_host_threads(monkeypatch):
    """the dense result table of a sub-batch is turned into rows by the host pool (engine_map.hip: collect_rows: rows per query,
    then every query's rows at their place); ANI_TEST_HOST_PAR_MIN_WORK=0 sends small tables through the pool too — same rows, same order,
    also with more threads than queries and with several kept sets per call"""
    def alloc(nbytes):
        a = np.zeros(nbytes // 4 + 4, dtype=np.uint32)
        return a, a.ctypes.data
    for threads in (3, 64):
        e = _emu_engine_with(monkeypatch, ANI_TEST_HOST_PAR_MIN_WORK=0, ANI_HOST_THREADS=threads)
        pc.case_self(e, combos=((16, 3000),))
        pc.case_fragset_wire(e, alloc)
        pc.case_species_dense(e)
        e.close()
    monkeypatch.delenv("ANI_TEST_HOST_PAR_MIN_WORK"); monkeypatch.delenv("ANI_HOST_THREADS")


def test_pool_prewarm_is_only_a_hint(emu_engine):
    """ani_pool_prewarm_index: the index arrays for an estimated minimizer count allocated ahead of the build, on another thread as
    the command line does it — while this thread sketches and builds, whose requests wait for the promised blocks; estimates that
    are right, far too small, far too large (refused: more than half the free memory) or zero change nothing but where the memory
    comes from"""
    import threading
    for est in (0, 30, 2400, 40000, 1 << 40):
        t = threading.Thread(target=emu_engine.pool_prewarm_index, args=(est,))
        t.start()
        pc.case_self(emu_engine, combos=((16, 3000),))
        t.join()
    emu_engine.pool_prewarm_index(2400)
    pc.case_synthetic_cluster(emu_engine, 30000)


def test_same_hash_links_rerun(monkeypatch):
    """the list of same-hash links (index.hpp: DupLinks) is sized from a guess; a repetitive reference (tandem repeats, one k-mer on
    thousands of contigs) holds more near-duplicate pairs than that and the links kernel runs again with room for all"""
    e = _emu_engine_with(monkeypatch, ANI_TEST_DUP_PAIR_CAP=3)
    pc.case_tandem_repeats(e)
    pc.case_low_complexity(e)
    pc.case_gap_counter_overflow(e)
    e.close()


def test_merged_fragment_sets_on_a_streamed_reference_set(monkeypatch):
    """a merged set (ani_fragset_unpack_merged: non-consecutive query ids) mapped against a reference set that is streamed chunk by
    chunk: the rows of a sub-batch come chunk by chunk and are put back into (query, reference) order through the id table"""
    def alloc(nbytes):
        a = np.zeros(nbytes // 4 + 4, dtype=np.uint32)
        return a, a.ctypes.data
    e = _emu_engine_with(monkeypatch, ANI_MAX_INDEX_MINIMIZERS=4000, ANI_MAX_RESIDENT_CHUNKS=1, ANI_SUBBATCH_FRAGS=9)
    pc.case_fragset_wire(e, alloc)
    e.close()


def test_host_pool_packing(monkeypatch):
    """the ingest packing loop on the persistent host thread pool (host/engine.hpp: HostPool; only batches of >= 2^22 bases take
    it): several calls on one pool with different thread counts, contigs that are not pure ACGT among them (raw second pass) — the
    minimizer stream must be the oracle's"""
    import orc
    rng = np.random.default_rng(5)
    genomes = []
    for g in range(4):
        s = orc.synth_genome(5, g, 1_100_000).copy()
        if g == 2:
            s[500_000:500_040] = ord("N")                     # one impure contig: copied raw in the second pass
        genomes.append([s, orc.synth_genome(6, g, 30_000)])   # two contigs per genome
    osk = orc.Sketch(genomes, 16, 24)
    want = osk.minimizers()
    for nt in ("3", "7", "2"):
        e = _emu_engine_with(monkeypatch, ANI_HOST_THREADS=nt)
        from fastani_amd.api import Sketch
        for _ in range(2):
            sk = Sketch(e, e.params(), genomes)
            assert np.array_equal(sk.minimizers(), want)
            sk.close()
        e.close()


def test_device_arena_allocator(emu_engine):
    """host/engine.hpp: DevicePool (round 6) through the C-ABI (ani_device_alloc / ani_device_free / ani_pool_stats): extents of segments,
    best fit, split, coalesced with free neighbours on release.  Random sizes on both sides of the small / large limit; every live block
    keeps its own pattern (no two overlap); released memory serves later requests (the segments stay within 1.5 x the largest live
    total); with everything released the extents have coalesced."""
    import ctypes
    e = emu_engine
    rng = np.random.default_rng(5)
    base = e.pool_stats()
    live = {}
    peak_segments = 0

    def check(ptr):
        n, tag = live[ptr]
        got = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
        assert got[0] == tag and got[n - 1] == tag and got[n // 2] == tag, "a live block was overwritten: extents overlap"

    for it in range(400):
        if live and (rng.random() < 0.45 or len(live) > 40):
            ptr = list(live)[int(rng.integers(len(live)))]
            check(ptr)
            e.device_free(ptr)
            del live[ptr]
        else:
            n = int(rng.choice([700, 5000, 300_000, 3 << 20, 40 << 20, 70 << 20, 90 << 20])) + int(rng.integers(0, 4096))
            ptr = e.device_alloc(n)
            assert ptr % 16 == 0 and ptr not in live          # (the emulator's segments come from malloc: 16-byte aligned; extents are multiples of 512 inside them)
            tag = int(rng.integers(1, 255))
            ctypes.memset(ptr, tag, 1); ctypes.memset(ptr + n - 1, tag, 1); ctypes.memset(ptr + n // 2, tag, 1)
            live[ptr] = (n, tag)
        st = e.pool_stats()
        assert st["live_bytes"] - base["live_bytes"] >= sum(n for n, _ in live.values())
        assert st["segment_bytes"] == st["live_bytes"] + st["free_bytes"]
        peak_segments = max(peak_segments, st["live_bytes"])
    for ptr in list(live):
        check(ptr)
        e.device_free(ptr)
    st = e.pool_stats()
    assert st["live_bytes"] == base["live_bytes"]
    # fragmentation stays bounded: what was taken from the driver is within 1.5 x the largest live total (+ the small segments)
    assert st["segment_bytes"] <= peak_segments * 3 // 2 + (512 << 20), (st, peak_segments)
    # a request as large as the largest free extent is served without new memory: the extents coalesced
    big = st["largest_free_extent"] * 15 // 16         # (size classes round a request up by < 1/32)
    assert big >= 32 << 20
    ptr = e.device_alloc(big)
    assert e.pool_stats()["hipmalloc_calls"] == st["hipmalloc_calls"]
    e.device_free(ptr)
