"""N > 1 path on CPU: world_size 2, gloo.  Each rank sketches its half of the references (ani_sketch_records with global
seqIds), the 12-byte records are all-gathered (the one collective of the path; RCCL on the GPU node, gloo here), every rank
builds the full index from the gathered records (ani_sketch_from_records) and maps its own share of the queries.
The union of the ranks' rows must equal the single-process result.  Uses the CPU build of the product sources (tests/emu)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _genomes(n):
    import orc
    ids = [0, 2, 7, 20, 21][:n]
    return [[orc.synth_genome(5, g, 15000)] if g != 7 else [orc.synth_genome(5, g, 9000), orc.synth_genome(5, 8, 6000)] for g in ids]


def _worker(rank, world, port, out_dir, n_genomes):
    """the staging of bench.py's N > 1 step: one slot per rank in the gather buffer, the slot's first record carries the count, ONE
    all-gather, the index built from the slots as record parts (no concatenation)"""
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fastani_amd.api import Engine, HostGenomes, Sketch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = Engine(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libfastani_emu.so")), 0)
    p = e.params()
    genomes = _genomes(n_genomes)
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    lo, hi = (len(genomes) * rank) // world, (len(genomes) * (rank + 1)) // world      # uneven, possibly empty
    slot = 4000                                                                        # records per rank, upper bound
    allrec = torch.zeros(world * (slot + 1) * 3, dtype=torch.int32)
    mine = allrec[rank * (slot + 1) * 3:(rank + 1) * (slot + 1) * 3]
    n = 0
    if hi > lo:
        ptr, n = e.sketch_records(p, HostGenomes(genomes[lo:hi]), int(gcs[lo]))
        assert n <= slot
        if n:
            e.device_copy(mine.data_ptr() + 12, ptr, n * 12)
            e.device_free(ptr)
    mine[0] = n
    gathered = torch.zeros_like(allrec)
    dist.all_gather_into_tensor(gathered, mine.clone())
    counts = [int(gathered[r * (slot + 1) * 3]) for r in range(world)]
    ptrs = [gathered.data_ptr() + (r * (slot + 1) + 1) * 12 for r in range(world)]
    part_g0 = np.array([(len(genomes) * r) // world for r in range(world + 1)], dtype=np.int32)
    sk = Sketch(e, p, record_parts=(ptrs, counts, part_g0, contig_len, gcs))
    my_q = list(range(rank, len(genomes), world))
    rows = [sk.map_cgi_batch([genomes[q]], q) for q in my_q]
    rows = np.concatenate(rows) if rows else np.zeros(0, dtype=fastani_rows_dtype())
    np.save(os.path.join(out_dir, "rows%d.npy" % rank), rows)
    np.save(os.path.join(out_dir, "mins%d.npy" % rank), sk.minimizers())
    dist.barrier()
    dist.destroy_process_group()


def _ring_worker(rank, world, port, out_dir, n_genomes, exchange="ring"):
    """the reference-sharded step of bench.py --config c4 (fastani_amd/multi_gpu.py): every rank sketches and indexes ITS genomes
    only (fused all-vs-all pass: reference records + kept fragment sketches), the packed fragment sets go round the ring and every
    rank maps every set against its shard"""
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fastani_amd.api import Engine, HostGenomes, Sketch
    from fastani_amd.multi_gpu import gather_map, ring_map
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = Engine(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libfastani_emu.so")), 0)
    p = e.params()
    genomes = _genomes(n_genomes)
    part_g0 = [(len(genomes) * r) // world for r in range(world + 1)]
    lo, hi = part_g0[rank], part_g0[rank + 1]                                           # uneven, possibly empty
    mine = genomes[lo:hi]
    contig_len = np.array([len(c) for g in mine for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in mine]).astype(np.int32)
    ptr, n, frags = e.sketch_records_self(p, HostGenomes(mine), 0)
    sk = Sketch(e, p, records=(ptr, n, contig_len, gcs))
    if n:
        e.device_free(ptr)

    def alloc(nbytes):
        t = torch.zeros(nbytes, dtype=torch.uint8)
        return t, t.data_ptr()
    timers = {}
    rows = (ring_map if exchange == "ring" else gather_map)(e, sk, frags, part_g0, lo, dist, rank, world, alloc, lambda: None, timers)
    frags.close()
    np.save(os.path.join(out_dir, "ring%d.npy" % rank), rows)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_genomes,exchange", [(2, 5, "ring"), (4, 3, "ring"), (2, 5, "gather"), (4, 3, "gather"), (4, 5, "gather")])
def test_reference_sharded_ring_matches_single_process(tmp_path, emu_engine, world, n_genomes, exchange):
    """world 4 with 3 genomes: rank 0 has no genome at all — an empty index and an empty fragment set still take part in the exchange
    (ring: one hop and one mapping call per set; gather: ONE all-gather, the foreign sets mapped as one merged set)"""
    import torch.multiprocessing as mp
    from fastani_amd.api import Sketch
    port = 29900 + (os.getpid() % 400) + 10 * world + n_genomes + (50 if exchange == "gather" else 0)
    mp.spawn(_ring_worker, args=(world, port, str(tmp_path), n_genomes, exchange), nprocs=world, join=True)
    genomes = _genomes(n_genomes)
    p = emu_engine.params()
    single = Sketch(emu_engine, p, genomes).map_cgi_batch(genomes, 0)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "ring%d.npy" % r)) for r in range(world)])
    got = got[np.lexsort((got["refGenomeId"], got["qryGenomeId"]))]
    assert np.array_equal(got, single)


def fastani_rows_dtype():
    from fastani_amd.api import CGI_DT
    return CGI_DT


@pytest.mark.parametrize("world,n_genomes", [(2, 4), (4, 3), (4, 5)])
def test_sharded_sketch_allgather_matches_single_process(tmp_path, emu_engine, world, n_genomes):
    """world 4 with 3 genomes: rank 0 holds no reference at all, the others one each; with 5 genomes the shards are 1/1/1/2"""
    import torch.multiprocessing as mp
    import orc
    from fastani_amd.api import Sketch
    port = 29500 + (os.getpid() % 400) + 10 * world + n_genomes
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_genomes), nprocs=world, join=True)
    genomes = _genomes(n_genomes)
    p = emu_engine.params()
    sk = Sketch(emu_engine, p, genomes)
    single = sk.map_cgi_batch(genomes, 0)
    mins = sk.minimizers()
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "rows%d.npy" % r)) for r in range(world)])
    got = got[np.lexsort((got["refGenomeId"], got["qryGenomeId"]))]
    assert np.array_equal(got, single)
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "mins%d.npy" % r)), mins)
    # and the oracle agrees with both
    osk = orc.Sketch(genomes, 16, p.windowSize)
    exp = []
    for qi, g in enumerate(genomes):
        maps, tot = osk.map_genome(g)
        exp.append(osk.compute_cgi(maps, tot, qi))
    assert np.array_equal(single, np.concatenate(exp))


@pytest.mark.parametrize("config,scaling,launcher", [("many-to-many", "auto", True), ("many-to-many", "strong", True), ("c4", "auto", True),
                                                     ("many-to-many", "auto", False)])
def test_bench_orchestration_two_ranks(config, scaling, launcher, emu_engine, tmp_path):
    """bench.py's own multi-rank steps launched the way the driver launches it, on the CPU build of the product sources over gloo
    (ANI_BENCH_BACKEND=emu, a test-only switch): 6 genomes of one cluster, every pair related.
      many-to-many (default at N > 1) = WEAK scaling (north_star's split): fixed 6-genome database, 6 queries per rank, query-sharded, reference
          records all-gathered; the strong-scaling leg (the fixed 6 x 6 job, reference-sharded, fragment sets all-gathered) is measured beside it
      many-to-many --scaling strong   = the strong job as the timed region, the weak leg beside it
      c4                              = the reference-sharded job again (configs[3] shape)
    launcher = False: plain `python bench.py --gpus 2` with no WORLD_SIZE around it — bench.py starts its own ranks (self_launch).
    The rows every rank dumps must add up to the single-process rows."""
    import json
    import subprocess
    from fastani_amd.api import DeviceGenomes, Sketch
    port = 29300 + (os.getpid() % 300) + (7 if config == "c4" else 0) + (13 if scaling == "weak" else 0)
    dump = os.path.join(str(tmp_path), "rows")
    outer = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher else [sys.executable]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(outer + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", config, "--scaling", scaling, "--genomes", "6",
                                "--genome-len", "24000", "--dump-rows", dump],
                       capture_output=True, env=dict(env, ANI_BENCH_BACKEND="emu"), timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and "EMULATION" in out["data"] and out["rows_identical_across_steps"]
    ranks = out["ranks"]
    assert ranks["ranks_seen_by_rccl"] == 2 and len(ranks["step_ms"]) == 2
    strong = config == "c4" or scaling == "strong"
    assert out["scaling"] == ("strong" if strong else "weak")
    if strong:
        assert out["config"]["query_genomes"] == 6 and sum(ranks["rows"]) == 36 and "reference-sharded" in ranks["mode"]
        assert ranks["bytes_moved_per_rank"] > 0
        if config == "many-to-many":
            wl = out["weak_scaling_leg"]
            assert wl["scaling"] == "weak" and wl["ranks"]["rows"] == [36, 36] and wl["value"] > 0
    else:
        assert out["config"]["query_genomes"] == 12 and ranks["rows"] == [36, 36] and "weak_scaling_leg" not in out
    for k in ("ref_records_ms", "fragsketch_ms", "allgather_ms", "index_ms", "map_ms", "ring_wait_ms", "other_ms"):
        assert len(ranks[k]) == 2
    # the union of the ranks' rows = the single-process rows of the same synthetic set (device generator on the emulator)
    got = np.concatenate([np.load("%s.rank%d.npy" % (dump, x)) for x in range(2)])
    got = got[np.lexsort((got["refGenomeId"], got["qryGenomeId"]))]
    L, n = 24000, 6
    words = (L + 15) // 16
    buf = np.zeros(n * words + 64, dtype=np.uint32)
    emu_engine.synth_packed(20260925, 0, n, L, buf.ctypes.data)
    dg = DeviceGenomes(buf.ctypes.data, n, L)
    p = emu_engine.params()
    single = Sketch(emu_engine, p, dg).map_cgi_batch(dg, 0)
    if strong:
        assert np.array_equal(got, single)
        # the line's own row check for multi-rank runs: count and multiset hash over all ranks = those of the single-process rows
        sys.path.insert(0, ROOT)
        import bench
        ms = out["rows_multiset"]
        assert ms["ranks"] == 2 and ms["rows"] == len(single) and ms["hash"] == "%016x" % bench.rows_multiset_hash(single), ms
        assert ms["expected"] is None and ms["ok"] is None          # no stored expectation for a 6-genome set
        half = len(single) // 2                                    # the hash adds up over any partition, in any order
        assert (bench.rows_multiset_hash(single[:half]) + bench.rows_multiset_hash(single[half:][::-1])) % (1 << 64) == bench.rows_multiset_hash(single)
        other = single.copy(); other["countSeq"][0] += 1
        assert bench.rows_multiset_hash(other) != bench.rows_multiset_hash(single)
    else:
        # rank 0 maps variant 0 (= the references themselves), rank 1 variant 1 with query ids 6..11
        assert np.array_equal(got[got["qryGenomeId"] < 6], single) and len(got) == 72
        # the strong-scaling leg beside the weak timed region: the ranks' rows add up to the single-process rows (count and multiset hash)
        sys.path.insert(0, ROOT)
        import bench
        sl = out["strong_scaling_leg"]
        assert "error" not in sl, sl
        assert sl["scaling"] == "strong" and sl["value"] > 0 and sl["rows_identical_across_steps"] and sum(sl["ranks"]["rows"]) == 36
        assert sl["rows_multiset"]["rows"] == len(single) and sl["rows_multiset"]["hash"] == "%016x" % bench.rows_multiset_hash(single)
        # rank 0 of the weak job maps variant 0 = the references themselves: its rows are the single-process rows (no stored expectation at this size)
        assert out["rows_multiset"]["rows"] == 72 and "rank0" not in out["rows_multiset"]


def test_bench_dry_collectives_two_ranks(tmp_path):
    """bench.py --dry-collectives: only the collectives of the strong-scaling step (one-word all-reduce, in-place all-gather with the
    real slot size, barrier), the line a first run on an 8-GPU node starts with"""
    import json
    import subprocess
    port = 29650 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-collectives", "--genomes", "6", "--genome-len", "24000"],
                       capture_output=True, env=dict(os.environ, ANI_BENCH_BACKEND="emu"), timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["dry_collectives"] and out["n_gpus"] == 2 and out["slots_arrived_intact"] == [True, True]
    assert out["slot_bytes"] % 256 == 0 and out["bytes_received_per_rank"] == out["slot_bytes"] and len(out["all_gather_GBps_received_per_rank"]) == 2


def test_bench_c5_reference_blocks(emu_engine, tmp_path):
    """bench.py --config c5 with the reference set taken in BLOCKS (--ref-block: each block of genomes sketched slice by slice,
    indexed, mapped by every query and dropped — how a set whose records exceed the device memory runs on one GPU): the rows equal
    those of one index over the whole set, whatever the block and slice sizes."""
    import json
    import subprocess
    from fastani_amd.api import DeviceGenomes, Sketch
    L, n, nq = 24000, 7, 3
    words = (L + 15) // 16
    buf = np.zeros(n * words + 64, dtype=np.uint32)
    emu_engine.synth_packed(20260925, 0, n, L, buf.ctypes.data)
    p = emu_engine.params()
    single = Sketch(emu_engine, p, DeviceGenomes(buf.ctypes.data, n, L)).map_cgi_batch(DeviceGenomes(buf.ctypes.data, nq, L), 0)
    assert len(single) >= 2 * nq
    # (--query-slice: the queries as one or several kept fragment sets; --query-block: the queries generated inside the step and taken in
    #  blocks, the reference set streamed once per query block — how 90 000 x 90 000 runs on one GPU)
    for block, slc, qslc, qblock in ((3, 2, 2, 0), (4, 4, 1, 0), (7, 3, 2000, 0), (3, 2, 1, 2)):
        dump = os.path.join(str(tmp_path), "rows_%d_%d_%d" % (block, slc, qblock))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--config", "c5", "--genomes", str(n), "--queries", str(nq),
                            "--genome-len", str(L), "--ref-block", str(block), "--slice-genomes", str(slc), "--query-slice", str(qslc), "--dump-rows", dump]
                           + (["--query-block", str(qblock)] if qblock else []),
                           capture_output=True, env=dict(os.environ, ANI_BENCH_BACKEND="emu"), timeout=1200)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
        nqb = -(-nq // qblock) if qblock else 1
        assert out["config"]["ref_block"]["genomes"] == block and out["config"]["ref_block"]["blocks"] == -(-n // block) * nqb
        assert out["config"]["query_block"]["blocks"] == nqb and out["config"]["query_block"]["queries_generated_inside_the_step"] == bool(qblock)
        got = np.load(dump + ".npy")
        got = got[np.lexsort((got["refGenomeId"], got["qryGenomeId"]))]    # (several blocks arrive block-major)
        assert np.array_equal(got, single)
    # --drop-rows: the rows are counted and checksummed block by block, only the sampled queries' rows stay
    dump = os.path.join(str(tmp_path), "rows_dropped")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--config", "c5", "--genomes", str(n), "--queries", str(nq),
                        "--genome-len", str(L), "--ref-block", "3", "--slice-genomes", "2", "--query-slice", "1", "--query-block", "2", "--drop-rows", "--sample-queries", "1",
                        "--dump-rows", dump], capture_output=True, env=dict(os.environ, ANI_BENCH_BACKEND="emu"), timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["rows_last_step"] == len(single) and out["rows_identical_across_steps"]
    kept_q = out["rows_kept"]["of_queries"]
    got = np.load(dump + ".npy")
    got = got[np.lexsort((got["refGenomeId"], got["qryGenomeId"]))]
    assert len(kept_q) == 1 and np.array_equal(got, single[single["qryGenomeId"] == kept_q[0]])
