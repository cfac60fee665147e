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
