"""N > 1 path on CPU: world_size 2, gloo.  Each rank sketches its half of the references (ani_sketch_records with global
seqIds), the 12-byte records are all-gathered (the one collective of the path; RCCL on the GPU node, gloo here), every rank
builds the full index from the gathered records (ani_sketch_from_records) and maps its own share of the queries.
The union of the ranks' rows must equal the single-process result.  Uses the CPU build of the product sources (tests/emu)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from fastani_amd.api import Engine, HostGenomes, Sketch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = Engine(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libfastani_emu.so")), 0)
    p = e.params()
    ids = [0, 2, 7, 20]
    genomes = [[orc.synth_genome(5, g, 15000)] if g != 7 else [orc.synth_genome(5, g, 9000), orc.synth_genome(5, 8, 6000)] for g in ids]
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    lo, hi = (len(genomes) * rank) // world, (len(genomes) * (rank + 1)) // world
    ptr, n = e.sketch_records(p, HostGenomes(genomes[lo:hi]), int(gcs[lo]))
    mine = torch.zeros(n * 3, dtype=torch.int32)
    if n:
        e.device_copy(mine.data_ptr(), ptr, n * 12)
        e.device_free(ptr)
    counts = torch.zeros(world, dtype=torch.int64)
    counts[rank] = n
    dist.all_reduce(counts)
    mx = int(counts.max())
    padded = torch.zeros(mx * 3, dtype=torch.int32)
    padded[:n * 3] = mine
    gathered = [torch.zeros(mx * 3, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(gathered, padded)
    rec = torch.cat([gathered[r][:int(counts[r]) * 3] for r in range(world)]).contiguous()
    sk = Sketch(e, p, records=(rec.data_ptr(), int(counts.sum()), contig_len, gcs))
    my_q = list(range(rank, len(genomes), world))
    rows = [sk.map_cgi_batch([genomes[q]], q) for q in my_q]
    rows = np.concatenate(rows) if rows else np.zeros(0)
    np.save(os.path.join(out_dir, "rows%d.npy" % rank), rows)
    np.save(os.path.join(out_dir, "mins%d.npy" % rank), sk.minimizers())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sketch_allgather_matches_single_process(tmp_path, emu_engine):
    import torch.multiprocessing as mp
    import orc
    from fastani_amd.api import Sketch
    world = 2
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ids = [0, 2, 7, 20]
    genomes = [[orc.synth_genome(5, g, 15000)] if g != 7 else [orc.synth_genome(5, g, 9000), orc.synth_genome(5, 8, 6000)] for g in ids]
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
