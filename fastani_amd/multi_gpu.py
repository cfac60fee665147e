"""Reference-sharded many-to-many over the GPUs of one node (one process per GPU; torch.distributed: "nccl" = RCCL over xGMI).

Two forms of the same exchange.  gather_map (round 4, what bench.py runs): ONE all-gather of the ranks' packed fragment sets, started
before the rank maps its own set and waited for after it; the N - 1 foreign sets are then mapped in ONE call as a merged set
(ani_fragset_unpack_merged).  ring_map (round 3): the sets go round a ring, one hop and one mapping call per step.  A fragment set
that meets a foreign shard is almost all launch latency (~107 chance seed hits and 0.7 candidates per fragment against 125 genomes),
so seven small passes of the L1 / L2 kernels cost more than one pass over seven sets: per rank of an 8-GPU 1000 x 1000 job 51 -> 45 ms
(bench.py --simulate-world 8).

The query-sharded form of SURVEY.md section 8e (every GPU holds the whole reference index) stops scaling where the index stops
fitting: N GPUs hold exactly as many references as one, and every rank repeats the whole index build.  Here the REFERENCES are
sharded: rank r sketches and indexes genomes [g0[r], g0[r+1]) only (1/N of the index memory and of the build), and what travels
is the query side — each rank's kept fragment sketches (4 bytes per sketch hash: a third of the 12-byte minimizer records) go
round the ring, one hop per step, while the previous set is being mapped:

    step s:  rank r maps the fragment set that started at rank (r - s) mod N against its shard,
             sends it on to rank r + 1 and receives the next one from rank r - 1  (isend / irecv, overlapped with the mapping)

After N steps every rank has mapped every query against its references; a (query, reference) result does not depend on what else
is in an index, so the union of the ranks' rows is the single-index result.  The reference itself splits its
database the same way and maps every query against every split (computeCoreIdentity.hpp:457-487, core_genome_identity.cpp:55-121).
xGMI is point to point: a ring hop uses one link per direction, 2 GB per 1250-genome set = ~15 ms against ~350 ms of mapping.
"""
import time

import numpy as np

from .api import FragmentSet


def join_rows(parts, dtype=None):
    """the parts of a result as one array.  (Through a byte view: numpy concatenates structured arrays field by field, several times
    slower than a copy — 10^7 rows of one rank of a 10 000 x 10 000 job took longer to join than a tenth of their mapping.)"""
    if dtype is None and len(parts):
        dtype = parts[0].dtype
    parts = [p for p in parts if len(p)]
    if not parts:
        return np.zeros(0, dtype=dtype)
    if len(parts) == 1:
        return parts[0]
    dt = parts[0].dtype
    return np.concatenate([np.ascontiguousarray(p).view(np.uint8) for p in parts]).view(dt)


def ring_map(engine, sk, frags, first_query_ids, ref_base, dist, rank, world, alloc, sync, timers=None, parts=False):
    """sk: this rank's sketch over ITS reference shard (local genome ids; global id = ref_base + local id).
    frags: this rank's kept fragment set (its query genomes); first_query_ids[r] = global id of rank r's first query genome.
    alloc(nbytes) -> (tensor, device pointer) of a byte buffer torch.distributed can send; sync() makes received bytes visible to
    the library's stream (torch.cuda.synchronize on a GPU).  Returns the rows of (all queries) x (this rank's references), with global
    reference ids (ani_sketch_set_ref_id_base); parts=True: as the list of the mapping calls' arrays, not joined (bench.py joins them
    outside its timed region)."""
    import torch
    t = timers if timers is not None else {}
    for k in ("pack_ms", "map_ms", "wait_ms"):
        t.setdefault(k, 0.0)
    t0 = time.perf_counter()
    nb = frags.packed_bytes()
    cap = nb
    if world > 1:
        m = torch.tensor([nb], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            m = m.cuda()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        cap = int(m.item())
    key = ("ring_bufs", cap)
    bufs = getattr(engine, "_ring_bufs", None)
    if bufs is None or bufs[0] != key:
        bufs = (key, [alloc(cap), alloc(cap)])
        engine._ring_bufs = bufs
    (b0, p0), (b1, p1) = bufs[1]
    tens, ptrs = [b0, b1], [p0, p1]
    frags.pack_into(ptrs[0], cap)
    sync()
    t["pack_ms"] += (time.perf_counter() - t0) * 1e3
    out = []
    cur = 0
    sk.set_ref_id_base(ref_base)                          # the rows come back with global reference ids
    for s in range(world):
        src = (rank - s) % world
        reqs = None
        if s < world - 1:
            ops = [dist.P2POp(dist.isend, tens[cur], (rank + 1) % world), dist.P2POp(dist.irecv, tens[1 - cur], (rank - 1) % world)]
            reqs = dist.batch_isend_irecv(ops)
        t1 = time.perf_counter()
        view = FragmentSet.unpack(engine, ptrs[cur], cap, keepalive=tens[cur])
        rows = sk.map_cgi_fragset(view, int(first_query_ids[src]))
        view.close()
        out.append(rows)
        t2 = time.perf_counter()
        if reqs is not None:
            for r in reqs:
                r.wait()
            sync()
        t3 = time.perf_counter()
        t["map_ms"] += (t2 - t1) * 1e3
        t["wait_ms"] += (t3 - t2) * 1e3
        cur = 1 - cur
    sk.set_ref_id_base(0)
    return out if parts else join_rows(out, rows.dtype)


def gather_map(engine, sk, frags, first_query_ids, ref_base, dist, rank, world, alloc, sync, timers=None, parts=False):
    """Same contract as ring_map.  The packed sets are all-gathered (slot = the largest set, agreed by one tiny all-reduce) while this
    rank maps its own set; the foreign sets are mapped as ONE merged set over the gather buffer."""
    import torch
    t = timers if timers is not None else {}
    for k in ("pack_ms", "map_ms", "wait_ms"):
        t.setdefault(k, 0.0)
    t0 = time.perf_counter()
    nb = frags.packed_bytes()
    cap = nb
    if world > 1:
        m = torch.tensor([nb], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            m = m.cuda()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        cap = int(m.item())
    cap = (cap + 255) // 256 * 256
    key = ("gather_buf", cap, world)
    bufs = getattr(engine, "_ring_bufs", None)
    if bufs is None or bufs[0] != key:
        bufs = (key, [alloc(cap * world)])
        engine._ring_bufs = bufs
    buf, ptr = bufs[1][0]
    # the in-place form of the all-gather (input = this rank's slot of the output) is only valid at offset rank * cap of a buffer of
    # exactly cap * world elements, and the merged view needs the ranks' query ranges in rank order (ani_fragset_unpack_merged checks
    # the order again on the headers that arrive)
    if buf.numel() * buf.element_size() != cap * world:
        raise RuntimeError("gather buffer of %d bytes for %d slots of %d" % (buf.numel() * buf.element_size(), world, cap))
    if any(int(first_query_ids[i]) > int(first_query_ids[i + 1]) for i in range(world - 1)):
        raise ValueError("first_query_ids must be non-decreasing by rank: %r" % (list(first_query_ids),))
    mine = buf[rank * cap:(rank + 1) * cap]
    sk.set_ref_id_base(ref_base)                          # the rows come back with global reference ids
    frags.pack_into(ptr + rank * cap, cap)
    sync()
    work = dist.all_gather_into_tensor(buf, mine, async_op=True) if world > 1 else None      # the one collective of the path ...
    t1 = time.perf_counter()
    out = [sk.map_cgi_fragset(frags, int(first_query_ids[rank]))]                              # ... under the mapping of the rank's own set
    t2 = time.perf_counter()
    if work is not None:
        work.wait()
        sync()
        t3 = time.perf_counter()
        base = [int(first_query_ids[s]) if s != rank else -1 for s in range(world)]
        merged = FragmentSet.unpack_merged(engine, ptr, cap, base, keepalive=buf)
        out.append(sk.map_cgi_fragset(merged, 0))
        merged.close()
        t4 = time.perf_counter()
        t["wait_ms"] += (t3 - t2) * 1e3
        t["map_ms"] += (t4 - t3) * 1e3
    t["pack_ms"] += (t1 - t0) * 1e3
    t["map_ms"] += (t2 - t1) * 1e3
    sk.set_ref_id_base(0)
    return out if parts else join_rows(out)
