#!/usr/bin/env python3
"""bench.py — ANI pairs/sec, many-to-many NxN ~5 Mbp genomes (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W [--config many-to-many|one-to-many|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...        (no launcher around it: bench.py starts its N ranks itself, see self_launch)

One step = one pass of the whole hot path over the workload with the packed genomes already resident in HBM:
reference sketch + index build (skch::Sketch), mapping of every query genome (skch::Map) and the ANI reducer
(cgi::computeCGI), ending with the CGI rows on the host.

Workloads (config.workload):
  many-to-many (default) = BASELINE.json configs[2], 1000x1000 synthetic ~5 Mbp genomes, k=16 fragLen=3000.
      N = 1 : 1000 reference genomes x 1000 query genomes (the same set: all-vs-all).
      N > 1 : WEAK scaling (default; BASELINE.json north_star's own split, "scaling": "weak"): the 1000-genome reference database is
              fixed, every GPU maps its own 1000 query genomes (rank r maps variant r of the clustered set; variant 0 is the
              reference set itself), each rank sketches 1/N of the references, the 12-byte minimizer records are all-gathered
              once over RCCL/xGMI while the rank sketches its queries' fragments, and every rank builds the full index.
              value = N x 10^6 pairs / step.  The STRONG-scaling figure is measured beside it (strong_scaling_leg; --scaling strong
              makes it the timed region and the weak job the leg): the SAME 1000 x 1000 job on N GPUs, reference-sharded — rank r
              hashes genomes [r*1000/N, (r+1)*1000/N) once for both roles, indexes that shard only, and the ranks' packed fragment
              sketches are all-gathered (one collective, under the mapping of the rank's own set): nothing is replicated.  The
              leg's rows over all ranks are the standard job's rows: rows_multiset holds them against the stored value.
      --simulate-world W (one GPU): the compute of ONE rank of the W-GPU strong job, without communication — the measured
              per-rank time behind the predicted W-GPU figure.
  one-to-many = configs[1]: 1 query genome (cluster 0, member 1) against the 1000-genome set; a step still sketches and
      indexes the references (the reference does too); the map-only latency is reported beside it.
  c5 = configs[4]'s reference side: a reference set that does not stay resident (default 30 000 x 5 Mbp, --genomes 90000 for the
      full count; generated and sketched slice by slice; only the 12-byte minimizer records of one BLOCK of references stay
      (--ref-block, default: a third of the device memory), the block's index is streamed chunk by chunk, every query is mapped
      against every block) x --queries genomes (default 300; their fragment sketches are built once per step as kept sets of
      --query-slice genomes and every block maps all of them in one call.  Measured: 90 000 x 10 000 = 9e8 pairs in 52.9 s).
  c4 = configs[3]: 10000 x 10000 (all-vs-all, 500 clusters).
      N = 1 : the reference set held as several index chunks (streamed through the device when they do not fit); --queries bounds
              the query count.
      N > 1 : REFERENCE-sharded (fastani_amd/multi_gpu.py): rank r sketches and indexes genomes [r*10000/N, (r+1)*10000/N) only;
              the ranks' query fragment sketches go round a ring (isend/irecv over RCCL, overlapped with the mapping) and every rank
              maps every query against its shard.  No replicated index build, 1/N of the index memory per GPU.

Prints ONE JSON line on rank 0 (see the driver contract): value = whole-job pairs/sec, plus
  roofline     — the dominant kernel (L2 sliding MinHash), algorithmic bytes (12*m_c + 4*s per candidate)
                 over its HIP-event time on the launch stream, against the 8 TB/s HBM peak; .stage = the L2 stage as a
                 whole on the same bytes; .int_ops = the two hashing kernels against the VALU integer ceiling;
  parity_timed_rows — the rows of the LAST TIMED STEP compared with the untouched reference binary (every reference x
                 the CPU-baseline queries) and with the C oracle on >= 200 random pairs;
  cpu_baseline — oracle/_ref/fastANI_ref (the untouched reference, built by oracle/Makefile) timed on this box's host cores
                 on a bounded sample of the same workload: ALL references x 24 queries in ONE run at -t 16 (its fastest thread
                 count on the 128-core host), fixed and per-query cost read from its own stderr timers (rank 0, N = 1);
  end_to_end   — the drop-in CLI (fastani_amd/fastANI) on the same FASTA files on local disk -> output file, wall clock.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  RCCL brings streams of its own, and with
# them in the process the library's main and side stream landed on ONE queue: the L2 simulation no longer ran beside the next chunk's
# codes kernel (L2 stage 86.4 instead of 80.0 ms per 1000 x 1000 step, measured with one rank over RCCL; profiles/r06r_dist_ab.txt).
# Eight queues restore it.  Must be in the environment before the runtime starts, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process device work (RCCL between the ranks): this host driver only supports dmabuf IPC — without the setting hipIpcGetMemHandle
# fails with "invalid argument".  The image exports it already; a launcher with a scrubbed environment would not.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU integer ceiling for the MurmurHash3 kernels, from the instruction costs measured on MI355X (tools/ubench/valu.hip,
# profiles/r02_ubench_valu.txt): every wave64 VALU instruction issues in ~4 SIMD cycles (v_add/v_xor 4.0, v_mul_lo/hi_u32 and
# v_mad_u64_u32 4.4-4.6: 32-bit integer multiplies are NOT quarter-rate on this part).  One hash = 8 64-bit multiplies = 24
# multiplier instructions + >= 57 others (rotates, xors, 64-bit adds).  1024 SIMDs at 2.4 GHz.
UBENCH = {"full_rate_cyc": 4.06, "mul32_cyc": 4.5, "clock_ghz": 2.4, "simds": 1024,
          "mul_instr_per_hash": 24, "other_instr_per_hash": 57}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="many-to-many", choices=["many-to-many", "one-to-many", "c4", "c5"])
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1, many-to-many: weak (default; BASELINE.json north_star's split) = fixed database, 1000 query genomes per GPU, the reference records "
                         "all-gathered once; strong = the fixed 1000 x 1000 job, references sharded, fragment sketches all-gathered.  The other of the two is measured "
                         "beside the timed region (strong_scaling_leg / weak_scaling_leg)")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1, many-to-many --scaling strong: skip the weak-scaling measurement beside the strong one")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1, many-to-many: skip the strong-scaling measurement beside the weak one")
    ap.add_argument("--simulate-world", type=int, default=0, help="ONE GPU computes what one rank of a W-GPU strong-scaling (ring) job computes, without the communication: "
                                                                  "the measured per-rank time behind the predicted W-GPU figure (DESIGN.md section 5)")
    ap.add_argument("--simulate-rank", type=int, default=0)
    ap.add_argument("--dry-collectives", action="store_true", help="N > 1: run ONLY the collectives of the multi-rank step — the one-word all-reduce, the in-place all-gather with the "
                    "real slot size of the workload, the barrier — 10 times, print per-rank seconds and GB/s as one JSON line and leave (tells a hang from a slow link in ~30 s)")
    ap.add_argument("--dump-rows", default="", help="write the rows of the last timed step to <path>[.rank<r>].npy (tests)")
    ap.add_argument("--slice-genomes", type=int, default=1000, help="c5: genomes generated and sketched per slice")
    ap.add_argument("--query-slice", type=int, default=2000, help="c5: query genomes per kept fragment set")
    ap.add_argument("--query-block", type=int, default=0, help="c5: query genomes whose fragment sets are resident together (0 = all of --queries).  With it the queries, too, are generated "
                    "slice by slice inside the step and the reference set is streamed once per query block: 90 000 x 90 000 = --queries 90000 --query-block 30000")
    ap.add_argument("--drop-rows", action="store_true", help="c5: count and checksum the rows of a step block by block and keep only those of --sample-queries query genomes "
                    "(the 6 x 10^9 rows of 90 000 x 90 000 are 126 GB)")
    ap.add_argument("--sample-queries", type=int, default=24, help="--drop-rows: query genomes whose rows are kept for the oracle check")
    ap.add_argument("--ref-block", type=int, default=0, help="c5: reference genomes indexed and mapped together (0 = as many as keep the records in a third of the device memory)")
    ap.add_argument("--genomes", type=int, default=0, help="reference genomes (0 = the config's: 1000, c4: 10000, c5: 30000)")
    ap.add_argument("--queries", type=int, default=0, help="query genomes per GPU (0 = the config's)")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--cluster-size", type=int, default=20, help="related genomes per cluster (20 = SURVEY.md section 8d; 100 / 500 = a species-dense database: "
                                                                  "every fragment has more seed hits than the LDS classes hold and takes the batched global-memory L1 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-self", action="store_true", help="all-vs-all: sketch the query role separately (as for unrelated query sets)")
    ap.add_argument("--cpu-queries", type=int, default=24, help="query genomes of the fastANI_ref sample (one run; fixed and marginal cost from its own stderr timers)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="fastANI_ref -t: 16 is where the reference is fastest on the 128-core benchmark host (1 x 1000: 39.6 s at 16, 51.9 at 32, "
                         "83.5 at 64, 117.4 at 128 threads — every thread builds its own hash-map index; profiles/r02_refscale.txt)")
    ap.add_argument("--cpu-refs", type=int, default=0, help="0 = all references of the workload")
    ap.add_argument("--oracle-pairs", type=int, default=240)
    ap.add_argument("--workdir", default="", help="where the FASTA copies of the synthetic set go (default: a temp dir)")
    return ap.parse_args()


def host_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return {"cpu_model": model, "logical_cpus": logical, "physical_cores": len(phys) or logical}


def write_fasta_set(orc, seed, ids, L, td, threads, cluster_size=20):
    """FASTA copies of genomes `ids` of the synthetic set (the oracle's generator is the CPU twin of ani_synth_packed,
    byte-compared in the tests)."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np

    def one(g):
        p = os.path.join(td, "g%05d.fa" % g)
        if not os.path.exists(p):
            seq = orc.synth_genome(seed, g, L, cluster_size=cluster_size)
            # 80-column FASTA, header >g<i>: 62 500 lines per 5 Mbp genome, the layout the reference's kseq
            # reader and the command line's block reader are both timed on
            b = seq.tobytes()
            n80 = len(b) // 80
            body = np.empty((n80, 81), dtype=np.uint8)
            body[:, :80] = np.frombuffer(b, dtype=np.uint8, count=n80 * 80).reshape(n80, 80)
            body[:, 80] = 10
            with open(p + ".tmp", "wb") as f:
                f.write(b">g%d\n" % g)
                f.write(body.tobytes())
                if len(b) > n80 * 80:
                    f.write(b[n80 * 80:] + b"\n")
            os.replace(p + ".tmp", p)
        return p
    with ThreadPoolExecutor(max(1, min(threads, 32))) as ex:
        return list(ex.map(one, ids))


def trusted(cnt, L, frag=3000, min_fraction=0.2):
    import numpy as np
    glen = (L // frag) * frag
    return cnt * frag >= glen * np.float32(min_fraction)       # computeCoreIdentity.hpp:328-332 (both genomes have length L)


def read_ref_out(path, index_of):
    rows = {}
    for line in open(path):
        q, r, ani, cnt, tot = line.rstrip("\n").split("\t")
        rows[(index_of[q], index_of[r])] = (float(ani), int(cnt), int(tot))
    return rows


class RowLookup:
    """the rows of a step by (query genome, reference genome) -> (identity, countSeq, totalQueryFragments).  A query's rows are put
    into a dict when the query is first asked for: 90 000 x 10 000 genomes leave 7 x 10^8 rows, and one dict over all of them (as
    this was until round 5) took minutes and > 100 GB of host memory for a check that samples two dozen pairs."""

    def __init__(self, rows):
        self.rows, self._by_q, self._all = rows, {}, None

    def _query(self, q):
        d = self._by_q.get(q)
        if d is None:
            r = self.rows[self.rows["qryGenomeId"] == q]
            d = {int(g): (float(a), int(c), int(t)) for g, a, c, t in zip(r["refGenomeId"], r["identity"], r["countSeq"], r["totalQueryFragments"])}
            self._by_q[q] = d
        return d

    def get(self, key, default=None):
        return self._query(int(key[0])).get(int(key[1]), default)

    def items(self):
        if self._all is None:
            r = self.rows
            self._all = {(int(q), int(g)): (float(a), int(c), int(t)) for q, g, a, c, t in
                         zip(r["qryGenomeId"], r["refGenomeId"], r["identity"], r["countSeq"], r["totalQueryFragments"])}
        return self._all.items()


def compare_with_reference(rows_by_pair, ref_rows, queries, nrefs, L):
    """rows of the timed step for the given query genomes vs the reference binary's output file (every printed row)"""
    got = {k: v for k, v in rows_by_pair.items() if k[0] in queries and k[1] < nrefs and trusted(v[1], L)}
    ok = set(got) == set(ref_rows)
    max_d = 0.0
    bad = 0
    for key, (ani, cnt, tot) in ref_rows.items():
        if key not in got:
            bad += 1
            continue
        g_ani, g_cnt, g_tot = got[key]
        if cnt != g_cnt or tot != g_tot:
            bad += 1
        max_d = max(max_d, abs(ani - g_ani))                   # the reference prints 6 significant digits
    ok = ok and bad == 0 and max_d <= 1e-3
    return {"rows_compared": len(ref_rows), "ok": bool(ok), "mismatching_rows": bad + len(set(got) ^ set(ref_rows)), "max_abs_ani_diff": round(max_d, 6)}


def oracle_spot_check(orc, args, rows_by_pair, n_refs, query_ids, L, window, pairs_wanted, ref_base=0):
    """>= pairs_wanted (query, reference) pairs of the timed result against the C oracle, pair by pair (SURVEY.md App. A.7:
    a pair's result does not depend on what else is in the index, so a small oracle index over the sampled references is
    comparable with the 1000-genome GPU index).  Bit-exact: countSeq, totalQueryFragments and the float identity.
    Half of the sampled queries come from the sampled references' clusters, so related pairs are covered."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(args.seed + 17)
    query_ids = sorted(int(q) for q in query_ids)
    n_r = 12 if len(query_ids) >= 20 else min(n_refs, 24)
    n_q = min(len(query_ids), max(1, (pairs_wanted + n_r - 1) // n_r))
    # the references of the rows are genomes [ref_base, ref_base + n_refs) (a rank's shard, or the whole set)
    in_shard = [q for q in query_ids if ref_base <= q < ref_base + n_refs] or query_ids
    rel = rng.choice(in_shard, size=min(len(in_shard), 4), replace=False)              # reference clusters = clusters of some queries
    cs = args.cluster_size
    clusters = sorted(set(int(q) // cs for q in rel))
    per = max(1, n_r // max(1, len(clusters)))
    ref_ids = []
    for cl in clusters:
        members = [cl * cs + m for m in range(cs) if ref_base <= cl * cs + m < ref_base + n_refs]
        ref_ids += [int(x) for x in rng.choice(members, size=min(per, len(members)), replace=False)] if members else []
    while len(ref_ids) < min(n_r, n_refs):
        g = ref_base + int(rng.integers(n_refs))
        if g not in ref_ids:
            ref_ids.append(g)
    ref_ids = sorted(ref_ids)[:n_r]
    near = [q for q in query_ids if q // cs in clusters]
    q_ids = set(int(x) for x in rng.choice(near, size=min(len(near), (n_q + 1) // 2), replace=False)) if near else set()
    for q in rng.permutation(query_ids):
        if len(q_ids) >= n_q:
            break
        q_ids.add(int(q))
    q_ids = sorted(q_ids)
    t0 = time.time()
    refs = [[orc.synth_genome(args.seed, g, L, cluster_size=cs)] for g in ref_ids]
    osk = orc.Sketch(refs, 16, window)

    def one(q):
        maps, tot = osk.map_genome([orc.synth_genome(args.seed, q, L, cluster_size=cs)])
        return q, osk.compute_cgi(maps, tot, q)
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, q_ids))
    checked = mism = rows_found = 0
    for q, cg in res:
        exp = {int(r["refGenomeId"]): (np.float32(r["identity"]), int(r["countSeq"]), int(r["totalQueryFragments"])) for r in cg}
        for j, g in enumerate(ref_ids):
            checked += 1
            got = rows_by_pair.get((q, g))
            e = exp.get(j)
            if (got is None) != (e is None):
                mism += 1
            elif got is not None:
                rows_found += 1
                if not (np.float32(got[0]) == e[0] and got[1] == e[1] and got[2] == e[2]):
                    mism += 1
    return {"pairs_checked": checked, "pairs_with_rows": rows_found, "mismatches": mism, "ok": mism == 0, "bar": "bit-exact (countSeq, totalQueryFragments, float identity)",
            "queries": len(q_ids), "refs": len(ref_ids), "oracle_seconds": round(time.time() - t0, 1)}


def ref_timeline(cmd, n_queries):
    """Runs the reference binary once and reads its cost model out of its own stderr (every line stamped on arrival):
    thread 0 prints 'Time spent sketching the reference', then per query 'Start Map i', 'Time spent mapping fragments in query #i',
    'Time spent post mapping' (core_genome_identity.cpp:60-105), and the process prints 'parallel_for execution finished' when the
    slowest thread is through its queries (:119)."""
    import statistics
    t0 = time.time()
    r = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    lines = []
    for raw in r.stderr:
        lines.append((time.time() - t0, raw.decode(errors="replace").rstrip("\n")))
    r.wait()
    wall = time.time() - t0
    if r.returncode != 0:
        raise SystemExit("fastANI_ref failed (%d):\n%s" % (r.returncode, "\n".join(ln for _, ln in lines[-20:])))
    sketch_timer = next((float(ln.split(":")[-1].split()[0]) for _, ln in lines if "Time spent sketching the reference" in ln), None)
    t_start_map1 = next((st for st, ln in lines if ln.rstrip().endswith("Start Map 1")), None)
    t_loop_done = next((st for st, ln in lines if "parallel_for execution finished" in ln), None)
    t_map = [float(ln.split(":")[-1].split()[0]) for _, ln in lines if "Time spent mapping fragments in query" in ln]
    t_post = [float(ln.split(":")[-1].split()[0]) for _, ln in lines if "Time spent post mapping" in ln]
    per_q = [a + b for a, b in zip(t_map, t_post)]
    if t_start_map1 is None or t_loop_done is None or len(per_q) != n_queries:
        raise SystemExit("fastANI_ref: unexpected stderr (%d per-query timers for %d queries)" % (len(per_q), n_queries))
    # what one more query costs a thread: the median of thread 0's own per-query timers (mapping + post mapping).  The loop-level figure
    # (arrival of 'parallel_for execution finished' - arrival of thread 0's 'Start Map 1') / queries is kept as a cross-check only: it
    # also holds the imbalance of the threads' SKETCH phases (the slowest thread starts its first query up to ~10 s after thread 0),
    # which is a fixed cost — read as a per-query cost of a 24-query sample it moved the extrapolation by +-15 % between runs on one box
    # (profiles/r05a_bench3_summary.txt), the median by +-2 %.
    m_loop = (t_loop_done - t_start_map1) / n_queries
    m_t0 = statistics.median(per_q)
    srt = sorted(per_q)
    return {"queries": n_queries, "wall_s": round(wall, 2), "thread0_sketch_timer_s": round(sketch_timer, 2) if sketch_timer is not None else None,
            "start_map1_at_s": round(t_start_map1, 2), "loop_done_at_s": round(t_loop_done, 2), "after_loop_s": round(wall - t_loop_done, 2),
            "marginal_s_per_query": m_t0, "fixed_s": wall - m_t0 * n_queries,
            "thread0_per_query_s": {"median": round(m_t0, 3), "mean": round(sum(per_q) / len(per_q), 3), "min": round(srt[0], 3), "p25": round(srt[len(srt) // 4], 3), "p75": round(srt[(3 * len(srt)) // 4], 3), "max": round(srt[-1], 3)},
            "loop_level_s_per_query": round(m_loop, 4), "marginal_crosscheck_ratio": round(m_t0 / m_loop, 3) if m_loop > 0 else None}


def cpu_legs(args, engine, params, rows, n_refs, query_ids, L):
    """rank 0, N = 1: FASTA copies of the set on local disk, the reference binary on a bounded sample, the drop-in CLI end to
    end, and the parity of the timed step's rows against both the reference's output and the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc   # oracle helpers (checker only): synthetic genome generator = CPU twin of ani_synth_packed, paths of oracle/_ref
    hi = host_info()
    out = {"host": hi}
    rows_by_pair = RowLookup(rows)
    parity = {}
    td = args.workdir or tempfile.mkdtemp(prefix="ani_bench_")
    os.makedirs(td, exist_ok=True)
    try:
        n_cpu_refs = min(args.cpu_refs or n_refs, n_refs)
        need_all = not args.no_e2e and args.config == "many-to-many"
        if args.config == "c4" and not args.cpu_refs:
            args.no_cpu_baseline = True                      # 10000 FASTA files = 50 GB: only on request (--cpu-refs N)
        n_queries = len(query_ids)
        want_ref = not args.no_cpu_baseline and os.path.exists(orc.REF_BIN)
        ids = list(range(n_refs if need_all else (n_cpu_refs if want_ref else 0)))
        t0 = time.time()
        paths = write_fasta_set(orc, args.seed, ids, L, td, hi["logical_cpus"], args.cluster_size)
        out["fasta_set"] = {"files": len(paths), "bytes": int(sum(os.path.getsize(p) for p in paths)), "seconds": round(time.time() - t0, 1), "dir": "local disk (%s)" % td}
        index_of = {p: i for i, p in enumerate(paths)}
        # ---- reference binary: every reference x a sample of the query genomes, ONE run.  The reference's cost is
        # T(Nq) = F + m * Nq: F = every thread sketches and indexes its split of the references, plus the serial re-read of all files for
        # the genome lengths (computeCoreIdentity.hpp:48-92) and the writers; m = one query mapped against every split.  Both come out of
        # the run's own stderr lines (core_genome_identity.cpp:60-70 "Time spent sketching the reference", :93-105 "Time spent mapping
        # fragments in query #i" / "Time spent post mapping") and their arrival times — see ref_timeline().  Rounds 3-4 took the slope
        # between two runs (2 and 10 queries), whose common 40-50 s sketch phase moved by more than the 15 s the slope was read from. ----
        if want_ref:
            all_q = [q for q in query_ids if q < len(paths)]
            q_hi = all_q[:args.cpu_queries]
            rl = os.path.join(td, "rl.txt")
            open(rl, "w").write("\n".join(paths[:n_cpu_refs]) + "\n")
            threads = max(1, min(args.cpu_threads, hi["physical_cores"]))
            ql = os.path.join(td, "ql_ref.txt")
            open(ql, "w").write("\n".join(paths[q] for q in q_hi) + "\n")
            ref_out = os.path.join(td, "ref.out")
            tl = ref_timeline([orc.REF_BIN, "--ql", ql, "--rl", rl, "-t", str(threads), "-o", ref_out], len(q_hi))
            m_q, fixed, t_hi = tl["marginal_s_per_query"], tl["fixed_s"], tl["wall_s"]
            t_full = fixed + m_q * n_queries
            tl["pairs_per_s_of_the_sample_itself"] = round(len(q_hi) * n_cpu_refs / t_hi, 1)
            tl["marginal_s_per_query"], tl["fixed_s"] = round(m_q, 4), round(fixed, 2)
            m_loop = tl["loop_level_s_per_query"]
            t_full_loop = (t_hi - m_loop * len(q_hi)) + m_loop * n_queries
            out["cpu_baseline"] = {"value": round(n_queries * n_cpu_refs / t_full, 3), "unit": "pairs/s", "cores": threads, "kind": "reference",
                                   "value_loop_level": round(n_queries * n_cpu_refs / t_full_loop, 3),
                                   "stated_baseline": "value (per-query cost = median of thread 0's own timers).  value_loop_level takes the per-query cost from the arrival times of "
                                                      "'Start Map 1' and 'parallel_for execution finished' instead, which also holds the threads' sketch-phase imbalance and is the LOWER "
                                                      "CPU figure; the higher one is the stated baseline because it is the conservative one for every GPU / CPU ratio",
                                   "cpu_model": hi["cpu_model"], "logical_cpus": hi["logical_cpus"], "physical_cores": hi["physical_cores"],
                                   "fixed_s": round(fixed, 2), "marginal_s_per_query": round(m_q, 3), "extrapolated_full_workload_s": round(t_full, 1),
                                   "measured": tl,
                                   "rule": "value = Nq x Nr / (F + m x Nq) with Nq = the workload's %d queries; m = median of thread 0's own per-query 'Time spent mapping fragments' + "
                                           "'Time spent post mapping' timers = what one more query costs a thread; F = wall - m x sampled queries (reference sketch + index per thread and "
                                           "their imbalance, genome lengths re-read, writers); cross-check: loop_level_s_per_query = (arrival of 'parallel_for execution finished' - arrival "
                                           "of thread 0's 'Start Map 1') / sampled queries, which also holds the sketch-phase imbalance" % n_queries,
                                   "threads_note": "-t %d: the thread count at which the reference is fastest on this host class (profiles/r03_refscale.txt)" % threads,
                                   "sample": "fastANI_ref -t %d, ONE run on %d query genomes x %d references of the same clustered %d bp set (80-column FASTA on local disk), wall %.1f s incl. FASTA parse; "
                                             "value = %d x %d pairs / (fixed %.1f s + %d x %.3f s per query)"
                                             % (threads, len(q_hi), n_cpu_refs, L, t_hi, n_queries, n_cpu_refs, fixed, n_queries, m_q)}
            q_ids = q_hi
            if not args.no_verify:
                parity["vs_reference_binary"] = compare_with_reference(rows_by_pair, read_ref_out(ref_out, index_of), set(q_ids), n_cpu_refs, L)
                parity["vs_reference_binary"]["what"] = "rows of the LAST TIMED STEP for query genomes %s vs fastANI_ref's output file on the same genomes" % q_ids
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/fastANI_ref not built"}
        # ---- oracle, pair by pair ----
        if not args.no_verify:
            parity["vs_oracle"] = oracle_spot_check(orc, args, rows_by_pair, n_refs, query_ids, L, params.windowSize, args.oracle_pairs)
        # ---- drop-in CLI, FASTA on disk -> output file ----
        cli = os.path.join(ROOT, "fastani_amd", "fastANI")
        if need_all and os.path.exists(cli):
            lst = os.path.join(td, "all.txt")
            open(lst, "w").write("\n".join(paths[:n_refs]) + "\n")
            e2e_out = os.path.join(td, "e2e.out")
            threads = min(hi["logical_cpus"], 64)
            t0 = time.time()
            r = subprocess.Popen([cli, "--ql", lst, "--rl", lst, "-t", str(threads), "-o", e2e_out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 env=dict(os.environ, ANI_CLI_TRACE="1"))
            err_lines, stamps = [], []
            for raw in r.stderr:                                  # the arrival time of every line: what lies before main() and behind its last line
                err_lines.append(raw.decode(errors="replace").rstrip("\n")); stamps.append(time.time() - t0)
            r.wait()
            t_e2e = time.time() - t0
            e2e = {"seconds": round(t_e2e, 2), "pairs_per_s": round(n_refs * n_refs / t_e2e, 1), "threads": threads, "returncode": r.returncode,
                   "what": "fastani_amd/fastANI --ql all --rl all (%d x %d FASTA files, %.1f GB, local disk) -> output file; first FASTA byte to output closed"
                           % (n_refs, n_refs, out["fasta_set"]["bytes"] / 1e9)}
            marks = [(st, float(ln.split()[2])) for ln, st in zip(err_lines, stamps) if ln.startswith("[fastANI trace]")]
            closed = [st for ln, st in zip(err_lines, stamps) if ln.startswith("[fastANI trace]") and "output written" in ln]
            if closed:
                # SURVEY.md section 8d's own end mark ("output file closed"), on the launcher's clock; `seconds` goes on until wait() returns
                e2e["seconds_to_output_closed"] = round(closed[-1], 3)
            if marks:
                e2e["outside_main"] = {"process_start_to_main_s": round(max(0.0, marks[0][0] - marks[0][1]), 3), "last_mark_to_exit_s": round(t_e2e - marks[-1][0], 3),
                                       "what": "the launcher's clock against the command line's own phase marks: loading the HIP runtime before main(), and the process going away after the output is closed (the command line parks a few threads at exit_group(), so the caller's wait() returns at once and the kernel releases the device and page-locked memory behind it: ANI_CLI_EXIT_THREADS, profiles/r08a_e2e_exit_ab.txt)"}
            tl = [ln for ln in err_lines if "Time spent sketching" in ln or "Time spent writing" in ln]
            if tl:
                e2e["stderr_timers"] = tl
            tr = [ln.replace("[fastANI trace]", "").strip() for ln in err_lines if ln.startswith("[fastANI trace]")]
            if tr:
                e2e["phases"] = tr
            pool = [ln for ln in err_lines if ln.startswith("[ani pool] hipMalloc")]          # only with ANI_POOL_TRACE=1 in the environment
            if pool:
                mb = [float(ln.split()[3]) for ln in pool]
                ms = [float(ln.split()[5]) for ln in pool]
                e2e["fresh_device_memory"] = {"hipMalloc_calls": len(pool), "GB": round(sum(mb) / 1024.0, 2), "seconds_inside_hipMalloc": round(sum(ms) / 1e3, 3),
                                              "largest": sorted(((round(a / 1024.0, 2), round(b, 1)) for a, b in zip(mb, ms)), reverse=True)[:12]}
            if r.returncode == 0 and not args.no_verify:
                printed = read_ref_out(e2e_out, index_of)
                want = {k: v for k, v in rows_by_pair.items() if trusted(v[1], L)}
                bad = sum(1 for k, v in printed.items() if k not in want or want[k][1:] != v[1:] or abs(want[k][0] - v[0]) > 1e-3)
                e2e["rows"] = len(printed)
                e2e["rows_equal_timed_step"] = bool(bad == 0 and len(printed) == len(want))
            out["end_to_end"] = e2e
    finally:
        if not args.workdir:
            shutil.rmtree(td, ignore_errors=True)
    if parity:
        parity["ok"] = all(v.get("ok", True) for v in parity.values() if isinstance(v, dict))
        out["parity_timed_rows"] = parity
    return out


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the library's sources: identifies the code a PMC traffic file was taken on (tools/pmc_traffic.py writes it)"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "fastani_amd", "csrc")
    names = []
    for d, _, files in os.walk(base):
        for f in files:
            if f.endswith((".hip", ".hpp", ".sh")):
                names.append(os.path.join(d, f))
    for fn in sorted(names):
        h.update(os.path.relpath(fn, base).encode()); h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def int_ops_block(c, steps, fused=False):
    """MurmurHash3 kernels against the VALU integer ceiling: 2 hashes (both strands) per k-mer start position."""
    u = UBENCH
    cyc_per_hash = u["mul_instr_per_hash"] * u["mul32_cyc"] + u["other_instr_per_hash"] * u["full_rate_cyc"]     # per wave of 64 hashes
    ceiling = u["simds"] * u["clock_ghz"] * 1e9 * 64.0 / cyc_per_hash
    out = {"ceiling_hashes_per_s": round(ceiling, 1),
           "ceiling_model": "per 64 hashes: %d 32-bit multiplier instr x %.2f cyc + %d other VALU instr x %.2f cyc (measured issue costs, tools/ubench/valu.hip) on 1 of %d SIMDs at %.1f GHz; hash arithmetic only: no base decoding, no winnowing, no sort"
                            % (u["mul_instr_per_hash"], u["mul32_cyc"], u["other_instr_per_hash"], u["full_rate_cyc"], u["simds"], u["clock_ghz"])}
    for name, ms, bases in (("ani::k_sketch_fused (hashes each genome once for both roles)" if fused else "ani::k_sketch_tiles", c["msSketch"], c["refBases"]),
                            ("ani::k_fragment_sketch", c["msFragSketch"], c["queryBases"])):
        if ms > 0:
            hps = 2.0 * bases / (ms / 1e3)
            out[name] = {"hashes_per_s": round(hps, 1), "mul64_per_s": round(8 * hps, 1), "frac_of_valu_ceiling": round(hps / ceiling, 4),
                         "ms_per_step": round(ms / steps, 3)}
    return out


class Run:
    """Everything a step needs: the process group (RCCL), the engine, the synthetic genomes in HBM, the staging buffers."""
    pass


def setup_runtime(args):
    import torch
    R = Run()
    R.args, R.torch = args, torch
    R.world = int(os.environ.get("WORLD_SIZE", "1"))
    R.rank = int(os.environ.get("RANK", "0"))
    R.local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and R.world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, R.world))
    R.dist = None
    # ANI_BENCH_BACKEND=emu (tests only, tests/test_distributed_gloo.py): the step logic of this file on the CPU build of the
    # product sources (tests/emu) over gloo, with tiny genomes — so that the multi-rank orchestration is exercised where there is
    # no GPU.  Never a measurement: the line it prints says so.
    R.emu = os.environ.get("ANI_BENCH_BACKEND", "") == "emu"
    # ANI_BENCH_FORCE_DIST=1 (hardware check on a 1-GPU box, tests/test_rccl_gpu.py): take the multi-rank code path — process group
    # over RCCL, the record all-gather or the fragment-set ring, the timing all-reduce — with a world of one rank
    R.multi = R.world > 1 or bool(os.environ.get("ANI_BENCH_FORCE_DIST"))
    if args.simulate_world > 1 and R.multi:
        raise SystemExit("--simulate-world is a single-process measurement")
    if R.multi:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # a bounded collective timeout: a rank that dies inside a step must turn the others' wait into an error, not a hang
        tmo = datetime.timedelta(seconds=int(os.environ.get("ANI_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if R.emu:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            torch.cuda.set_device(R.local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", R.local), timeout=tmo)
        R.dist = dist
    R.dev = torch.device("cpu") if R.emu else torch.device("cuda", R.local)
    if not R.emu:
        torch.cuda.set_device(R.dev)
    # host threads per rank: N ranks share the host, so each takes cores / N (the library's packing pool reads ANI_HOST_THREADS; torch's
    # intra-op pool would otherwise start one thread per core in every rank).  ANI_BENCH_THREADS_PER_RANK overrides.
    R.threads_per_rank = max(1, int(os.environ.get("ANI_BENCH_THREADS_PER_RANK", "0")) or (os.cpu_count() or 1) // max(1, R.world))
    if R.world > 1:
        os.environ.setdefault("ANI_HOST_THREADS", str(min(64, R.threads_per_rank)))
        torch.set_num_threads(max(1, min(R.threads_per_rank, 16)))
    import fastani_amd
    if R.emu:
        import ctypes
        from fastani_amd.api import Engine
        R.e = Engine(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libfastani_emu.so")), 0)
        args.no_cpu_baseline = args.no_e2e = args.no_verify = True
    else:
        R.e = fastani_amd.engine(R.local)
    R.p = R.e.params(16, 3000)
    return R


def dev_sync(R):
    if not R.emu:
        R.torch.cuda.synchronize()


def resolve_mode(R):
    """which step runs in the timed region:
         single    one GPU: sketch + index + map (all-vs-all: every genome hashed once for both roles)
         gather    N GPUs, WEAK scaling over the query stream (the default of the headline set at N > 1: north_star's own split, and what the task's
                   tier rules ask of a path that shards): every rank maps its own queries against the full index built from all-gathered records
         ring      N GPUs, STRONG scaling (--scaling strong; configs[3] / [4] shapes): the references are sharded, the query fragment sketches are
                   all-gathered (fastani_amd/multi_gpu.py)
         simulate  one GPU doing the compute of ONE rank of a W-rank ring job (no communication): the measured per-rank time behind the predicted N-GPU figure"""
    a = R.args
    if a.simulate_world > 1:
        return "simulate"
    if not R.multi:
        return "single"
    if a.config in ("c4", "c5"):
        return "ring"
    if a.config == "many-to-many" and a.scaling == "strong":
        return "ring"
    return "gather"


def make_inputs(R):
    """synthetic input, generated straight into HBM (untimed)"""
    import numpy as np
    from fastani_amd.api import DeviceGenomes
    a, e, torch = R.args, R.e, R.torch
    L = a.genome_len
    words = (L + 15) // 16
    cfg = a.config
    R.L, R.words = L, words
    R.NR = NR = a.genomes or {"c4": 10000, "c5": 30000}.get(cfg, 1000)
    R.mode = resolve_mode(R)
    W = R.world if R.mode != "simulate" else a.simulate_world
    r = R.rank if R.mode != "simulate" else a.simulate_rank
    R.W, R.r = W, r
    R.part_g0 = np.array([(NR * x) // W for x in range(W + 1)], dtype=np.int32)
    lo, hi = int(R.part_g0[r]), int(R.part_g0[r + 1])          # this rank's share of the references
    R.lo, R.hi = lo, hi
    R.contig_len = np.full(NR, L, dtype=np.int32)
    R.gcs = np.arange(NR + 1, dtype=np.int32)
    R.contig_len_local = np.full(hi - lo, L, dtype=np.int32)
    R.gcs_local = np.arange(hi - lo + 1, dtype=np.int32)
    R.qry_buf = None
    if cfg == "c5":
        # a reference set whose genomes do not stay resident: generated and sketched slice by slice through ONE slice buffer
        # (the minimizer records are what stays); the queries are the first --queries genomes of the set
        if R.mode != "single":
            raise SystemExit("--config c5 is a single-GPU run (on N GPUs configs[4] is the reference-sharded ring of --config c4 with more genomes)")
        R.nq_local = min(a.queries or 300, NR)
        R.slice_n = min(a.slice_genomes, NR)
        R.query_slice = max(1, a.query_slice)
        R.ref_block = a.ref_block
        if R.ref_block <= 0:
            # ~ 2 L / (w + 1) minimizers of 12 bytes per genome; a third of the device memory for the records leaves room for an
            # index chunk (36 bytes per minimizer of the chunk), the mapping pools and the slice buffers
            per_genome = 12.0 * 2.0 * L / 25.0 * 1.05
            total = torch.cuda.get_device_properties(R.dev).total_memory if not R.emu else 1 << 40
            R.ref_block = max(R.slice_n, int(total / 3 / per_genome) // R.slice_n * R.slice_n)
        R.ref_block = min(R.ref_block, NR)
        R.ref_buf = torch.empty(R.slice_n * words + 64, dtype=torch.int32, device=R.dev)
        R.query_block = min(a.query_block or R.nq_local, R.nq_local)
        R.qry_streamed = bool(a.query_block)              # the queries are generated slice by slice inside the step, like the references
        if R.qry_streamed:
            R.qry_buf = torch.empty(min(R.query_slice, R.nq_local) * words + 64, dtype=torch.int32, device=R.dev)
            R.qrys = None
        else:
            R.qry_buf = torch.empty(R.nq_local * words + 64, dtype=torch.int32, device=R.dev)
            e.synth_packed(a.seed, 0, R.nq_local, L, R.qry_buf.data_ptr(), variant=0, cluster_size=a.cluster_size)
            R.qrys = DeviceGenomes(R.qry_buf.data_ptr(), R.nq_local, L)
        R.drop_rows = bool(a.drop_rows)
        R.sample_queries = np.sort(np.random.default_rng(a.seed + 11).choice(R.nq_local, size=min(a.sample_queries, R.nq_local), replace=False)).astype(np.int32)
        R.refs = R.my_refs = None
        R.first_query_id = 0
        R.n_queries_total = R.nq_local
    elif cfg == "c4" and R.mode in ("single", "ring"):
        # all-vs-all: the rank's query genomes ARE its share of the references; only that share is resident
        R.nq_local = min(a.queries or (hi - lo), hi - lo)
        R.ref_buf = torch.empty((hi - lo) * words + 64, dtype=torch.int32, device=R.dev)
        e.synth_packed(a.seed, lo, hi - lo, L, R.ref_buf.data_ptr(), variant=0, cluster_size=a.cluster_size)
        R.my_refs = DeviceGenomes(R.ref_buf.data_ptr(), hi - lo, L)
        R.refs = R.my_refs if R.mode == "single" else None
        R.qrys = DeviceGenomes(R.ref_buf.data_ptr(), hi - lo, L, first=0, count=R.nq_local)
        R.first_query_id = lo
        R.n_queries_total = R.nq_local * W
    else:
        R.ref_buf = torch.empty(NR * words + 64, dtype=torch.int32, device=R.dev)
        e.synth_packed(a.seed, 0, NR, L, R.ref_buf.data_ptr(), variant=0, cluster_size=a.cluster_size)
        R.refs = DeviceGenomes(R.ref_buf.data_ptr(), NR, L)
        R.my_refs = DeviceGenomes(R.ref_buf.data_ptr(), NR, L, first=lo, count=hi - lo)
        if cfg == "one-to-many":
            R.nq_local = 1
            R.qrys = DeviceGenomes(R.ref_buf.data_ptr(), NR, L, first=1, count=1)     # cluster 0, member 1
            R.first_query_id = 1
            R.n_queries_total = R.nq_local * W
        elif R.mode in ("ring", "simulate"):
            # strong scaling of the all-vs-all set: the rank's queries are its share of the references
            R.nq_local = hi - lo
            R.qrys = R.my_refs
            R.first_query_id = lo
            R.n_queries_total = NR
        elif R.mode == "gather":
            R.nq_local = a.queries or NR
            alloc_variant_queries(R)
            R.first_query_id = R.rank * R.nq_local
            R.n_queries_total = R.nq_local * W
        else:
            R.nq_local = a.queries or NR
            R.qry_buf = R.ref_buf                                 # all-vs-all
            R.qrys = DeviceGenomes(R.qry_buf.data_ptr(), R.nq_local, L)
            R.first_query_id = 0
            R.n_queries_total = R.nq_local
    if R.mode in ("ring", "simulate"):
        R.nq_local = hi - lo
        R.n_queries_total = NR
    # all-vs-all on one GPU (the query set IS the reference set): every genome is hashed once for both roles
    R.self_slice = int(os.environ.get("ANI_BENCH_SELF_SLICE", "1000"))        # genomes per fused-pass slice of a large all-vs-all set (tests lower it)
    R.self_mode = (not a.no_self) and R.mode == "single" and ((cfg == "many-to-many" and not a.queries) or (cfg == "c4" and R.nq_local == NR))
    # per-rank timeline of a step (host clock; the library calls return when their device work is done):
    #   ref_records = reference sketching (the rank's share), fragsketch = fragment sketches of the rank's own queries,
    #   allgather = the part of the record all-gather that is NOT hidden behind the fragment sketching, index = index build,
    #   map = mapping + reduce + rows to the host, ring_wait = waiting for the next fragment set of the ring
    R.timers = {k: 0.0 for k in TIMER_KEYS}
    R.slot = 0
    R.allrec = None


TIMER_KEYS = ["ref_records_ms", "fragsketch_ms", "allgather_ms", "index_ms", "map_ms", "ring_pack_ms", "ring_wait_ms"]


def alloc_variant_queries(R):
    """weak scaling: rank r maps variant r of the clustered set (same ancestors and divergences, fresh substitutions)"""
    from fastani_amd.api import DeviceGenomes
    a = R.args
    R.qry_buf = R.torch.empty(R.nq_local * R.words + 64, dtype=R.torch.int32, device=R.dev)
    R.e.synth_packed(a.seed, 0, R.nq_local, R.L, R.qry_buf.data_ptr(), variant=R.rank, cluster_size=a.cluster_size)
    R.qrys = DeviceGenomes(R.qry_buf.data_ptr(), R.nq_local, R.L)


def ring_alloc_fn(R):
    def ring_alloc(nbytes):
        t = R.torch.empty(nbytes, dtype=R.torch.uint8, device=R.dev)
        return t, t.data_ptr()
    return ring_alloc


# ---------------------------------------------------------------------------------------------------------------------------
# the steps
# ---------------------------------------------------------------------------------------------------------------------------
def step_single(R):
    """one GPU: reference sketch + index (skch::Sketch), every query mapped (skch::Map) and reduced (cgi::computeCGI)"""
    import numpy as np
    from fastani_amd.api import DeviceGenomes, Sketch
    e, p, T = R.e, R.p, R.timers
    t_a = time.perf_counter()
    frags = None
    if R.args.config == "c5" or (R.self_mode and R.NR > 2 * R.self_slice):
        # a large set in slices of 1000 genomes (a slice's minimizers and sketch hashes are 32-bit counts): record parts (+ kept
        # fragment sets when the genomes are queries too), ONE index over the parts — handed over to the library, so a streamed set
        # keeps them without a copy — and ONE mapping call (a streamed set builds each chunk once)
        c5 = R.args.config == "c5"
        step_n = R.slice_n if c5 else R.self_slice
        # c5 beyond the memory that holds the set's RECORDS (12 bytes per minimizer: 60 000 genomes of 5 Mbp fill the GPU): the set
        # is taken in blocks of --ref-block genomes, each block sketched, indexed (streamed chunk by chunk), mapped by every query
        # and dropped — a (query, reference) result does not depend on what else is indexed, and the reference
        # does the same with its database splits (core_genome_identity.cpp:55-121, computeCoreIdentity.hpp:457-487)
        block = R.ref_block if c5 else R.NR
        out = []
        R.last_residency = None
        R.blocks_last_step = 0
        R.step_rows, R.step_crc = 0, 0
        worker = None
        if c5 and R.drop_rows:
            import queue
            import threading
            R.row_jobs = queue.Queue(maxsize=2)          # (two blocks' rows in flight at most: 8 GB each at 90 000 x 30 000)

            def bookkeeping():
                # checksum = sum of the rows' 32-bit words mod 2^64, at memory speed (crc32 runs at ~1 GB/s and the rows of 90 000 x 90 000 are
                # 126 GB); sampled queries by binary search: the rows of a call arrive ordered by query
                while True:
                    job = R.row_jobs.get()
                    if job is None:
                        return
                    rws, b0_ = job
                    R.step_rows += len(rws)
                    R.step_crc = (R.step_crc + int(np.ascontiguousarray(rws).view(np.uint32).sum(dtype=np.uint64))) & 0xffffffffffffffff
                    qcol = np.ascontiguousarray(rws["qryGenomeId"])
                    if len(qcol) and bool(np.all(qcol[:-1] <= qcol[1:])):
                        lo, hi = np.searchsorted(qcol, R.sample_queries, "left"), np.searchsorted(qcol, R.sample_queries, "right")
                        keep = np.concatenate([rws[a_:b_] for a_, b_ in zip(lo, hi)]) if len(lo) else rws[:0].copy()
                    else:
                        keep = rws[np.isin(qcol, R.sample_queries)]
                    out.append(keep)
                    del qcol, rws, job
            worker = threading.Thread(target=bookkeeping, daemon=True)
            worker.start()
        qblocks = [(q0, min(R.nq_local, q0 + R.query_block)) for q0 in range(0, R.nq_local, R.query_block)] if c5 else [(0, 0)]
        for qb0, qb1 in qblocks:
            qsets, qfirsts = [], []
            if c5:
                # the fragment sketches of one BLOCK of queries (--query-block; default: all of them), in sets of --query-slice genomes (a
                # set's sketch hashes are a 32-bit count: 2000 genomes of 5 Mbp hold 8 x 10^8); every reference block maps all sets of the
                # query block in ONE call, so a streamed reference block builds each of its index chunks once per query block
                t_q = time.perf_counter()
                for q0 in range(qb0, qb1, R.query_slice):
                    q1 = min(qb1, q0 + R.query_slice)
                    if R.qry_streamed:
                        e.synth_packed(R.args.seed, q0, q1 - q0, R.L, R.qry_buf.data_ptr(), variant=0, cluster_size=R.args.cluster_size)
                        dg = DeviceGenomes(R.qry_buf.data_ptr(), q1 - q0, R.L)
                    else:
                        dg = DeviceGenomes(R.qry_buf.data_ptr(), R.nq_local, R.L, first=q0, count=q1 - q0)
                    qsets.append(e.fragment_set(p, dg))
                    qfirsts.append(R.first_query_id + q0)
                T["fragsketch_ms"] += (time.perf_counter() - t_q) * 1e3
            for b0 in range(0, R.NR, block):
                b1 = min(R.NR, b0 + block)
                parts, sets, firsts = [], [], []
                t_a = time.perf_counter()
                for s0 in range(b0, b1, step_n):
                    s1 = min(b1, s0 + step_n)
                    if c5:
                        e.synth_packed(R.args.seed, s0, s1 - s0, R.L, R.ref_buf.data_ptr(), variant=0, cluster_size=R.args.cluster_size)
                        ptr, n = e.sketch_records(p, DeviceGenomes(R.ref_buf.data_ptr(), s1 - s0, R.L), s0 - b0)
                    else:
                        ptr, n, fr = e.sketch_records_self(p, DeviceGenomes(R.ref_buf.data_ptr(), R.NR, R.L, first=s0, count=s1 - s0), s0)
                        sets.append(fr); firsts.append(s0)
                    parts.append((ptr, n, s0 - b0))
                t_b = time.perf_counter()
                sk = Sketch(e, p, record_parts=([x[0] or 0 for x in parts], [x[1] for x in parts], [x[2] for x in parts] + [b1 - b0],
                                                R.contig_len[:b1 - b0], R.gcs[:b1 - b0 + 1]), adopt=True)
                t_d = time.perf_counter()
                sk.set_ref_id_base(b0)                    # rows with the set's global reference ids
                rows = sk.map_cgi_fragsets(qsets, qfirsts) if c5 else sk.map_cgi_fragsets(sets, firsts)
                if c5 and R.drop_rows:
                    # the step's rows are counted and checksummed block by block and only the sampled queries' rows stay (oracle check) —
                    # the bench's own bookkeeping, done by a worker thread beside the next block's GPU work (numpy releases the interpreter
                    # lock inside its loops): the rows are on the host when the mapping call returns, which is where a step's work ends.
                    # (First form, inline: 95 of the 353 s of the 90 000 x 90 000 step, profiles/r06e_bench_c5_90000x90000_one_gpu.json.log.)
                    R.row_jobs.put((rows, b0))
                    rows = None
                if rows is not None:
                    out.append(rows)
                t_e = time.perf_counter()
                res = sk.residency()
                R.last_residency = res if R.last_residency is None else dict(streaming=res["streaming"] or R.last_residency["streaming"],
                                                                             max_resident=max(res["max_resident"], R.last_residency["max_resident"]), resident_now=res["resident_now"])
                R.blocks_last_step += 1
                for fr in sets:
                    fr.close()
                sk.close()
                T["ref_records_ms"] += (t_b - t_a) * 1e3; T["index_ms"] += (t_d - t_b) * 1e3; T["map_ms"] += (t_e - t_d) * 1e3
            for fr in qsets:
                fr.close()
        if worker is not None:
            R.row_jobs.put(None)
            worker.join()                                 # the step ends when its last rows are counted
        # (several blocks: the rows stay block-major — (query, reference) order inside a block.  Sorting the 7 x 10^8 rows of
        #  90 000 x 10 000 on the host took longer than computing them: 55 of 104 s, profiles/r05c5_bench_c5_90000x10000.json.log)
        return out[0] if len(out) == 1 else out           # (several blocks: the parts, joined outside the timed region)
    if R.self_mode:
        # queries == references: one pass over the k-mer hashes gives the reference minimizers and the fragment sketches
        ptr, n, frags = e.sketch_records_self(p, R.refs, 0)
        t_b = time.perf_counter()
        sk = Sketch(e, p, records=(ptr, n, R.contig_len, R.gcs))
        if n:
            e.device_free(ptr)
    else:
        t_b = t_a
        sk = Sketch(e, p, R.refs)
    t_d = time.perf_counter()
    if frags is not None:
        rows = sk.map_cgi_fragset(frags, R.first_query_id)
        frags.close()
    else:
        rows = sk.map_cgi_batch(R.qrys, R.first_query_id)
    t_e = time.perf_counter()
    sk.close()
    T["ref_records_ms"] += (t_b - t_a) * 1e3; T["index_ms"] += (t_d - t_b) * 1e3; T["map_ms"] += (t_e - t_d) * 1e3
    return rows


def step_ring(R):
    """reference-sharded all-vs-all (fastani_amd/multi_gpu.py): this rank's genomes hashed once for both roles, indexed here and
    nowhere else; the ranks' packed fragment sets are all-gathered (ONE collective) while the rank maps its own set, then the N - 1
    foreign sets are mapped as one merged set (ANI_BENCH_EXCHANGE=ring: the round-3 ring, one hop and one mapping call per set).
    STRONG scaling: the job (NR x NR pairs) is fixed, every stage of it shards."""
    from fastani_amd.api import Sketch
    from fastani_amd.multi_gpu import gather_map, ring_map
    exchange = ring_map if os.environ.get("ANI_BENCH_EXCHANGE", "gather") == "ring" else gather_map
    e, p, T = R.e, R.p, R.timers
    t_a = time.perf_counter()
    ptr, n, frags = e.sketch_records_self(p, R.my_refs, 0)
    t_b = time.perf_counter()
    sk = Sketch(e, p, records=(ptr, n, R.contig_len_local, R.gcs_local))
    if n:
        e.device_free(ptr)
    t_d = time.perf_counter()
    rt = {}
    rows = exchange(e, sk, frags, R.part_g0, R.lo, R.dist, R.rank, R.world, ring_alloc_fn(R), lambda: dev_sync(R), rt, parts=True)
    frags.close()
    sk.close()
    T["ref_records_ms"] += (t_b - t_a) * 1e3; T["index_ms"] += (t_d - t_b) * 1e3
    T["map_ms"] += rt["map_ms"]; T["ring_pack_ms"] += rt["pack_ms"]; T["ring_wait_ms"] += rt["wait_ms"]
    return rows


def step_gather(R):
    """query-sharded (WEAK scaling over the query stream): rank r sketches its share of the references, ONE all-gather moves every
    rank's 12-byte records (the slot's first record carries the count) while the rank sketches the fragments of its own queries,
    then every rank builds the full index and maps its queries."""
    from fastani_amd.api import Sketch
    e, p, T, torch, dist = R.e, R.p, R.timers, R.torch, R.dist
    t_a = time.perf_counter()
    ptr, n = e.sketch_records(p, R.my_refs, R.lo)         # records with global seqIds
    # the gather slot is sized from the ranks' REAL counts (one tiny all-reduce), never from a density guess: no rank can find its
    # records too many for the slot and leave the others waiting in the collective
    m = torch.tensor([n], dtype=torch.int64, device=R.dev)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    need = int(m.item())
    if R.allrec is None or need > R.slot:
        R.slot = need + need // 16 + 1024                  # every rank sees the same maximum, so every rank makes the same decision
        R.allrec = torch.empty(R.world * (R.slot + 1) * 3, dtype=torch.int32, device=R.dev)
    slot, allrec = R.slot, R.allrec
    mine = allrec[R.rank * (slot + 1) * 3:(R.rank + 1) * (slot + 1) * 3]
    mine[:3].copy_(torch.tensor([n, 0, 0], dtype=torch.int32))
    if n:
        e.device_copy(mine.data_ptr() + 12, ptr, n * 12)
        e.device_free(ptr)
    dev_sync(R)                          # the library copies on its own stream
    t_r = time.perf_counter()
    work = dist.all_gather_into_tensor(allrec, mine, async_op=True)   # one collective: the reference sketch over RCCL/xGMI ...
    frags = e.fragment_set(p, R.qrys)                 # ... while this rank sketches the fragments of its own queries
    t_f = time.perf_counter()
    work.wait()
    dev_sync(R)
    t_c = time.perf_counter()
    counts = [int(x) for x in allrec[0::(slot + 1) * 3][:R.world].tolist()]
    ptrs = [allrec.data_ptr() + (r * (slot + 1) + 1) * 12 for r in range(R.world)]
    sk = Sketch(e, p, record_parts=(ptrs, counts, R.part_g0, R.contig_len, R.gcs))
    t_d = time.perf_counter()
    rows = sk.map_cgi_fragset(frags, R.first_query_id)
    frags.close()
    t_e = time.perf_counter()
    sk.close()
    T["ref_records_ms"] += (t_r - t_a) * 1e3; T["fragsketch_ms"] += (t_f - t_r) * 1e3; T["allgather_ms"] += (t_c - t_f) * 1e3
    T["index_ms"] += (t_d - t_c) * 1e3; T["map_ms"] += (t_e - t_d) * 1e3
    return rows


def prepare_simulation(R):
    """--simulate-world W: the fragment sets the other W - 1 ranks would contribute to the all-gather, made here (untimed) and packed
    into the slots of one buffer the way the collective leaves them"""
    from fastani_amd.api import DeviceGenomes
    e, p = R.e, R.p
    sets = []
    for x in range(R.W):
        g0, g1 = int(R.part_g0[x]), int(R.part_g0[x + 1])
        ptr, n, fr = e.sketch_records_self(p, DeviceGenomes(R.ref_buf.data_ptr(), R.NR, R.L, first=g0, count=g1 - g0), 0)
        if n:
            e.device_free(ptr)
        sets.append(fr)
    R.sim_cap = (max(fr.packed_bytes() for fr in sets) + 255) // 256 * 256
    R.sim_buf = R.torch.empty(R.sim_cap * R.W, dtype=R.torch.uint8, device=R.dev)
    for x, fr in enumerate(sets):
        fr.pack_into(R.sim_buf.data_ptr() + x * R.sim_cap, R.sim_cap)
        fr.close()
    dev_sync(R)


def step_simulate(R):
    """the compute of rank r of a W-rank job, alone on this GPU: its shard sketched and indexed, its own set packed into its slot and
    mapped, then the W - 1 foreign sets — already in their slots: no communication is timed — mapped as one merged set
    (ANI_BENCH_EXCHANGE=ring: one mapping call per set, the round-3 ring)"""
    from fastani_amd.api import FragmentSet, Sketch
    import numpy as np
    e, p, T = R.e, R.p, R.timers
    t_a = time.perf_counter()
    ptr, n, frags = e.sketch_records_self(p, R.my_refs, 0)
    t_b = time.perf_counter()
    sk = Sketch(e, p, records=(ptr, n, R.contig_len_local, R.gcs_local))
    if n:
        e.device_free(ptr)
    t_d = time.perf_counter()
    base = R.sim_buf.data_ptr()
    frags.pack_into(base + R.r * R.sim_cap, R.sim_cap)
    dev_sync(R)
    t_p = time.perf_counter()
    out = []
    sk.set_ref_id_base(R.lo)
    if os.environ.get("ANI_BENCH_EXCHANGE", "gather") == "ring":
        frags.close()
        for s in range(R.W):
            src = (R.r - s) % R.W
            view = FragmentSet.unpack(e, base + src * R.sim_cap, R.sim_cap, keepalive=R.sim_buf)
            out.append(sk.map_cgi_fragset(view, int(R.part_g0[src])))
            view.close()
    else:
        out.append(sk.map_cgi_fragset(frags, int(R.part_g0[R.r])))
        frags.close()
        merged = FragmentSet.unpack_merged(e, base, R.sim_cap, [int(R.part_g0[s]) if s != R.r else -1 for s in range(R.W)], keepalive=R.sim_buf)
        out.append(sk.map_cgi_fragset(merged, 0))
        merged.close()
    t_e = time.perf_counter()
    sk.close()
    T["ref_records_ms"] += (t_b - t_a) * 1e3; T["index_ms"] += (t_d - t_b) * 1e3; T["ring_pack_ms"] += (t_p - t_d) * 1e3; T["map_ms"] += (t_e - t_p) * 1e3
    return out                               # the mapping calls' arrays (global reference ids); joined outside the timed region


def dry_collectives(R, reps=10):
    """--dry-collectives: the collectives of the strong-scaling step with nothing around them.  Slot size = the packed fragment set of
    this rank's shard (~1.6 MB per 5 Mbp genome: 4 bytes per sketch hash + tables), agreed by the same one-word all-reduce the step
    uses; the all-gather is the in-place form of fastani_amd/multi_gpu.py: gather_map."""
    torch, dist = R.torch, R.dist
    est = int((R.hi - R.lo) * (R.L / 3000.0) * (240 * 4 + 16) + 4096)         # ~240 sketch hashes per 3 kb fragment
    times = {"all_reduce_s": [], "all_gather_s": [], "barrier_s": []}
    m = torch.tensor([est], dtype=torch.int64, device=R.dev)
    t0 = time.perf_counter(); dist.all_reduce(m, op=dist.ReduceOp.MAX); dev_sync(R); first_allreduce = time.perf_counter() - t0
    cap = (int(m.item()) + 255) // 256 * 256
    buf = torch.empty(cap * R.world, dtype=torch.uint8, device=R.dev)
    buf[R.rank * cap:(R.rank + 1) * cap].fill_(R.rank + 1)
    ok = True
    for i in range(reps):
        dev_sync(R)
        t0 = time.perf_counter(); dist.all_reduce(m, op=dist.ReduceOp.MAX); dev_sync(R); t1 = time.perf_counter()
        work = dist.all_gather_into_tensor(buf, buf[R.rank * cap:(R.rank + 1) * cap], async_op=True); work.wait(); dev_sync(R); t2 = time.perf_counter()
        dist.barrier(); dev_sync(R); t3 = time.perf_counter()
        times["all_reduce_s"].append(t1 - t0); times["all_gather_s"].append(t2 - t1); times["barrier_s"].append(t3 - t2)
        if i == 0:
            ok = all(int(buf[s * cap]) == s + 1 and int(buf[(s + 1) * cap - 1]) == s + 1 for s in range(R.world))
    mine = torch.tensor([min(times["all_reduce_s"]), min(times["all_gather_s"]), sorted(times["all_gather_s"])[len(times["all_gather_s"]) // 2], min(times["barrier_s"]), float(ok), first_allreduce],
                        dtype=torch.float64, device=R.dev)
    allv = torch.empty(R.world * mine.numel(), dtype=torch.float64, device=R.dev)
    dist.all_gather_into_tensor(allv, mine)
    g = allv.view(R.world, -1).tolist()
    recv = cap * (R.world - 1)
    if R.rank == 0:
        print(json.dumps({"dry_collectives": True, "n_gpus": R.world, "slot_bytes": cap, "bytes_received_per_rank": recv, "reps": reps,
                          "first_all_reduce_s": [round(x[5], 4) for x in g], "all_reduce_min_s": [round(x[0], 6) for x in g],
                          "all_gather_min_s": [round(x[1], 6) for x in g], "all_gather_median_s": [round(x[2], 6) for x in g],
                          "all_gather_GBps_received_per_rank": [round(recv / x[1] / 1e9, 2) if x[1] > 0 else None for x in g],
                          "barrier_min_s": [round(x[3], 6) for x in g], "slots_arrived_intact": [bool(x[4]) for x in g],
                          "threads_per_rank": R.threads_per_rank}), flush=True)


STEPS = {"single": step_single, "ring": step_ring, "gather": step_gather, "simulate": step_simulate}


# Rows of the standard jobs as (count, multiset hash): what a run on ANY number of ranks must add up to.  The values were taken from one-GPU
# runs whose rows were checked against fastANI_ref's output file and the oracle (parity_timed_rows); a (query, reference) result does
# not depend on how the references are sharded, so an N-rank run that reproduces the hash is checked against the
# reference by transitivity — the only row check a multi-rank hardware run can carry (its legs with the CPU reference are off).
# key: (config, reference genomes, genome length, cluster size, seed)
EXPECTED_ROWS = {
    ("many-to-many", 1000, 5_000_000, 20, 20260925): (785832, "ddc7eb890eaff1a3"),      # profiles/r08b_bench.json.log (parity_timed_rows ok in the same line)
}


def rows_multiset_hash(rows):
    """order- and partition-independent 64-bit hash of a set of result rows: every 20-byte row is mixed into one 64-bit word, the words
    are added modulo 2^64 — so the ranks' hashes of a sharded job simply add up to the hash of the whole job's rows"""
    import numpy as np
    if len(rows) == 0:
        return 0
    w = np.ascontiguousarray(rows).view(np.uint32).reshape(len(rows), -1).astype(np.uint64)
    ks = np.array([0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0xD6E8FEB86659FD93, 0xFF51AFD7ED558CCD,
                   0xA0761D6478BD642F, 0xE7037ED1A0B428DB, 0x8EBC6AF09C88C6E3][: w.shape[1]], dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (w * ks).sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        return int(h.sum(dtype=np.uint64))


def rows_multiset(R, rows, with_rank0=False):
    """(rows, hash) of the last timed step over ALL ranks: one all-reduce of the count and of the hash in 16-bit limbs.
    with_rank0: also rank 0's own (rows, hash) — in the query-sharded job rank 0 maps variant 0 of the set, i.e. the standard job"""
    n, h = len(rows), rows_multiset_hash(rows)
    n0, h0 = (n, h) if R.rank == 0 else (0, 0)
    if R.dist is not None:
        limbs = lambda x: [(x >> (16 * i)) & 0xffff for i in range(4)]                     # noqa: E731
        t = R.torch.tensor([n] + limbs(h) + [n0] + limbs(h0), dtype=R.torch.int64, device=R.dev)
        R.dist.all_reduce(t, op=R.dist.ReduceOp.SUM)
        v = [int(x) for x in t.tolist()]
        join = lambda w: sum(w[i] << (16 * i) for i in range(4)) & 0xffffffffffffffff      # noqa: E731
        n, h, n0, h0 = v[0], join(v[1:5]), v[5], join(v[6:10])
    return (n, h, n0, h0) if with_rank0 else (n, h)


def sync(R):
    dev_sync(R)
    if R.dist is not None:
        R.dist.barrier()
    dev_sync(R)


def settle(R, budget_s=5.0, need=5):
    """Before the warm-up: wait until the device is quiet.  A box may still be busy with what ran before this process — on hosts whose
    driver wipes freed device memory in the background a step that runs beside it was measured at 327 ms instead of 191
    (profiles/r08z_bench_first_step_disturbed.json.log: the benchmark started right behind the GPU test suite).  Probe = 1 GiB of
    read-modify-write (~0.5 ms alone), by HIP events on the current stream; quiet = `need` probes in a row within 1.5 x the fastest one
    seen.  At most budget_s seconds; nothing of the timed region is touched, and the line says what was seen."""
    if R.emu:
        return None
    torch = R.torch
    x = torch.zeros(1 << 28, dtype=torch.int32, device=R.dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, best, run, n, worst = time.perf_counter(), None, 0, 0, 0.0
    x.add_(1); torch.cuda.synchronize()
    while time.perf_counter() - t0 < budget_s:
        ev0.record(); x.add_(1); ev1.record(); ev1.synchronize()
        ms = ev0.elapsed_time(ev1)
        n += 1
        worst = max(worst, ms)
        best = ms if best is None else min(best, ms)
        run = run + 1 if ms <= 1.5 * best else 0
        if run >= need:
            break
        time.sleep(0.02)
    del x
    return {"probes": n, "waited_s": round(time.perf_counter() - t0, 3), "quiet": run >= need, "probe_ms_best": round(best or 0.0, 3), "probe_ms_worst": round(worst, 3),
            "what": "before the warm-up steps: 1 GiB read-modify-write probes until %d in a row run within 1.5 x the fastest (a device still busy with an earlier process's leftovers would disturb the first steps)" % need}


def timed_loop(R, step, steps, warmup):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device synchronisation; the job's time is the MAX over ranks"""
    import zlib
    import numpy as np
    for _ in range(warmup):
        step(R)
    R.e.reset_counters()
    for k in R.timers:
        R.timers[k] = 0.0
    sync(R)
    t0 = time.perf_counter()
    rows = None
    step_ms, crc = [], []
    trace = bool(os.environ.get("ANI_POOL_TRACE"))
    rows_all, dropped = [], []
    for i in range(steps):
        if trace:
            print("[bench] timed step %d" % i, file=sys.stderr, flush=True)
        ts = time.perf_counter()
        rows = step(R)                                        # returns with the rows on the host: the step's device work is done
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 2))
        rows_all.append(rows)
        if getattr(R, "drop_rows", False):
            dropped.append((R.step_rows, R.step_crc))
    sync(R)
    dt_local = time.perf_counter() - t0
    # a step may hand its rows back as the arrays of its mapping calls: one array from here on (outside the timed region)
    from fastani_amd.multi_gpu import join_rows
    rows_all = [join_rows(r) if isinstance(r, list) else r for r in rows_all]
    rows = rows_all[-1] if rows_all else rows
    # outside the timed region: every step must have produced the same rows (run-to-run determinism)
    crc = [zlib.crc32(np.ascontiguousarray(r).view(np.uint8)) & 0xffffffff for r in rows_all]          # (a view: no copy of the rows)
    del rows_all
    if dropped:                                              # --drop-rows: the steps' own counts and running checksums stand for the rows
        crc = [c for _, c in dropped]
        R.rows_last_step = dropped[-1][0]
    dt = dt_local
    if R.dist is not None:
        t = R.torch.tensor([dt], dtype=R.torch.float64, device=R.dev)
        R.dist.all_reduce(t, op=R.dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"rows": rows, "dt": dt, "dt_local": dt_local, "step_ms": step_ms, "rows_crc": crc}


def gather_rank_info(R, res, steps, mode):
    """per-rank timeline of the timed region, gathered on every rank (rank 0 prints it)"""
    torch, dist = R.torch, R.dist
    step_local = res["dt_local"] * 1e3 / steps
    vals = [step_local] + [R.timers[k] / steps for k in TIMER_KEYS]
    vals.append(step_local - sum(vals[1:]))                       # host time of the step outside every bracket
    vals.append(float(len(res["rows"])))
    mine_info = torch.tensor(vals, dtype=torch.float64, device=R.dev)
    nv = len(vals)
    gathered = torch.empty(R.world * nv, dtype=torch.float64, device=R.dev)
    dist.all_gather_into_tensor(gathered, mine_info)
    g = gathered.view(R.world, nv).tolist()
    info = {"ranks_seen_by_rccl": dist.get_world_size(), "mode": ("reference-sharded, fragment sets all-gathered (fastani_amd/multi_gpu.py: gather_map)" if os.environ.get("ANI_BENCH_EXCHANGE", "gather") != "ring" else "reference-sharded ring (fastani_amd/multi_gpu.py: ring_map)") if mode == "ring" else "query-sharded, reference records all-gathered",
            "timeline_note": "per rank, ms per step: ref_records = sketching the rank's references, fragsketch = its queries' fragment sketches (runs while the records are "
                             "all-gathered), allgather = the part of the gather not hidden behind it, index = index build, map = mapping + reduce + rows to the host, "
                             "ring_pack / ring_wait = packing the fragment set / waiting for the next set of the ring, other = the rest of the step",
            "step_ms": [round(x[0], 2) for x in g]}
    for i, k in enumerate(TIMER_KEYS + ["other_ms"]):
        info[k] = [round(x[1 + i], 2) for x in g]
    info["rows"] = [int(x[-1]) for x in g]
    info["bytes_moved_per_rank"] = (int(getattr(R.e, "_ring_bufs", (("", 0, 0),))[0][1]) * (R.world - 1)) if mode == "ring" else int((R.slot + 1) * 12 * (R.world - 1))
    return info


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks here — the same
    `torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` line the driver uses, on a free port, host threads
    split between the ranks — and leave with its exit code.  Rank 0 of the children prints the ONE JSON line.
    (The reference's only parallel split, core_genome_identity.cpp:55-121, is what the ranks mirror: bench.py: step_ring.)"""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or args.simulate_world > 1:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("ANI_BENCH_THREADS_PER_RANK", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    env.setdefault("OMP_NUM_THREADS", env["ANI_BENCH_THREADS_PER_RANK"])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("ANI_BENCH_BACKEND", "") != "emu":
        # fail before the rendezvous when the box has fewer devices than ranks (an 8-rank start on a 1-GPU box would otherwise end
        # in N - 1 ranks dying inside set_device and the survivor waiting for the store)
        probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, env=env)
        have = int((probe.stdout.decode().strip().splitlines() or ["0"])[-1]) if probe.returncode == 0 else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this box shows %d GPU(s)" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    self_launch(args)
    R = setup_runtime(args)
    import numpy as np
    e, p, cfg = R.e, R.p, args.config
    make_inputs(R)
    mode, NR, L, world, rank = R.mode, R.NR, R.L, R.world, R.rank
    if args.dry_collectives:
        if R.dist is None:
            raise SystemExit("--dry-collectives needs a process group (launch with torch.distributed.run, or ANI_BENCH_FORCE_DIST=1 for one rank)")
        dry_collectives(R)
        e.close()
        R.dist.destroy_process_group()
        return
    if mode == "simulate":
        prepare_simulation(R)
    R.settle = settle(R)
    res = timed_loop(R, STEPS[mode], args.steps, args.warmup)
    rows, dt = res["rows"], res["dt"]
    rank_info = gather_rank_info(R, res, args.steps, mode) if R.dist is not None else None
    # the rows of the last timed step over all ranks as (count, multiset hash); --drop-rows jobs keep no rows and carry their own checksums
    R.rows_multiset = rows_multiset(R, rows, with_rank0=True) if not getattr(R, "drop_rows", False) and mode != "simulate" else None
    c = e.counters()
    host_timeline = {k: round(v / args.steps, 2) for k, v in R.timers.items() if v}
    if args.dump_rows:
        np.save(args.dump_rows + (".rank%d" % rank if R.multi else "") + ".npy", rows)

    # N > 1, headline set: the timed region above is the STRONG-scaling job (fixed 1000 x 1000); the weak-scaling figure of the same
    # kernels (fixed database, 1000 queries per GPU) is measured beside it, outside the contract's timed region
    weak_leg = None
    if mode == "ring" and cfg == "many-to-many" and not args.no_weak_leg:
        # (the contract's figure is already in hand: whatever happens in this extra leg must not cost the line — a failure here is
        # reported inside it.  The state the report reads is put back afterwards.)
        keep = (R.nq_local, R.first_query_id)
        try:
            R.nq_local = args.queries or NR
            alloc_variant_queries(R)
            R.first_query_id = rank * R.nq_local
            wsteps = max(1, min(args.steps, 3))
            wres = timed_loop(R, step_gather, wsteps, 1)
            winfo = gather_rank_info(R, wres, wsteps, "gather")
            weak_leg = {"scaling": "weak", "value": round(NR * R.nq_local * world * wsteps / wres["dt"], 1), "unit": "pairs/s", "steps": wsteps, "ms_per_step": round(wres["dt"] / wsteps * 1e3, 2),
                        "workload": "many-to-many %dx%d: fixed %d-genome database, %d query genomes per GPU (rank r maps variant r of the clustered set); queries sharded, reference records all-gathered"
                                    % (NR, R.nq_local * world, NR, R.nq_local), "ranks": winfo}
        except Exception as ex:                              # noqa: BLE001 — any failure of the extra leg
            import traceback
            traceback.print_exc()
            weak_leg = {"scaling": "weak", "error": "%s: %s" % (type(ex).__name__, str(ex)[:400])}
        R.nq_local, R.first_query_id = keep[0], keep[1]

    # N > 1, headline set, default: the timed region above is the WEAK-scaling job (north_star's split).  The strong-scaling figure of the
    # same kernels — the fixed NR x NR job, references sharded — is measured beside it; its rows over all ranks are the standard job's
    # rows, so its rows_multiset is held against the stored value: the row check of a multi-rank hardware run.
    strong_leg = None
    if mode == "gather" and cfg == "many-to-many" and not args.no_strong_leg and not args.queries:
        keep = (R.nq_local, R.first_query_id, R.qrys, R.n_queries_total, R.mode)
        try:
            R.nq_local, R.qrys, R.first_query_id, R.n_queries_total, R.mode = R.hi - R.lo, R.my_refs, R.lo, NR, "ring"
            ssteps = max(1, min(args.steps, 3))
            sres = timed_loop(R, step_ring, ssteps, 1)
            sinfo = gather_rank_info(R, sres, ssteps, "ring")
            n_all, h_all = rows_multiset(R, sres["rows"])
            exp = EXPECTED_ROWS.get((cfg, NR, L, args.cluster_size, args.seed))
            strong_leg = {"scaling": "strong", "value": round(NR * NR * ssteps / sres["dt"], 1), "unit": "pairs/s", "steps": ssteps, "ms_per_step": round(sres["dt"] / ssteps * 1e3, 2),
                          "workload": "many-to-many %dx%d: the fixed job on %d GPU(s), references sharded, the ranks' fragment sketches all-gathered" % (NR, NR, world),
                          "rows_multiset": {"rows": int(n_all), "hash": "%016x" % h_all, "expected": {"rows": exp[0], "hash": exp[1]} if exp else None,
                                            "ok": (int(n_all) == exp[0] and "%016x" % h_all == exp[1]) if exp else None},
                          "rows_identical_across_steps": len(set(sres["rows_crc"])) == 1, "ranks": sinfo}
        except Exception as ex:                              # noqa: BLE001 — any failure of the extra leg is reported, it does not cost the line
            import traceback
            traceback.print_exc()
            strong_leg = {"scaling": "strong", "error": "%s: %s" % (type(ex).__name__, str(ex)[:400])}
        R.nq_local, R.first_query_id, R.qrys, R.n_queries_total, R.mode = keep

    # one-to-many: the latency-shaped number — map the one query against a resident index
    map_only = None
    if cfg == "one-to-many" and rank == 0:
        from fastani_amd.api import Sketch
        sk = Sketch(e, p, R.refs)
        sk.map_cgi_batch(R.qrys, R.first_query_id)
        dev_sync(R)
        ts = []
        for _ in range(max(3, args.steps)):
            t1 = time.perf_counter()
            sk.map_cgi_batch(R.qrys, R.first_query_id)
            ts.append((time.perf_counter() - t1) * 1e3)
        sk.close()
        map_only = {"ms": round(min(ts), 3), "ms_all": [round(x, 3) for x in ts], "what": "ani_map_cgi_batch of the one query genome (1666 fragments) against the resident 1000-genome index, rows on the host"}

    if rank == 0:
        out = report(R, args, res, c, rank_info, host_timeline, weak_leg, map_only, strong_leg)
        print(json.dumps(out), flush=True)
    e.close()
    if R.dist is not None:
        R.dist.destroy_process_group()


def report(R, args, res, c, rank_info, host_timeline, weak_leg, map_only, strong_leg=None):
    """the ONE JSON line of the driver contract (rank 0)"""
    e, p, cfg, mode, NR, L, world = R.e, R.p, args.config, R.mode, R.NR, R.L, R.world
    rows, dt = res["rows"], res["dt"]
    self_mode = R.self_mode
    n_queries_total, nq_local = R.n_queries_total, R.nq_local
    sim = mode == "simulate"
    pairs = NR * n_queries_total if not sim else NR * NR
    value = pairs * args.steps / dt
    fused = self_mode or mode in ("ring", "simulate")
    # roofline of the dominant kernel: algorithmic bytes / HIP-event time of its launches, timed region only
    l2_bytes = 12.0 * c["l2WindowEntries"] + 4.0 * c["l2QueryHashes"]
    cand = {
        "ani::k_l2_sim": (c["msL2Kernel"], l2_bytes - (12.0 * c["l2WindowEntriesB"] + 4.0 * c["l2QueryHashesB"]),
                          "class-A launches (k_l2_sim<L2Geom<255>>): 12 B x reference minimizers in the candidate range + 4 B x fragment sketch size, per class-A candidate"),
        "ani::k_l2_codes": (c["msL2Codes"], l2_bytes, "12 B x reference minimizers in the candidate range + 4 B x fragment sketch size, per candidate (the SAME bytes as k_l2_sim: the two kernels share one set of algorithmic bytes, see roofline.stage)"),
        "ani::k_l2": (c["msL2Slow"], l2_bytes * (c["l2SlowCandidates"] / max(1, c["l1Candidates"])), "general L2 kernel, share of the L2 bytes by candidate count"),
        "ani::k_l1_probe": (c["msL1Probe"], 4.0 * c["l1Probes"], "4 B x fragment sketch hashes probed (per index chunk)"),
        "ani::k_l1<0,2048>": (c["msL1Main"], 8.0 * c["seedHits"], "8 B x seed hits (all LDS classes; the small class handles nearly all fragments)"),
        "ani::k_l1_tiny": (c["msL1Tiny"], 4.0 * c["l1TinyFragments"] * 240.0, "4 B x ~240 probe results per fragment with <= 64 seed hits (one wave each)"),
        ("ani::k_sketch_fused" if fused else "ani::k_sketch_tiles"):
            (c["msSketch"], c["refBases"] / 4.0 + 12.0 * c["refMinimizers"] + (4.0 * c["querySketchHashes"] if fused else 0.0),
             "G/4 packed bases + 12 B x minimizers" + (" + 4 B x fragment sketch hashes (fused all-vs-all pass)" if fused else "")),
        "ani::k_fragment_sketch": (c["msFragSketch"], c["queryBases"] / 4.0 + 4.0 * c["querySketchHashes"], "G/4 packed bases + 4 B x sketch hashes"),
    }
    # whole-job figure of SURVEY.md §8d: B_total = N_r (G/4 + 36 M) + N_q (G/4 + 8 F s) + 8 H + sum(12 m_c + 4 s) + 112 #mappings
    # (#mappings bounded below by the candidates: the fused path does not count the survivors of the identity filter separately)
    b_total = (c["refBases"] / 4.0 + 36.0 * c["refMinimizers"] + c["queryBases"] / 4.0 + 8.0 * c["querySketchHashes"]
               + 8.0 * c["seedHits"] + l2_bytes)
    job = {"algorithmic_bytes_per_step": round(b_total / args.steps, 1), "bytes_per_pair": round(b_total / args.steps / max(1, NR * nq_local), 1),
           "achieved_GBs": round(b_total * world / dt / 1e9, 2), "frac_of_hbm_peak": round(b_total / dt / 1e9 / HBM_PEAK_GBS, 5),
           "note": "rank 0's counters; achieved_GBs scales them by the number of ranks"}
    dom = max(cand, key=lambda k: cand[k][0])
    ms, nbytes, what = cand[dom]
    achieved = nbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "kernel_achieved": round(achieved, 2), "kernel_frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes": what, "kernel_ms_per_step": round(ms / args.steps, 3),
            "all_kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in cand.items()},
            "all_kernels_achieved_GBs": {k: (round(v[1] / (v[0] / 1e3) / 1e9, 2) if v[0] > 0 else None) for k, v in cand.items()},
            "whole_job": job}
    # the L2 stage as a whole (ranges + length order + codes + simulation + leftovers) on the one set of L2 bytes
    if c["msL2"] > 0:
        st = l2_bytes / (c["msL2"] / 1e3) / 1e9
        roof["stage"] = {"stage": "L2 (k_l2_ranges + k_l2_len_* + k_l2_codes + k_l2_sim<A,B> + k_l2)", "ms_per_step": round(c["msL2"] / args.steps, 3),
                         "algorithmic_bytes_per_step": round(l2_bytes / args.steps, 1), "achieved": round(st, 2), "unit": "GB/s", "frac": round(st / HBM_PEAK_GBS, 5)}
        if dom in ("ani::k_l2_sim", "ani::k_l2_codes"):
            # the kernels of the L2 stage share ONE set of algorithmic bytes: quoting them per kernel counts the bytes twice, so the
            # headline fraction is the stage's (kernel_frac keeps the dominant kernel's own figure)
            roof["achieved"], roof["frac"] = roof["stage"]["achieved"], roof["stage"]["frac"]
            roof["frac_note"] = "achieved / frac = the L2 stage as a whole (its kernels share one set of algorithmic bytes); kernel_achieved / kernel_frac = the dominant kernel alone; whole_job = the step"
    roof["int_ops"] = int_ops_block(c, args.steps, fused)
    roof["bound_note"] = ("'hbm' by the algorithmic-byte accounting of SURVEY.md section 8d (12 B per reference minimizer in a candidate range); what "
                          "limits k_l2_sim and k_l2_codes is vector instruction issue (k_l2_sim: VALU wave-instructions x 4 cycles = 86 % of its SIMD cycles at 2.5 "
                          "waves per SIMD, profiles/r02X_pmc_sq_summary.txt); since round 4 the simulation of a chunk runs on a side stream beside the next chunk's "
                          "codes kernel (ANI_TEST_L2_OVERLAP=0 serialises them: stage +3.6 ms), so the two kernels' own durations — HIP events here, rocprofv3 in "
                          "profiles/ — are those of kernels that share the machine and add up to more than the stage; the stage figure (achieved / frac) is the one to read")
    # what binds, beside the accounting convention (`bound` keeps the contract's vocabulary: hbm | mfma)
    roof["accounting"] = "hbm"
    roof["binds"] = {"what": "vector instruction issue", "evidence": "profiles/r02X_pmc_sq_summary.txt (SQ counters: VALU wave-instructions x 4 cycles = 86 % of k_l2_sim's SIMD cycles), "
                                                                        "tools/ubench/valu.hip (every wave64 instruction ~4 SIMD cycles); exceptions: k_l1_probe (random reads, 0.75 of the "
                                                                        "measured random-read rate, tools/ubench/gather.hip) and the k_l1 gathers (memory-paced)"}
    # every stage against SURVEY.md section 8d's own bytes: sketch G/4 + 12 M (+ 4 F s, fused pass), index 24 M, L1 4 F s + 8 H, L2 sum(12 m_c + 4 s), reducer 112 B per mapping
    st_bytes = {"sketch": c["refBases"] / 4.0 + 12.0 * c["refMinimizers"] + (4.0 * c["querySketchHashes"] if fused else c["queryBases"] / 4.0 + 4.0 * c["querySketchHashes"]),
                "index": 24.0 * c["refMinimizers"], "L1": 4.0 * c["l1Probes"] + 8.0 * c["seedHits"], "L2": l2_bytes, "reduce": 112.0 * c["l1Candidates"]}
    st_ms = {"sketch": c["msSketch"] + c["msFragSketch"], "index": c["msIndex"], "L1": c["msL1"], "L2": c["msL2"], "reduce": c["msReduce"]}
    roof["stages"] = {k: {"algorithmic_bytes_per_step": round(st_bytes[k] / args.steps, 1), "ms_per_step": round(st_ms[k] / args.steps, 3),
                          "achieved": round(st_bytes[k] / (st_ms[k] / 1e3) / 1e9, 2) if st_ms[k] > 0 else None, "unit": "GB/s",
                          "frac": round(st_bytes[k] / (st_ms[k] / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if st_ms[k] > 0 else None, "counter_traffic_ratio": None} for k in st_bytes}
    roof["stages_note"] = ("bytes = SURVEY.md section 8d's per-stage figures from this run's counters (reduce: 112 B x candidates, an upper bound of the mappings); ms = HIP-event time of the stage "
                           "on its launch stream; counter_traffic_ratio = HBM bytes of the stage's kernels from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes")
    # HBM traffic from the committed PMC passes of the same workload (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, per launch).  The
    # passes are separate runs (rocprofv3 --pmc cannot ride in the timed run), so the file carries a hash of the kernel sources it was
    # taken on: it is quoted only when that is the hash of the sources of THIS run, and only for this default workload.
    try:
        if cfg == "many-to-many" and NR == 1000 and L == 5_000_000 and world == 1 and mode == "single":
            fn = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
            if os.path.exists(fn):
                tj = json.load(open(fn))
                cur = kernel_sources_sha16()
                if tj.get("kernel_sources_sha16") != cur:
                    roof["traffic_note"] = "profiles/r06_pmc_traffic.json was taken on kernel sources %s, this run is on %s: not quoted" % (tj.get("kernel_sources_sha16"), cur)
                else:
                    key = {"ani::k_l2_sim": "void ani::k_l2_sim<ani::L2Geom<255> >"}.get(dom, dom)
                    hit = [k for k in tj["kernels"] if k.startswith(key)]
                    if hit:
                        roof["traffic"] = tj["kernels"][hit[0]]["hbm_bytes_per_launch_corrected"]
                        roof["traffic_source"] = ("profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel sources %s = this run's)" % cur)
                    psteps = max(1, int(tj.get("steps", 1)))
                    for stg, tot in (tj.get("stage_hbm_bytes_total") or {}).items():
                        if stg in roof["stages"] and st_bytes[stg] > 0:
                            roof["stages"][stg]["counter_traffic_ratio"] = round((tot / psteps) / (st_bytes[stg] / args.steps), 2)
    except Exception as ex:
        roof["traffic_note"] = "traffic file unreadable: %r" % (ex,)
    if dom in ("ani::k_l2_sim", "ani::k_l2_codes"):          # one launch of each per L2 chunk
        launches = max(1, c["l2Launches"])
        roof.update({"launches": int(launches), "avg_launch_ms": round(ms / launches, 4),
                     "algorithmic_bytes_per_launch": round(nbytes / launches, 1)})
    stages = {k: round(c[k] / args.steps, 3) for k in ("msSketch", "msIndex", "msFragSketch", "msL1", "msL2", "msReduce")}
    wl = {"many-to-many": "many-to-many %dx%d" % (NR, n_queries_total), "one-to-many": "one-to-many 1x%d (query = cluster 0 member 1)" % NR,
          "c4": "many-to-many %dx%d (configs[3] shape, all-vs-all%s)" % (NR, n_queries_total, "" if n_queries_total == NR else ", query count bounded by --queries"),
          "c5": "many-to-many %dx%d (configs[4] shape: a reference set beyond the device memory's index capacity, streamed; the set is generated and sketched slice by slice, "
                "only the minimizer records of one block of references stay resident)" % (NR, n_queries_total)}[cfg]
    how = ""
    if sim:
        how = "; SIMULATED rank %d of a %d-rank reference-sharded ring job on one GPU (its compute only, no communication)" % (R.r, R.W)
    elif world > 1 or R.multi:
        how = ("; STRONG scaling: references sharded %d ways, query fragment sketches all-gathered over RCCL" % world if mode == "ring"
               else "; WEAK scaling: queries sharded %d ways (%d per GPU), reference sketch all-gathered over RCCL" % (world, nq_local))
    # `value` is the contract's figure: whole-job throughput with the inputs already resident in HBM when the timed region starts
    # (2-bit packed genomes); the wall-clock figure of SURVEY.md section 8d (first FASTA byte -> output file closed) is `end_to_end`
    out = {"metric": "ANI pairs/sec, many-to-many NxN ~5 Mbp genomes", "value": round(value, 1), "unit": "pairs/s",
           "value_is": "device-resident inputs (2-bit packed genomes in HBM at the start of the timed region; rows on the host at its end); end_to_end.pairs_per_s = FASTA on disk -> output file",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
           "higher_is_better": True,
           # the N = 1 line is the base of the series the N > 1 lines continue: per-GPU work fixed (weak) for the headline set unless --scaling strong
           "scaling": "weak" if mode == "gather" or (mode == "single" and cfg == "many-to-many" and args.scaling != "strong") else "strong", "vs_baseline": None, "dtype": "u32",
           "data": "synthetic" if not R.emu else "synthetic; CPU EMULATION OF THE KERNELS (test of the orchestration, not a measurement)",
           "config": {"workload": "%s synthetic %d bp genomes (clusters of %d, 0-25%% divergence), k=16 fragLen=3000 w=%d%s" % (wl, L, args.cluster_size, p.windowSize, how),
                      "name": cfg, "ref_genomes": NR, "query_genomes": n_queries_total, "genome_len": L, "inputs": "2-bit packed, resident in HBM" if cfg != "c5" else "2-bit packed, generated in HBM slice by slice inside the step",
                      "all_vs_all_single_hash_pass": bool(fused), "mode": mode,
                      "index_chunks": int(c["indexChunks"] // max(1, args.steps))},
           "rows_last_step": int(getattr(R, "rows_last_step", len(rows))), "rows_identical_across_steps": len(set(res["rows_crc"])) == 1, "step_ms_rank0": res["step_ms"],
           "stage_ms_per_step_rank0": stages, "host_timeline_ms_per_step_rank0": host_timeline, "l1_big_path": {"fragments_per_step": int(c["l1BigFragments"] // args.steps), "ms_per_step": round(c["msL1Big"] / args.steps, 3)},
           "counters_per_step_rank0": {k: int(c[k] // args.steps) for k in ("refMinimizers", "queryFragments", "seedHits", "l1Candidates",
                                                                          "l2WindowEntries", "l2Steps", "l2FastCandidates", "l2SlowCandidates",
                                                                          "l2SlowLimit", "l2SlowDup", "l2SlowOverflow", "cgiRows", "indexChunkBuilds", "l1Probes", "l1MidFragments", "l1TinyFragments")},
           "roofline": roof}
    if sim:
        out["simulated"] = {"world": R.W, "rank": R.r, "what": "value = %d x %d pairs / the time ONE rank of a %d-GPU strong-scaling job computes (sketch + index of its %d genomes, %d mapping calls: its own set, the others merged); "
                                                              "the all-gather of the packed fragment sets (%d bytes per rank) runs under the mapping of the rank's own set and is not in it"
                                                              % (NR, NR, R.W, R.hi - R.lo, 2, R.sim_cap),
                            "n_gpus_simulated": R.W}
    if getattr(R, "last_residency", None):
        out["config"]["residency"] = R.last_residency
    if cfg == "c5":
        out["config"]["ref_block"] = {"genomes": R.ref_block, "blocks": getattr(R, "blocks_last_step", 1),
                                      "what": "the reference set is indexed and mapped in blocks of this many genomes (records of one block resident at a time)"}
        out["config"]["query_block"] = {"genomes": R.query_block, "blocks": -(-R.nq_local // R.query_block), "queries_generated_inside_the_step": bool(R.qry_streamed),
                                        "what": "the fragment sets of one block of queries are resident together; the reference set is streamed once per query block"}
        if R.drop_rows:
            out["rows_kept"] = {"rows": int(len(rows)), "of_queries": [int(q) for q in R.sample_queries],
                                "what": "--drop-rows: a step's rows are counted and checksummed block by block (rows_last_step, rows_identical_across_steps); only these queries' rows stay for the oracle check"}
    if getattr(R, "rows_multiset", None) is not None:
        n_all, h_all, n_r0, h_r0 = R.rows_multiset
        std = EXPECTED_ROWS.get((cfg, NR, L, args.cluster_size, args.seed))
        exp = std if n_queries_total == NR and (mode in ("single", "ring") or (mode == "gather" and world == 1)) else None
        out["rows_multiset"] = {"rows": int(n_all), "hash": "%016x" % h_all, "ranks": world,
                                "expected": {"rows": exp[0], "hash": exp[1]} if exp else None,
                                "ok": (int(n_all) == exp[0] and "%016x" % h_all == exp[1]) if exp else None,
                                "what": "the rows of the last timed step over all ranks: count and an order- / partition-independent 64-bit hash (bench.py: rows_multiset_hash); "
                                        "expected = the same job's rows on one GPU, which were checked against fastANI_ref and the oracle (parity_timed_rows)"}
        if mode == "gather" and world > 1 and nq_local == NR and std:
            # query-sharded: rank 0 maps variant 0 of the set = the references themselves = the standard job, whatever the other ranks map
            out["rows_multiset"]["rank0"] = {"rows": int(n_r0), "hash": "%016x" % h_r0, "expected": {"rows": std[0], "hash": std[1]},
                                             "ok": int(n_r0) == std[0] and "%016x" % h_r0 == std[1]}
            out["rows_multiset"]["ok"] = out["rows_multiset"]["rank0"]["ok"]
    if rank_info:
        out["ranks"] = rank_info
    if weak_leg:
        out["weak_scaling_leg"] = weak_leg
    if strong_leg:
        out["strong_scaling_leg"] = strong_leg
    if map_only:
        out["map_only"] = map_only
    if world == 1 and mode == "single" and not R.multi and not (args.no_cpu_baseline and args.no_e2e and args.no_verify):
        qids = [int(q) for q in R.sample_queries] if getattr(R, "drop_rows", False) else list(range(R.first_query_id, R.first_query_id + nq_local))
        legs = cpu_legs(args, e, p, rows, NR, qids, L)
        out.update(legs)
    elif (mode == "simulate" or R.multi) and world == 1 and not args.no_verify and not R.emu:
        # the rows of a one-rank ring / gather / simulated-rank run against the oracle, pair by pair
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        rows_by_pair = RowLookup(rows)
        qids = list(range(NR)) if mode != "gather" else list(range(R.first_query_id, R.first_query_id + nq_local))
        out["parity_timed_rows"] = {"vs_oracle": oracle_spot_check(orc, args, rows_by_pair, R.hi - R.lo if mode != "gather" else NR, qids, L, p.windowSize, min(args.oracle_pairs, 120),
                                                                  ref_base=R.lo if mode != "gather" else 0)}
        out["parity_timed_rows"]["ok"] = out["parity_timed_rows"]["vs_oracle"]["ok"]
    if "cpu_baseline" not in out:
        out["cpu_baseline"] = None
    if out.get("end_to_end"):
        # SURVEY.md section 8d's wall-clock metric beside the contract's device-resident `value`
        out["value_wall"] = {"value": out["end_to_end"]["pairs_per_s"], "unit": "pairs/s", "seconds": out["end_to_end"]["seconds"],
                             "what": "the same 1000 x 1000 job through the command line: first FASTA byte read -> output file closed (end_to_end)"}
    import resource
    if getattr(R, "settle", None):
        out["settle"] = R.settle
    out["host_max_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 2)       # this process, checks included
    return out


if __name__ == "__main__":
    main()
