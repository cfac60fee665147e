#!/usr/bin/env python3
"""bench.py — ANI pairs/sec, many-to-many NxN ~5 Mbp genomes (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole hot path over the workload with the packed genomes already resident in HBM:
reference sketch + index build (skch::Sketch), mapping of every query genome (skch::Map) and the ANI reducer
(cgi::computeCGI), ending with the CGI rows on the host.

Workload (config.workload): BASELINE.json configs[2], "Many-to-many 1000x1000 synthetic ~5 Mbp genomes, k=16 fragLen=3000".
  N = 1 : 1000 reference genomes x 1000 query genomes (the same set: all-vs-all).
  N > 1 : weak scaling over the query stream — the 1000-genome reference database is fixed, every GPU maps its own
          1000 query genomes (rank r maps variant r of the clustered set: same ancestors and divergences, fresh
          substitutions), so the job is 1000 x (1000*N) pairs.  Each rank sketches 1/N of the references; the 12-byte
          minimizer records are all-gathered once over RCCL/xGMI and every rank builds the full index.

Prints ONE JSON line on rank 0 (see the driver contract): value = whole-job pairs/sec, plus
  roofline     — the dominant kernel (L2 sliding MinHash), algorithmic bytes (12*m_c + 4*s per candidate)
                 over its HIP-event time on the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline — the untouched reference (oracle/_ref/fastANI_ref, built from /root/reference by oracle/Makefile)
                 timed on this box's host cores on a bounded sample of the same clustered workload (rank 0, N = 1 only),
                 with the GPU timed on the identical sample and the two outputs compared.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="reference genomes (= query genomes per GPU)")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-refs", type=int, default=40)
    ap.add_argument("--cpu-queries", type=int, default=0, help="0 = pick from the core count")
    return ap.parse_args()


def cpu_baseline(args, engine, params):
    """Reference CPU path on a bounded sample + the GPU on the same sample + output comparison."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc   # oracle helpers: synthetic genome generator (CPU twin of ani_synth_packed) and paths of oracle/_ref
    from fastani_amd.api import Sketch
    if not os.path.exists(orc.REF_BIN):
        return {"value": None, "unit": "pairs/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/fastANI_ref not built"}
    cores = os.cpu_count() or 1
    nr = min(args.cpu_refs, args.genomes)
    nq = args.cpu_queries or 10
    nq = min(nq, nr)
    L = args.genome_len
    with tempfile.TemporaryDirectory(prefix="ani_cpu_") as td:
        genomes = []
        paths = []
        for g in range(nr):
            seq = orc.synth_genome(args.seed, g, L)
            genomes.append([seq])
            p = os.path.join(td, "g%d.fa" % g)
            orc.write_fasta(p, [seq], names=["g%d" % g])
            paths.append(p)
        with open(os.path.join(td, "rl.txt"), "w") as f:
            f.write("\n".join(paths) + "\n")
        with open(os.path.join(td, "ql.txt"), "w") as f:
            f.write("\n".join(paths[:nq]) + "\n")
        out = os.path.join(td, "ref.out")
        t0 = time.time()
        subprocess.check_call([orc.REF_BIN, "--ql", os.path.join(td, "ql.txt"), "--rl", os.path.join(td, "rl.txt"),
                               "-t", str(cores), "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_cpu = time.time() - t0
        ref_rows = {}
        for line in open(out):
            q, r, ani, cnt, tot = line.split("\t")
            ref_rows[(paths.index(q), paths.index(r))] = (float(ani), int(cnt), int(tot))
    # GPU on the identical sample, host FASTA bytes in (PCIe-inclusive), rows out
    t0 = time.time()
    sk = Sketch(engine, params, genomes)
    rows = sk.map_cgi_batch(genomes[:nq], 0)
    t_gpu = time.time() - t0
    sk.close()
    # fastANI prints a row iff countSeq*fragLen >= min(lenQ, lenR)*minFraction (computeCoreIdentity.hpp:328-332)
    glen = (L // 3000) * 3000
    got = {(int(r["qryGenomeId"]), int(r["refGenomeId"])): (float(r["identity"]), int(r["countSeq"]), int(r["totalQueryFragments"]))
           for r in rows if int(r["countSeq"]) * 3000 >= glen * np.float32(0.2)}
    ok = set(got) == set(ref_rows)
    max_dani = 0.0
    if ok:
        for key, (ani, cnt, tot) in ref_rows.items():
            g_ani, g_cnt, g_tot = got[key]
            ok = ok and cnt == g_cnt and tot == g_tot
            max_dani = max(max_dani, abs(ani - g_ani))   # the reference prints 6 significant digits
        ok = ok and max_dani <= 1e-3
    return {"value": round(nq * nr / t_cpu, 3), "unit": "pairs/s", "cores": cores, "kind": "reference",
            "sample": "%d queries x %d refs of the clustered %d bp set (genomes 0..%d), fastANI_ref -t %d, wall %.1f s incl. FASTA parse"
                      % (nq, nr, L, nr - 1, cores, t_cpu),
            "gpu_same_sample_pairs_per_s": round(nq * nr / t_gpu, 1), "gpu_same_sample_note": "host ASCII in, PCIe + 2-bit packing included",
            "rows_compared": len(ref_rows), "parity_vs_reference": "ok" if ok else "MISMATCH", "max_abs_ani_diff": round(max_dani, 6)}


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import numpy as np
    import fastani_amd
    from fastani_amd.api import DeviceGenomes, Sketch
    e = fastani_amd.engine(local)
    p = e.params(16, 3000)
    NG, L = args.genomes, args.genome_len
    words = (L + 15) // 16

    # ---- synthetic input, generated straight into HBM (untimed) ----
    ref_buf = torch.empty(NG * words + 64, dtype=torch.int32, device=dev)
    e.synth_packed(args.seed, 0, NG, L, ref_buf.data_ptr(), variant=0)
    if rank == 0 and world == 1:
        qry_buf = ref_buf                                   # all-vs-all
    else:
        qry_buf = torch.empty(NG * words + 64, dtype=torch.int32, device=dev)
        e.synth_packed(args.seed, 0, NG, L, qry_buf.data_ptr(), variant=rank)
    refs = DeviceGenomes(ref_buf.data_ptr(), NG, L)
    qrys = DeviceGenomes(qry_buf.data_ptr(), NG, L)
    lo, hi = (NG * rank) // world, (NG * (rank + 1)) // world
    my_refs = DeviceGenomes(ref_buf.data_ptr(), NG, L, first=lo, count=hi - lo)
    contig_len = np.full(NG, L, dtype=np.int32)
    gcs = np.arange(NG + 1, dtype=np.int32)

    def step():
        if world == 1:
            sk = Sketch(e, p, refs)
        else:
            ptr, n = e.sketch_records(p, my_refs, lo)         # records with global seqIds
            counts = torch.zeros(world, dtype=torch.int64, device=dev)
            counts[rank] = n
            dist.all_reduce(counts)
            cl = counts.tolist()
            mx = max(cl)
            mine = torch.zeros(mx * 3, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                          # the library copies on its own stream
            if n:
                e.device_copy(mine.data_ptr(), ptr, n * 12)
                e.device_free(ptr)
            allrec = torch.empty(world * mx * 3, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(allrec, mine)         # one collective: the reference sketch over RCCL/xGMI
            parts = [allrec[r * mx * 3: r * mx * 3 + cl[r] * 3] for r in range(world)]
            rec = torch.cat(parts) if world > 1 else parts[0]
            torch.cuda.synchronize()
            sk = Sketch(e, p, records=(rec.data_ptr(), int(sum(cl)), contig_len, gcs))
        rows = sk.map_cgi_batch(qrys, rank * NG)
        sk.close()
        return rows

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    e.reset_counters()
    sync()
    t0 = time.perf_counter()
    rows = None
    step_ms = []
    trace = bool(os.environ.get("ANI_POOL_TRACE"))
    for i in range(args.steps):
        if trace:
            print("[bench] timed step %d" % i, file=sys.stderr, flush=True)
        ts = time.perf_counter()
        rows = step()                                        # returns with the rows on the host: the step's device work is done
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 2))
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    c = e.counters()

    if rank == 0:
        pairs = NG * NG * world
        value = pairs * args.steps / dt
        # roofline of the dominant kernel: algorithmic bytes / HIP-event time of its launches, timed region only
        l2_bytes = 12.0 * c["l2WindowEntries"] + 4.0 * c["l2QueryHashes"]
        cand = {
            "ani::k_l2_sim": (c["msL2Kernel"], l2_bytes - (12.0 * c["l2WindowEntriesB"] + 4.0 * c["l2QueryHashesB"]),
                              "class-A launches (k_l2_sim<L2Geom<255>>): 12 B x reference minimizers in the candidate range + 4 B x fragment sketch size, per class-A candidate"),
            "ani::k_l2_codes": (c["msL2Codes"], l2_bytes, "12 B x reference minimizers in the candidate range + 4 B x fragment sketch size, per candidate"),
            "ani::k_l2": (c["msL2Slow"], l2_bytes * (c["l2SlowCandidates"] / max(1, c["l1Candidates"])), "general L2 kernel, share of the L2 bytes by candidate count"),
            "ani::k_l1": (c["msL1"], 4.0 * c["querySketchHashes"] + 8.0 * c["seedHits"], "4 B x fragment sketch hashes + 8 B x seed hits"),
            "ani::k_sketch_tiles": (c["msSketch"], c["refBases"] / 4.0 + 12.0 * c["refMinimizers"], "G/4 packed bases + 12 B x minimizers"),
            "ani::k_fragment_sketch": (c["msFragSketch"], c["queryBases"] / 4.0 + 4.0 * c["querySketchHashes"], "G/4 packed bases + 4 B x sketch hashes"),
        }
        # whole-job figure of SURVEY.md §8d: B_total = N_r (G/4 + 36 M) + N_q (G/4 + 8 F s) + 8 H + sum(12 m_c + 4 s) + 112 #mappings
        # (#mappings bounded below by the candidates: the fused path does not count the survivors of the identity filter separately)
        b_total = (c["refBases"] / 4.0 + 36.0 * c["refMinimizers"] + c["queryBases"] / 4.0 + 8.0 * c["querySketchHashes"]
                   + 8.0 * c["seedHits"] + l2_bytes)
        job = {"algorithmic_bytes_per_step": round(b_total / args.steps, 1), "bytes_per_pair": round(b_total / args.steps / (NG * NG), 1),
               "achieved_GBs": round(b_total * world / dt / 1e9, 2), "frac_of_hbm_peak": round(b_total / dt / 1e9 / HBM_PEAK_GBS, 5)}
        dom = max(cand, key=lambda k: cand[k][0])
        ms, nbytes, what = cand[dom]
        achieved = nbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes": what, "kernel_ms_per_step": round(ms / args.steps, 3),
                "all_kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in cand.items()},
                "all_kernels_achieved_GBs": {k: (round(v[1] / (v[0] / 1e3) / 1e9, 2) if v[0] > 0 else None) for k, v in cand.items()},
                "whole_job": job}
        # HBM traffic of the dominant kernel from the committed PMC passes of the same workload (FETCH_SIZE x2 gfx950 correction
        # + WRITE_SIZE, per launch; profiles/r01p_pmc_traffic.json) — only quoted when it is this default workload
        try:
            if NG == 1000 and L == 5_000_000 and world == 1:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01p_pmc_traffic.json")))
                key = {"ani::k_l2_sim": "void ani::k_l2_sim<ani::L2Geom<255> >"}.get(dom, dom)
                if key in tj["kernels"]:
                    roof["traffic"] = tj["kernels"][key]["hbm_bytes_per_launch_corrected"]
                    roof["traffic_source"] = "profiles/r01p_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            pass
        if dom == "ani::k_l2_sim":
            launches = max(1, c["l2Launches"])
            roof.update({"launches": int(launches), "avg_launch_ms": round(ms / launches, 4),
                         "algorithmic_bytes_per_launch": round(nbytes / launches, 1)})
        stages = {k: round(c[k] / args.steps, 3) for k in ("msSketch", "msIndex", "msFragSketch", "msL1", "msL2", "msReduce")}
        out = {"metric": "ANI pairs/sec, many-to-many NxN ~5 Mbp genomes", "value": round(value, 1), "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": "many-to-many %dx%d synthetic %d bp genomes (clusters of 20, 0-25%% divergence), k=16 fragLen=3000 w=%d%s"
                                      % (NG, NG * world, L, p.windowSize, "" if world == 1 else "; query stream sharded %d ways, reference sketch all-gathered over RCCL" % world),
                          "ref_genomes": NG, "query_genomes": NG * world, "genome_len": L, "inputs": "2-bit packed, resident in HBM"},
               "rows_last_step": int(len(rows)), "step_ms_rank0": step_ms,
               "stage_ms_per_step_rank0": stages,
               "counters_per_step_rank0": {k: int(c[k] // args.steps) for k in ("refMinimizers", "queryFragments", "seedHits", "l1Candidates",
                                                                              "l2WindowEntries", "l2Steps", "l2FastCandidates", "l2SlowCandidates",
                                                                              "l2SlowLimit", "l2SlowDup", "l2SlowOverflow", "cgiRows")},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, e, p)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
