"""Merge the FETCH_SIZE and WRITE_SIZE passes of tools/pmc_run.sh into the per-kernel HBM traffic file bench.py reads.
usage: python tools/pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <out.json> [steps of the profiled run: default 1]
The file carries the hash of the kernel sources it was taken on (bench.py: kernel_sources_sha16; bench.py quotes the traffic only when
that is the hash of the sources it runs on) and the per-stage totals behind roofline.stages[*].counter_traffic_ratio.

Units and correction (see /opt/skills/guides/MI355X_MICROARCH.md, HBM section, and profiles/README.md): both counters are in
KiB on this rocprofv3; FETCH_SIZE reports half of a wide coalesced read stream on gfx950, so read bytes = 2 x FETCH_SIZE x 1024
(calibrated on kernels whose traffic is known exactly: k_synth_packed writes, k_index_split reads/writes)."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stage_of(name):
    """the stage of the bench line a kernel belongs to (roofline.stages)"""
    n = name.replace("void ", "")
    if n.startswith("ani::k_l2"):
        return "L2"
    if n.startswith(("ani::k_l1", "ani::k_clamp_counts", "ani::k_finish_candidates", "ani::k_frag_order_keys")) or "k_radix_pass<unsigned long, unsigned int" in n:
        return "L1"
    if n.startswith(("ani::k_oneway_bins", "ani::k_pair_reduce")):
        return "reduce"
    if n.startswith(("ani::k_index", "ani::k_table", "ani::k_radix")):
        return "index"
    if n.startswith(("ani::k_sketch", "ani::k_expand", "ani::k_pack_fragment_pool", "ani::k_records_contig_first", "ani::k_fragment")):
        return "sketch"
    return "other"


def parse(path, ctr):
    out = {}
    for line in open(path):
        m = re.search(r"^(.*) %s=([0-9.e+]+)\(n=(\d+)\)" % ctr, line)
        if m:
            out[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
    n = max(nf, nw, 1)
    kernels[k] = {"FETCH_SIZE_KiB_total": f, "WRITE_SIZE_KiB_total": w, "launches": n,
                  "hbm_bytes_per_launch_corrected": round((2.0 * f + w) * 1024.0 / n, 1), "stage": stage_of(k)}
stages = {}
for k, v in kernels.items():
    stages[v["stage"]] = stages.get(v["stage"], 0.0) + (2.0 * v["FETCH_SIZE_KiB_total"] + v["WRITE_SIZE_KiB_total"]) * 1024.0
import bench
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 1 --warmup 0 "
                     "--no-cpu-baseline (1000x1000x5Mbp), tools/pmc_run.sh + tools/pmc_traffic.py",
           "units": "counter values are KiB",
           "gfx950_correction": "read bytes = 2 x FETCH_SIZE x 1024; write bytes = WRITE_SIZE x 1024",
           "steps": int(sys.argv[4]) if len(sys.argv) > 4 else 1, "kernel_sources_sha16": bench.kernel_sources_sha16(),
           "stage_hbm_bytes_total": {k: round(v, 1) for k, v in sorted(stages.items())},
           "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(kernels), "kernels")
