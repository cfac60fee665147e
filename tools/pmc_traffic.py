"""Merge the FETCH_SIZE and WRITE_SIZE passes of tools/pmc_run.sh into the per-kernel HBM traffic file bench.py reads.
usage: python tools/pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <out.json>

Units and correction (see /opt/skills/guides/MI355X_MICROARCH.md, HBM section, and profiles/README.md): both counters are in
KiB on this rocprofv3; FETCH_SIZE reports half of a wide coalesced read stream on gfx950, so read bytes = 2 x FETCH_SIZE x 1024
(calibrated on kernels whose traffic is known exactly: k_synth_packed writes, k_index_split reads/writes)."""
import json, re, sys


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
                  "hbm_bytes_per_launch_corrected": round((2.0 * f + w) * 1024.0 / n, 1)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 1 --warmup 0 "
                     "--no-cpu-baseline (1000x1000x5Mbp), tools/pmc_run.sh + tools/pmc_traffic.py",
           "units": "counter values are KiB",
           "gfx950_correction": "read bytes = 2 x FETCH_SIZE x 1024; write bytes = WRITE_SIZE x 1024",
           "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(kernels), "kernels")
