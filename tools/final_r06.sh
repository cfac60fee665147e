#!/bin/bash
# the last visit of round 6: the default line once more on the committed code (traffic quoted from profiles/r06_pmc_traffic.json), one rank over RCCL
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06s
timeout 900 python bench.py 2> gpurun_out/r06s/bench.err > gpurun_out/r06s/bench.json.log
ANI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2> gpurun_out/r06s/dist1.err > gpurun_out/r06s/dist1.json.log
python - <<'PY'
import json
for f in ("bench", "dist1"):
    d = json.loads([l for l in open("gpurun_out/r06s/%s.json.log" % f) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["stage_ms_per_step_rank0"], "frac", r["frac"], "traffic", r.get("traffic"), r.get("traffic_note"), "e2e", (d.get("end_to_end") or {}).get("seconds"), "parity", (d.get("parity_timed_rows") or {}).get("ok"))
    print({k: v.get("counter_traffic_ratio") for k, v in r["stages"].items()})
PY
