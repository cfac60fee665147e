#!/bin/bash
# One GPU-box visit of round 5.  usage: tools/gpu_r05.sh <tag> [stage ...]
#   stages: tests paritytests bench3 quick prof trace c4sim8 c4one c5warm c5big disk ubench overlap sim8 ab pmc (cli90k: lost the box twice, do not run)
# Writes everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
set -u
exec < /dev/null
TAG=${1:-r05}; shift || true
STAGES=${*:-tests quick}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
k = d.get('roofline', {}).get('all_kernels_ms_per_step', {})
print('$1', d.get('value'), d.get('ms_per_step'), d.get('stage_ms_per_step_rank0'), {x: k[x] for x in ('ani::k_l2_sim', 'ani::k_l2_codes', 'ani::k_l1_probe', 'ani::k_l1') if x in k}, 'l2Steps', d.get('counters_per_step_rank0', {}).get('l2Steps'), 'rows_ok', d.get('rows_identical_across_steps'))
"; }
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify"

if has disk; then
  { echo "== disk"; df -h "$REPO" /tmp /dev/shm 2>&1; nproc; free -g | head -2
    timeout 120 dd if=/dev/zero of=/tmp/ddtest bs=1M count=6000 oflag=direct 2>&1 | tail -1
    timeout 120 dd if=/tmp/ddtest of=/dev/null bs=1M iflag=direct 2>&1 | tail -1; rm -f /tmp/ddtest; } | tee "$OUT/disk.txt"
fi
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30; } | tee "$OUT/tests.log"
fi
if has paritytests; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/paritytests.log"
fi
if has quick; then
  echo "== bench (no cpu legs)"
  timeout 600 python bench.py $QUICK 2> "$OUT/quick.err" | tee "$OUT/quick.json.log" | line quick
  tail -3 "$OUT/quick.err"
fi
if has bench3; then
  # the full default line three times on this box with ONE FASTA set: cpu_baseline must agree within +-15 %
  mkdir -p /tmp/ani_bench_wd
  for i in 1 2 3; do
    echo "== bench run $i"
    extra=""; [ $i -gt 1 ] && extra="--no-e2e"          # (the command-line leg once)
    timeout 900 python bench.py --workdir /tmp/ani_bench_wd $extra 2> "$OUT/bench$i.err" > "$OUT/bench$i.json.log"
    python - "$OUT/bench$i.json.log" <<'EOF' | tee -a "$OUT/bench3_summary.txt"
import sys, json
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
cb, e2e = d.get('cpu_baseline') or {}, d.get('end_to_end') or {}
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'cpu', cb.get('value'), 'F', cb.get('fixed_s'), 'm', cb.get('marginal_s_per_query'),
      'xcheck', (cb.get('measured') or {}).get('marginal_crosscheck_ratio'), 'e2e', e2e.get('seconds'), 'parity', (d.get('parity_timed_rows') or {}).get('ok'))
EOF
  done
  rm -rf /tmp/ani_bench_wd
fi
if has prof; then
  echo "== rocprofv3 kernel stats (same command, no cpu legs)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/prof_bench.log" 2>&1)
  f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -28 "$f" | cut -c1-160
  rm -rf "$OUT/prof"
fi
if has trace; then
  echo "== kernel timeline of the last index build (rocprofv3 --kernel-trace)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace" -o bench --output-format csv -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify ${TRACE_ARGS:-} > "$OUT/trace_bench.log" 2>&1)     # TRACE_ARGS: e.g. "--simulate-world 8"
  f=$(find "$OUT/trace" -name "*kernel_trace*.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'EOF2' | tee "$OUT/index_timeline.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_radix_histogram<unsigned int, ani::RecordSrc" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0][:64]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%9.3f %9.3f  %7.3f  q%s  %s" % (a, b, b - a, r.get("Queue_Id", "?"), n))
    if "k_l1_probe" in n:
        break
EOF2
  # the whole last step: every interval of >= 40 us in which NO kernel of any queue runs, with the kernel before and after it,
  # and the busy / idle totals (a step = from the last k_sketch_fused chain's first kernel to the last kernel of the trace)
  [ -n "$f" ] && python - "$f" <<'EOF3' | tee "$OUT/step_gaps.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][:56]
idx = max(i for i, r in enumerate(rows) if "k_radix_histogram<unsigned int, ani::RecordSrc" in r["Kernel_Name"])
# walk back to the start of the step: the first kernel after a gap of >= 5 ms before the index build's sketch kernels, or 60 ms back at most
start = idx
while start > 0 and int(rows[idx]["Start_Timestamp"]) - int(rows[start - 1]["Start_Timestamp"]) < 45e6 and "k_pair_reduce" not in rows[start - 1]["Kernel_Name"] and "k_oneway" not in rows[start - 1]["Kernel_Name"]:
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
cur_end, prev, idle, gaps = int(rows[start]["End_Timestamp"]), rows[start], 0, []
for r in rows[start + 1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        idle += s - cur_end
        if s - cur_end >= 40000:
            gaps.append(((cur_end - t0) / 1e6, (s - cur_end) / 1e6, name(prev), name(r)))
    if e > cur_end:
        cur_end, prev = e, r
total = (cur_end - t0) / 1e6
print("step of %.2f ms from %s: %.2f ms with no kernel running (%d gaps >= 40 us listed: at ms, length ms, after -> before)" % (total, name(rows[start]), idle / 1e6, len(gaps)))
for g in gaps:
    print("%9.3f  %6.3f  %s -> %s" % g)
import os
if os.environ.get("STEP_KERNELS"):
    print("-- every kernel of the step (start, end, ms, queue)")
    for r in rows[start:]:
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        print("%9.3f %9.3f  %7.3f  q%s  %s" % (a, b, b - a, r.get("Queue_Id", "?"), name(r)))
EOF3
  rm -rf "$OUT/trace"
fi
if has c4sim8; then
  echo "== configs[3] in its multi-rank form: rank 0 of 8 at 10 000 x 10 000 (1250-genome shard, own set + merged 8750-genome foreign set)"
  timeout 1500 python bench.py --config c4 --simulate-world 8 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c4sim8.err" | tee "$OUT/c4sim8.json.log" | cut -c1-1500
  tail -3 "$OUT/c4sim8.err"
fi
if has c4one; then
  echo "== configs[3] on one GPU, warm (for the ratio)"
  timeout 1500 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c4one.err" | tee "$OUT/c4one.json.log" | cut -c1-1500
  tail -3 "$OUT/c4one.err"
fi
if has sim8; then
  for g in ${SIM_GENOMES:-1000}; do
    for w in ${SIM_WORLDS:-8}; do
      echo "== simulate-world $w, $g genomes"
      timeout 900 python bench.py --simulate-world $w --genomes $g --steps 5 --warmup 2 --no-verify 2> "$OUT/sim${w}_$g.err" | tee "$OUT/sim${w}_$g.json.log" | line "sim$w/$g"
    done
    echo "== one GPU, $g genomes"
    timeout 900 python bench.py --genomes $g --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2> "$OUT/one_$g.err" | tee "$OUT/one_$g.json.log" | line "one/$g"
  done
fi
if has c5warm; then
  echo "== configs[4] reference side, warm: 90 000 x 5 Mbp references x 1000 queries, --steps 2 --warmup 1"
  ANI_POOL_TRACE=1 timeout 1500 python bench.py --config c5 --genomes 90000 --queries 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c5warm.err" | tee "$OUT/c5warm.json.log" | cut -c1-1500
  grep -c "hipMalloc" "$OUT/c5warm.err"; grep "hipMalloc" "$OUT/c5warm.err" | awk '{mb+=$4; ms+=$6} END {print "fresh device memory:", mb/1024, "GB,", ms/1000, "s inside hipMalloc"}' | tee "$OUT/c5warm_pool.txt"
  grep -n "timed step" "$OUT/c5warm.err" | head; grep "hipMalloc" "$OUT/c5warm.err" | sort -k6 -n -r | head -12 >> "$OUT/c5warm_pool.txt"
  gzip -f "$OUT/c5warm.err"
fi
if has c5big; then
  # configs[4] with a query count that makes the matrix real: 90 000 x 5 Mbp references x C5_QUERIES (default 10 000) queries on one GPU.
  # The queries' fragment sketches stay resident (1.6 GB per 1000 genomes), the references stream block by block, chunk by chunk.
  if [ -z "${C5_SKIP_SMALL:-}" ]; then
  echo "== configs[4]: 90 000 references x 1000 queries (the known-safe size, new query-set path), one cold step"
  timeout 600 python bench.py --config c5 --genomes 90000 --queries 1000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --oracle-pairs 20 2> "$OUT/c5_1000.err" | tee "$OUT/c5_1000.json.log" | cut -c1-1200
  tail -2 "$OUT/c5_1000.err"; free -g | head -2
  fi
  echo "== configs[4]: 90 000 references x ${C5_QUERIES:-10000} queries, one cold step"
  timeout 900 python bench.py --config c5 --genomes 90000 --queries ${C5_QUERIES:-10000} --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --oracle-pairs 20 2> "$OUT/c5_big.err" | tee "$OUT/c5_big.json.log" | cut -c1-1500
  tail -3 "$OUT/c5_big.err"; free -g | head -2
fi
if has ubench; then
  echo "== random reads / writes" | tee "$OUT/ubench_gather.txt"
  timeout 300 tools/ubench/gather 2>&1 | tee -a "$OUT/ubench_gather.txt"
fi
if has overlap; then
  echo "== L1 of one sub-batch beside L2 of another: two contexts on one device" | tee "$OUT/overlap_probe.txt"
  timeout 600 python tools/overlap_probe.py --slices 2 4 8 --reps 4 2>&1 | tail -30 | tee -a "$OUT/overlap_probe.txt"
fi
if has ab; then
  # A/B/A/B of an environment switch on one box: AB_VAR=<name> AB_VALUES="1 0" [AB_ARGS="--simulate-world 8"]
  for v in ${AB_VALUES:-1 0} ${AB_VALUES:-1 0}; do
    env ${AB_VAR:-ANI_L2_TRIM}=$v timeout 300 python bench.py $QUICK ${AB_ARGS:-} 2>/dev/null | line "${AB_VAR:-ANI_L2_TRIM}=$v" | tee -a "$OUT/ab_${AB_VAR:-ANI_L2_TRIM}.txt"
  done
fi
if has pmc; then
  IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS}"
  for set in "${SETS[@]}"; do
    name=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$OUT/pmc_$name" -o pmc --output-format csv -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/pmc_$name.log" 2>&1)
    f=$(find "$OUT/pmc_$name" -name "*counter_collection*.csv" | head -1)
    [ -n "$f" ] && python "$REPO/tools/pmc_summary.py" "$f" > "$OUT/pmc_${name}_summary.txt" 2>&1 && head -30 "$OUT/pmc_${name}_summary.txt" | cut -c1-200
    rm -rf "$OUT/pmc_$name"
  done
fi
if has cli90k; then
  echo "== the command line on a 90 000-genome sketch file"
  timeout 2400 python tests/scale/c5_cli.py ${CLI90K_ARGS:-} 2>&1 | tail -40 | tee "$OUT/cli90k.txt"
fi
echo "== done: $STAGES"
