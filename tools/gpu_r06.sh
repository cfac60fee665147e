#!/bin/bash
# One GPU-box visit of round 6.  usage: tools/gpu_r06.sh <tag> [stage ...]
#   stages: alloc radixbits libab e2e e2etrace tests quick dist1 bench3 c4one c4sim8 c5warm cli10k c5full sim prof pmc
# Writes everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
set -u
exec < /dev/null
TAG=${1:-r06}; shift || true
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
print('$1', d.get('value'), d.get('ms_per_step'), d.get('stage_ms_per_step_rank0'), {x: k[x] for x in ('ani::k_l2_sim', 'ani::k_l2_codes', 'ani::k_l1_probe', 'ani::k_l1<0,2048>') if x in k}, 'rows_ok', d.get('rows_identical_across_steps'))
"; }
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify"
if has alloc; then
  { echo "== fresh device memory on this box"; timeout 300 tools/ubench/alloc; } 2>&1 | tee "$OUT/ubench_alloc.txt"
fi
if has radixbits; then
  { echo "== radix bits per pass (rocPRIM onesweep as the yardstick)"; timeout 300 tools/ubench/radix_bits; } 2>&1 | tee "$OUT/ubench_radix_bits.txt"
fi
if has libab; then
  # A/B/A/B of two builds of the library on one box: LIB_VARIANTS="<tag> default"
  for v in ${LIB_VARIANTS:-default} ${LIB_VARIANTS:-default}; do
    if [ "$v" = default ]; then vv=""; else vv=$v; fi
    ANI_LIB_VARIANT=$vv timeout 300 python bench.py $QUICK 2>/dev/null | line "lib=$v" | tee -a "$OUT/libab.txt"
  done
fi
if has e2e; then
  echo "== command line end to end (1000 x 5 Mbp FASTA on local disk)"
  timeout 900 python tools/e2e_probe.py 1000 ${E2E_VARIANTS:-default} 2>&1 | tee "$OUT/e2e_probe.txt"
fi
if has e2etrace; then
  mkdir -p "$OUT/e2e_stderr"
  E2E_STDERR_DIR="$OUT/e2e_stderr" timeout 900 python tools/e2e_probe.py 1000 ANI_POOL_TRACE=1 2>&1 | tee "$OUT/e2e_trace.txt"
fi
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30; } | tee "$OUT/tests.log"
fi
if has quick; then
  echo "== bench (no cpu legs)"
  timeout 600 python bench.py $QUICK 2> "$OUT/quick.err" | tee "$OUT/quick.json.log" | line quick
  tail -3 "$OUT/quick.err"
fi
if has dist1; then
  echo "== one-rank RCCL through the self-launcher path"
  ANI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 $QUICK 2> "$OUT/dist1.err" | tee "$OUT/dist1.json.log" | line dist1
fi
if has bench3; then
  mkdir -p /tmp/ani_bench_wd
  for i in 1 2 3; do
    echo "== bench run $i"
    extra=""; [ $i -gt 1 ] && extra="--no-e2e"
    timeout 900 python bench.py --workdir /tmp/ani_bench_wd $extra 2> "$OUT/bench$i.err" > "$OUT/bench$i.json.log"
    python - "$OUT/bench$i.json.log" <<'PYEOF' | tee -a "$OUT/bench3_summary.txt"
import sys, json
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
cb, e2e = d.get('cpu_baseline') or {}, d.get('end_to_end') or {}
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline'].get('frac'), 'cpu', cb.get('value'), 'e2e', e2e.get('seconds'), 'parity', (d.get('parity_timed_rows') or {}).get('ok'))
PYEOF
  done
  rm -rf /tmp/ani_bench_wd
fi
if has c4one; then
  echo "== configs[3] on one GPU, warm"
  ANI_POOL_TRACE=1 timeout 1500 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c4one.err" | tee "$OUT/c4one.json.log" | cut -c1-1500
  grep "hipMalloc\|reserved" "$OUT/c4one.err" | awk '{mb+=$4; ms+=$6} END {print "c4 fresh device memory:", mb/1024, "GB,", ms/1000, "s inside hipMalloc"}' | tee "$OUT/c4one_pool.txt"
  grep -n "timed step" "$OUT/c4one.err" | head -3; grep -v "ani pool" "$OUT/c4one.err" | tail -3; gzip -f "$OUT/c4one.err"
fi
if has c4sim8; then
  echo "== configs[3]: rank 0 of 8 at 10 000 x 10 000"
  timeout 1500 python bench.py --config c4 --simulate-world 8 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c4sim8.err" | tee "$OUT/c4sim8.json.log" | cut -c1-1500
  tail -3 "$OUT/c4sim8.err"
fi
if has c5warm; then
  echo "== configs[4] reference side, warm: 90 000 x 5 Mbp references x 1000 queries, --steps 2 --warmup 1"
  ANI_POOL_TRACE=1 timeout 1500 python bench.py --config c5 --genomes 90000 --queries 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/c5warm.err" | tee "$OUT/c5warm.json.log" | cut -c1-1500
  grep "hipMalloc\|reserved" "$OUT/c5warm.err" | awk '{mb+=$4; ms+=$6} END {print "c5 fresh device memory:", mb/1024, "GB,", ms/1000, "s inside hipMalloc"}' | tee "$OUT/c5warm_pool.txt"
  grep -n "timed step" "$OUT/c5warm.err" | head -3; grep -v "ani pool" "$OUT/c5warm.err" | tail -3; gzip -f "$OUT/c5warm.err"
fi
if has cli10k; then
  # configs[4] through the COMMAND LINE at a size the box's disk holds: 10 000 x 5 Mbp references -> a 48 GB sketch file written in 4 blocks,
  # read back in >= 4 blocks (ANI_CLI_REF_BLOCK_BYTES), 2000 query genomes in >= 3 waves (ANI_CLI_QUERY_WAVE_BYTES), --matrix
  { df -h /tmp "$REPO" /dev/shm 2>&1; free -g | head -2; } | tee "$OUT/cli10k_disk.txt"
  mkdir -p /tmp/ani_c5_cli
  ANI_CLI_REF_BLOCK_BYTES=12000000000 ANI_CLI_QUERY_WAVE_BYTES=3500000000 timeout 2400 python tests/scale/c5_cli.py --genomes ${CLI_GENOMES:-10000} --queries ${CLI_QUERIES:-2000} --block ${CLI_BLOCK:-2500} \
      --oracle-pairs 60 --workdir /tmp/ani_c5_cli 2> "$OUT/cli10k.err" | tee "$OUT/cli10k.json.log" | cut -c1-2500
  tail -5 "$OUT/cli10k.err"; rm -rf /tmp/ani_c5_cli; df -h /tmp | tail -1
fi
if has c5full; then
  # configs[4] as stated: all-vs-all over 90 000 x 5 Mbp genomes = 8.1 x 10^9 pairs on ONE GPU — queries generated inside the step and taken in
  # blocks of 30 000 (48 GB of fragment sets), the references streamed block by block once per query block; rows counted + checksummed, 24 queries kept
  echo "== configs[4]: ${C5_GENOMES:-90000} x ${C5_QUERIES:-90000}, one cold step"
  timeout 2400 python bench.py --config c5 --genomes ${C5_GENOMES:-90000} --queries ${C5_QUERIES:-90000} --query-block ${C5_QBLOCK:-30000} --drop-rows --steps 1 --warmup 0 \
      --no-cpu-baseline --no-e2e --oracle-pairs 48 2> "$OUT/c5full.err" | tee "$OUT/c5full.json.log" | cut -c1-1800
  tail -3 "$OUT/c5full.err"; free -g | head -2
fi
if has sim; then
  for w in 8 4 2; do
    echo "== simulate-world $w, 1000 genomes"
    timeout 900 python bench.py --simulate-world $w --steps 5 --warmup 2 --no-verify 2> "$OUT/sim${w}_1000.err" | tee "$OUT/sim${w}_1000.json.log" | line "sim$w/1000"
  done
fi
if has prof; then
  echo "== rocprofv3 kernel stats (same command, no cpu legs)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/prof_bench.log" 2>&1)
  f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -28 "$f" | cut -c1-160
  rm -rf "$OUT/prof"
fi
if has pmc; then
  # HBM traffic of the step's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters + kernel trace only (the guide's recipe);
  # 3 profiled steps (--steps 2 --warmup 1); merged into the file bench.py quotes (it carries the hash of the kernel sources it was taken on)
  for set in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace -d "$OUT/pmc_$set" -o pmc --output-format csv -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/pmc_$set.log" 2>&1)
    f=$(find "$OUT/pmc_$set" -name "*counter_collection*.csv" | head -1)
    [ -n "$f" ] && python "$REPO/tools/pmc_summary.py" "$f" > "$OUT/pmc_${set}_summary.txt" 2>&1
    rm -rf "$OUT/pmc_$set"
  done
  python tools/pmc_traffic.py "$OUT/pmc_FETCH_SIZE_summary.txt" "$OUT/pmc_WRITE_SIZE_summary.txt" "$OUT/r06_pmc_traffic.json" 3 2>&1 | tail -2
fi
