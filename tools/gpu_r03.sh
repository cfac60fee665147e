#!/bin/bash
# One GPU-box visit of round 3.  usage: tools/gpu_r03.sh <tag> [stage ...]   stages: tests bench prof cluster c4big o2m pmc
# Writes everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
set -u
TAG=${1:-r03}; shift || true
STAGES=${*:-tests bench prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30; } | tee "$OUT/tests.log"
fi
if has bench; then
  echo "== bench" | tee "$OUT/bench.log"
  timeout 900 python bench.py --steps 10 --warmup 2 2> "$OUT/bench.err" | tee "$OUT/bench.json.log" | cut -c1-600
  tail -3 "$OUT/bench.err"
fi
if has prof; then
  echo "== rocprofv3 kernel stats (same command, no cpu legs)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/prof_bench.log" 2>&1)
  f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -30 "$f" | cut -c1-200
  find "$OUT/prof" -name "*kernel_trace*" -size +5M -delete 2>/dev/null
fi
if has cluster; then
  for cs in 100 500; do
    echo "== bench --cluster-size $cs"
    timeout 900 python bench.py --steps 1 --warmup 1 --cluster-size $cs --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/bench_cs$cs.err" | tee "$OUT/bench_cs$cs.json.log" | cut -c1-400
    tail -2 "$OUT/bench_cs$cs.err"
  done
fi
if has c4big; then
  echo "== 20000 references on one GPU (streamed index), 300 queries"
  timeout 900 python bench.py --config c4 --genomes 20000 --queries 300 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-verify 2> "$OUT/bench_c4_20000.err" | tee "$OUT/bench_c4_20000.json.log" | cut -c1-600
  tail -3 "$OUT/bench_c4_20000.err"
  echo "== c4 10000 x 300"
  timeout 900 python bench.py --config c4 --queries 300 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2> "$OUT/bench_c4.err" | tee "$OUT/bench_c4.json.log" | cut -c1-600
fi
if has o2m; then
  echo "== one-to-many"
  timeout 600 python bench.py --config one-to-many --steps 5 --warmup 1 --no-e2e 2> "$OUT/bench_o2m.err" | tee "$OUT/bench_o2m.json.log" | cut -c1-400
fi
if has pmc; then
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    name=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace -d "$OUT/pmc_$name" -o pmc --output-format csv -- python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-verify > "$OUT/pmc_$name.log" 2>&1)
    f=$(find "$OUT/pmc_$name" -name "*counter_collection*.csv" | head -1)
    [ -n "$f" ] && python "$REPO/tools/pmc_summary.py" "$f" > "$OUT/pmc_${name}_summary.txt" 2>&1 && head -20 "$OUT/pmc_${name}_summary.txt"
    rm -rf "$OUT/pmc_$name"
  done
fi
du -sh "$OUT" | tail -1
