#!/bin/bash
# One GPU-box visit of round 4.  usage: tools/gpu_r04.sh <tag> [stage ...]
#   stages: tests bench prof ubench sim8 sim rccl o2m c4 c5 cluster pmc
# Writes everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
set -u
exec < /dev/null
TAG=${1:-r04}; shift || true
STAGES=${*:-tests bench prof}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has ubench; then
  echo "== radix sort microbenchmark" | tee "$OUT/ubench_radix.txt"
  timeout 300 tools/ubench/radix 2>&1 | tee -a "$OUT/ubench_radix.txt"
fi
if has profradix; then
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/profradix" -o radix --output-format csv -- "$REPO/tools/ubench/radix" 400000000 > "$OUT/profradix.log" 2>&1 < /dev/null)
  f=$(find "$OUT/profradix" -name "*kernel_stats*.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/radix_kernel_stats.csv"; cut -c1-240 "$f" | head -24; fi
  rm -rf "$OUT/profradix"
fi
if has pmcradix; then
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace -d "$OUT/pmcr_$name" -o pmc --output-format csv -- "$REPO/tools/ubench/radix" 400000000 > "$OUT/pmcr_$name.log" 2>&1)
    f=$(find "$OUT/pmcr_$name" -name "*counter_collection*.csv" | head -1)
    [ -n "$f" ] && python "$REPO/tools/pmc_summary.py" "$f" > "$OUT/pmcradix_${name}_summary.txt" 2>&1 && grep -E "radix|onesweep|Kernel|kernel" "$OUT/pmcradix_${name}_summary.txt" | cut -c1-250 | head -12
    rm -rf "$OUT/pmcr_$name"
  done
fi
if has l1tests; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/l1tests.log"
fi
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30; } | tee "$OUT/tests.log"
fi
if has quick; then
  echo "== bench (no cpu legs)"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify 2> "$OUT/quick.err" | tee "$OUT/quick.json.log" | cut -c1-900
  tail -3 "$OUT/quick.err"
fi
if has overlap; then
  echo "== bench with ANI_L2_OVERLAP=1 / =0 (A/B on one box)"
  for v in 1 0 1 0; do
    ANI_L2_OVERLAP=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$v', d['ms_per_step'], d['stage_ms_per_step_rank0'])" | tee -a "$OUT/overlap_ab.txt"
  done
fi
if has pair; then
  echo "== parity suite with the pair simulation kernel"
  ANI_L2_PAIR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pair_tests.log"
  echo "== bench with ANI_L2_PAIR=1 / =0 (A/B on one box)"
  for v in 1 0 1 0; do
    ANI_L2_PAIR=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('pair=$v', d['ms_per_step'], d['stage_ms_per_step_rank0'], {x: k[x] for x in ('ani::k_l2_sim','ani::k_l2_codes')})" | tee -a "$OUT/pair_ab.txt"
  done
  ANI_L2_PAIR=1 ANI_L2_OVERLAP=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('pair=1 overlap=0', d['ms_per_step'], d['stage_ms_per_step_rank0'], {x: k[x] for x in ('ani::k_l2_sim','ani::k_l2_codes')})" | tee -a "$OUT/pair_ab.txt"
  ANI_L2_PAIR=0 ANI_L2_OVERLAP=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('pair=0 overlap=0', d['ms_per_step'], d['stage_ms_per_step_rank0'], {x: k[x] for x in ('ani::k_l2_sim','ani::k_l2_codes')})" | tee -a "$OUT/pair_ab.txt"
fi
if has bench; then
  echo "== bench" | tee "$OUT/bench.log"
  timeout 900 python bench.py --steps 10 --warmup 2 2> "$OUT/bench.err" | tee "$OUT/bench.json.log" | cut -c1-900
  tail -3 "$OUT/bench.err"
fi
if has prof; then
  echo "== rocprofv3 kernel stats (same command, no cpu legs)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-verify > "$OUT/prof_bench.log" 2>&1)
  f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -32 "$f" | cut -c1-200
  find "$OUT/prof" -name "*kernel_trace*" -size +5M -delete 2>/dev/null
fi
if has profsim8; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/profsim" -o sim --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --simulate-world 8 --no-verify > "$OUT/profsim.log" 2>&1)
  f=$(find "$OUT/profsim" -name "*kernel_stats*.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/sim8_kernel_stats.csv"; cut -c1-200 "$f" | head -24; fi
  rm -rf "$OUT/profsim"
fi
if has sim8; then
  for w in 8; do
    echo "== simulated rank 0 of $w (strong scaling of the 1000 x 1000 set)"
    timeout 600 python bench.py --steps 5 --warmup 2 --simulate-world $w --oracle-pairs 60 2> "$OUT/sim$w.err" | tee "$OUT/bench_sim$w.json.log" | cut -c1-700
    tail -2 "$OUT/sim$w.err"
  done
fi
if has sim; then
  for w in 2 4; do
    echo "== simulated rank 0 of $w"
    timeout 600 python bench.py --steps 5 --warmup 2 --simulate-world $w --no-verify 2> "$OUT/sim$w.err" | tee "$OUT/bench_sim$w.json.log" | cut -c1-700
  done
fi
if has rccl; then
  for cfg in "many-to-many" "c4 --genomes 2000"; do
    echo "== one rank over RCCL: $cfg"
    ANI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 3 --warmup 1 --config $cfg --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/rccl.err" | tee "$OUT/rccl_one_rank_$(echo $cfg | cut -d' ' -f1).json.log" | cut -c1-600
    tail -2 "$OUT/rccl.err"
  done
fi
if has cluster; then
  for cs in 100 500; do
    echo "== bench --cluster-size $cs"
    timeout 900 python bench.py --steps 1 --warmup 1 --cluster-size $cs --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/bench_cs$cs.err" | tee "$OUT/bench_cs$cs.json.log" | cut -c1-400
    tail -2 "$OUT/bench_cs$cs.err"
  done
fi
if has c4; then
  echo "== c4 10000 x 300"
  timeout 900 python bench.py --config c4 --queries 300 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2> "$OUT/bench_c4.err" | tee "$OUT/bench_c4.json.log" | cut -c1-600
fi
if has c5; then
  echo "== c5: ${C5_GENOMES:-30000} references (streamed index) x 300 queries"
  timeout 1500 python bench.py --config c5 --genomes ${C5_GENOMES:-30000} --queries 300 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --oracle-pairs 60 2> "$OUT/bench_c5.err" | tee "$OUT/bench_c5.json.log" | cut -c1-900
  tail -3 "$OUT/bench_c5.err"
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
