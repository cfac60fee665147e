#!/bin/bash
# One GPU-box visit of round 2: parity tests, VALU microbenchmark, the three bench configurations, rocprofv3 kernel stats.
# usage: tools/gpu_r02.sh <tag> [what...]   what: tests ubench bench o2m c4 prof profo2m (default: all)
set -u
TAG=${1:-r02a}; shift || true
WHAT=${*:-tests ubench bench o2m c4 prof profo2m}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > "$OUT/tests_full.log" 2>&1; echo "rc=$?"; grep -v "^  File" "$OUT/tests_full.log" | tail -12; } | tee "$OUT/tests.log"
fi
if has ubench; then timeout 300 tools/ubench/valu > "$OUT/ubench_valu.txt" 2>&1; tail -22 "$OUT/ubench_valu.txt"; fi
if has bench; then
  echo "== bench (default)"; timeout 1500 python bench.py --steps 3 --warmup 1 2> "$OUT/bench.err" | tee "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has benchfast; then
  echo "== bench (default, no reference leg)"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workdir /tmp/ani_fa 2> "$OUT/bench.err" | tee "$OUT/bench.json" | cut -c1-300; tail -3 "$OUT/bench.err"
fi
if has refscale; then
  # how does the untouched reference scale with -t on this host? (1 query x 1000 references, FASTA in /tmp/ani_fa from benchfast)
  ls /tmp/ani_fa/g*.fa | head -1000 > /tmp/ani_fa/rl1000.txt; head -1 /tmp/ani_fa/rl1000.txt > /tmp/ani_fa/q1.txt; head -8 /tmp/ani_fa/rl1000.txt > /tmp/ani_fa/q8.txt
  TIMEFORMAT=%R
  for t in 16 32 64 128; do
    w=$( { time oracle/_ref/fastANI_ref --ql /tmp/ani_fa/q1.txt --rl /tmp/ani_fa/rl1000.txt -t $t -o /tmp/ani_fa/o.$t > /dev/null 2> "$OUT/ref_t$t.err"; } 2>&1 )
    echo "fastANI_ref 1x1000 -t $t: wall $w s" | tee -a "$OUT/refscale.txt"; grep -i "time spent" "$OUT/ref_t$t.err" | tail -3 >> "$OUT/refscale.txt"
  done
  for t in 32 64; do
    w=$( { time oracle/_ref/fastANI_ref --ql /tmp/ani_fa/q8.txt --rl /tmp/ani_fa/rl1000.txt -t $t -o /tmp/ani_fa/o.8 > /dev/null 2>&1; } 2>&1 )
    echo "fastANI_ref 8x1000 -t $t: wall $w s" | tee -a "$OUT/refscale.txt"
  done
fi
if has ab; then
  echo "== A/B: plain hipcc build vs assembly peephole (device-only, 3 steps)"
  for v in plain amd plain amd; do
    ANI_LIB_PATH=$REPO/fastani_amd/csrc/libfastani_$v.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], d['roofline']['all_kernels_ms_per_step'])" | tee -a "$OUT/ab.txt"
  done
fi
if has e2e; then
  # the command line alone, twice, with allocation and phase traces (FASTA set in /tmp/ani_fa from benchfast)
  ls /tmp/ani_fa/g*.fa | head -1000 > /tmp/ani_fa/all.txt
  for i in 1 2; do
    TIMEFORMAT=%R; w=$( { time ANI_POOL_TRACE=1 ANI_CLI_TRACE=1 fastani_amd/fastANI --ql /tmp/ani_fa/all.txt --rl /tmp/ani_fa/all.txt -t 64 -o /tmp/ani_fa/e2e.out > /dev/null 2> "$OUT/e2e_$i.err"; } 2>&1 )
    echo "fastANI 1000x1000 run $i: wall $w s" | tee -a "$OUT/e2e.txt"
    grep "fastANI trace" "$OUT/e2e_$i.err" | tee -a "$OUT/e2e.txt"
    grep "ani pool" "$OUT/e2e_$i.err" | awk '{mb+=$4; ms+=$6} END {print "hipMalloc calls", NR, "MB", mb, "ms", ms}' | tee -a "$OUT/e2e.txt"
  done
fi
if has noself; then
  echo "== bench (default workload, query role sketched separately)"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-verify --no-self 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('no-self', d['value'], d['ms_per_step'], d['stage_ms_per_step_rank0'], d['roofline']['all_kernels_ms_per_step'])" | tee "$OUT/noself.txt"
fi
if has c4w; then
  echo "== bench c4 (10000 refs, 300 queries, 1 GPU, warm)"; timeout 900 python bench.py --config c4 --queries 300 --steps 2 --warmup 1 --no-e2e 2> "$OUT/bench_c4.err" | tee "$OUT/bench_c4.json" | cut -c1-400; tail -3 "$OUT/bench_c4.err"
fi
if has o2m; then
  echo "== bench one-to-many"; timeout 600 python bench.py --config one-to-many --steps 5 --warmup 1 --no-e2e 2> "$OUT/bench_o2m.err" | tee "$OUT/bench_o2m.json"; tail -3 "$OUT/bench_o2m.err"
fi
if has c4; then
  echo "== bench c4 (10000 refs, 200 queries, 1 GPU)"; timeout 900 python bench.py --config c4 --queries 200 --steps 1 --warmup 0 --no-e2e 2> "$OUT/bench_c4.err" | tee "$OUT/bench_c4.json"; tail -3 "$OUT/bench_c4.err"
fi
prof() {  # name, bench args...
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o bench --output-format csv -- python "$REPO/bench.py" "$@" --no-cpu-baseline --no-e2e --no-verify > "$OUT/prof_$name.log" 2>&1)
  tail -1 "$OUT/prof_$name.log" | cut -c1-400
  local f=$(find "$OUT/prof_$name" -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv" && head -12 "$f" | cut -c1-200
  find "$OUT/prof_$name" -name "*kernel_trace*" -size +20M -delete 2>/dev/null
}
if has prof; then echo "== rocprofv3 kernel stats (default workload)"; prof m2m --steps 3 --warmup 1; fi
if has profo2m; then echo "== rocprofv3 kernel stats (one-to-many)"; prof o2m --config one-to-many --steps 5 --warmup 1; fi
if has pmc; then
  echo "== PMC passes (FETCH_SIZE, WRITE_SIZE separately; SQ counters at 200 genomes)"
  bash tools/pmc_run.sh ${TAG}_pmc_fetch FETCH_SIZE --steps 1 --warmup 0 > /dev/null 2>&1; tail -3 gpurun_out/${TAG}_pmc_fetch/pmc.log | cut -c1-200
  bash tools/pmc_run.sh ${TAG}_pmc_write WRITE_SIZE --steps 1 --warmup 0 > /dev/null 2>&1
  bash tools/pmc_run.sh ${TAG}_pmc_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" --steps 1 --warmup 0 --genomes 200 > /dev/null 2>&1
  bash tools/pmc_run.sh ${TAG}_pmc_sqwait "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" --steps 1 --warmup 0 --genomes 200 > /dev/null 2>&1
  python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch/pmc_summary.txt gpurun_out/${TAG}_pmc_write/pmc_summary.txt gpurun_out/${TAG}_pmc_traffic.json
  head -12 gpurun_out/${TAG}_pmc_sq/pmc_summary.txt | cut -c1-330
  rm -rf gpurun_out/${TAG}_pmc_fetch/pmc gpurun_out/${TAG}_pmc_write/pmc gpurun_out/${TAG}_pmc_sq/pmc gpurun_out/${TAG}_pmc_sqwait/pmc
fi
du -sh "$OUT" | tail -1
