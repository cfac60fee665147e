#!/bin/bash
# Round 6, second session: one box visit.  usage: tools/gpu_r07.sh <tag> [stage ...]
#   stages: exitprobe e2eexit quick dist1 tests bench1
set -u
exec < /dev/null
TAG=${1:-r07}; shift || true
STAGES=${*:-quick}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify"
ms() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', d.get('value'), d.get('ms_per_step'), d.get('stage_ms_per_step_rank0'), 'rows_multiset', d.get('rows_multiset'))
"; }
if has exitprobe; then
  { echo "== stamp -> process gone (tools/ubench/exit_probe)"; timeout 600 python tools/exit_probe.py; } 2>&1 | tee "$OUT/exit_probe.txt"
fi
if has e2eexit; then
  echo "== command line end to end, parked threads at the end A/B"
  E2E_REPS=${E2E_REPS:-4} timeout 1200 python tools/e2e_probe.py 1000 ${E2E_VARIANTS:-default ANI_TEST_EXIT_THREADS=96 default ANI_TEST_EXIT_THREADS=96} 2>&1 | cut -c1-420 | tee "$OUT/e2e_exit_ab.txt"
fi
if has quick; then
  echo "== bench (no cpu legs)"
  timeout 600 python bench.py $QUICK 2> "$OUT/quick.err" | tee "$OUT/quick.json.log" | ms quick
  tail -3 "$OUT/quick.err"
fi
if has dist1; then
  echo "== one-rank RCCL through the multi-rank path"
  ANI_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 $QUICK 2> "$OUT/dist1.err" | tee "$OUT/dist1.json.log" | ms dist1
  tail -3 "$OUT/dist1.err"
fi
if has tests; then
  { echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20; } | tee "$OUT/tests.log"
fi
if has bench1; then
  echo "== the default line"
  mkdir -p /tmp/ani_bench_wd
  timeout 900 python bench.py --workdir /tmp/ani_bench_wd 2> "$OUT/bench.err" > "$OUT/bench.json.log"
  ms bench < "$OUT/bench.json.log"; tail -3 "$OUT/bench.err"
  rm -rf /tmp/ani_bench_wd
fi
