#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats of the same bench command.
# usage: tools/gpu_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
{
  echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
  echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} | tee "$OUT/tests.log"
echo "== bench" | tee "$OUT/bench.log"
timeout 1500 python bench.py "$@" 2> "$OUT/bench.err" | tee -a "$OUT/bench.log"
tail -5 "$OUT/bench.err"
echo "== rocprofv3 kernel stats (same command, no cpu baseline)"
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- python "$REPO/bench.py" "$@" --no-cpu-baseline > "$OUT/prof_bench.log" 2>&1
cd "$REPO"
tail -2 "$OUT/prof_bench.log"
find "$OUT/prof" -name "*kernel_stats*" | head -3
f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -25 "$f"
# keep the merged-back payload small: drop the raw per-dispatch trace
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete 2>/dev/null
du -sh "$OUT" | tail -1
