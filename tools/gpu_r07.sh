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
if has initprobe; then
  { echo "== start of the HIP runtime, call by call (tools/ubench/init_probe): idle device (2 s pause before each run)"
    for i in 1 2 3; do sleep 2; echo "run $i"; timeout 60 tools/ubench/init_probe; done
    echo "== back to back (no pause)"
    for i in 1 2 3; do echo "run $i"; timeout 60 tools/ubench/init_probe; done
    echo "== the library's own start: dlopen of libfastani_amd.so, ani_init"
    python - <<'PYEOF'
import time, ctypes, os
t0 = time.time(); lib = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "fastani_amd/csrc/libfastani_amd.so")); print("  dlopen libfastani_amd.so %.1f ms" % ((time.time() - t0) * 1e3))
ctx = ctypes.c_void_p(); t0 = time.time(); rc = lib.ani_init(0, ctypes.byref(ctx)); print("  ani_init(0) rc %d  %.1f ms" % (rc, (time.time() - t0) * 1e3))
t0 = time.time(); rc = lib.ani_init(0, ctypes.byref(ctx)); print("  second ani_init(0) rc %d  %.1f ms" % (rc, (time.time() - t0) * 1e3))
os._exit(0)
PYEOF
  } 2>&1 | tee "$OUT/init_probe.txt"
fi
if has exitcli; then
  echo "== command line end to end: parked threads at the end (default 16) against none"
  E2E_REPS=${E2E_REPS:-4} timeout 1200 python tools/e2e_probe.py 1000 default ANI_CLI_EXIT_THREADS=0 default ANI_CLI_EXIT_THREADS=0 2>&1 | cut -c1-420 | tee "$OUT/e2e_exit_default_ab.txt"
  timeout 300 python tools/exit_probe.py 2>&1 | tee "$OUT/exit_probe.txt"
  timeout 300 python tools/ubench/exit_threads.py 2>&1 | tee "$OUT/exit_threads.txt"
  uname -a | tee "$OUT/uname.txt"
fi
if has e2eiso; then
  echo "== command line end to end, runs 2 s apart (a user's single run): default against ANI_CLI_EXIT_THREADS=0"
  E2E_PAUSE=2 E2E_REPS=${E2E_REPS:-5} timeout 1200 python tools/e2e_probe.py 1000 default ANI_CLI_EXIT_THREADS=0 default ANI_CLI_EXIT_THREADS=0 2>&1 | cut -c1-700 | tee "$OUT/e2e_isolated_ab.txt"
fi
