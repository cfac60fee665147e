cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
for v in 0 1; do
  ANI_TEST_POOL_CLASSES=$v ANI_POOL_TRACE=1 timeout 600 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2> gpurun_out/r05k/c4_classes$v.err > gpurun_out/r05k/c4_classes$v.json.log
  python - gpurun_out/r05k/c4_classes$v.json.log <<'PY'
import sys, json
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print('c4 classes', sys.argv[1][-10], d['ms_per_step'], d['stage_ms_per_step_rank0'])
PY
  awk '/timed step 0/{t=1} t && /hipMalloc/{n++; mb+=$4; ms+=$6} END {print "  timed step: hipMalloc calls", n, "GB", mb/1024, "s", ms/1000}' gpurun_out/r05k/c4_classes$v.err
  grep -c "device full" gpurun_out/r05k/c4_classes$v.err
  gzip -f gpurun_out/r05k/c4_classes$v.err
done
