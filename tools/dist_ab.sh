cd $GRAFT_REPO_ROOT
line() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
k = d.get('roofline', {}).get('all_kernels_ms_per_step', {})
print('$1', d.get('value'), d.get('ms_per_step'), d.get('stage_ms_per_step_rank0'), {x: k[x] for x in ('ani::k_l2_sim', 'ani::k_l2_codes') if x in k})
"; }
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-verify --no-weak-leg"
for rep in 1 2; do
  ANI_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 $Q 2>/dev/null | line "dist1 default"
  GPU_MAX_HW_QUEUES=8 ANI_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 $Q 2>/dev/null | line "dist1 GPU_MAX_HW_QUEUES=8"
  timeout 300 python bench.py --gpus 1 $Q 2>/dev/null | line "single"
done
