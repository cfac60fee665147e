exec < /dev/null
# L2 chunk / sub-batch sizes around their defaults, one short bench each (profiles/r04s_tune.txt: the defaults are the flat optimum).
mkdir -p gpurun_out/r04s
run() { env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['stage_ms_per_step_rank0'])" | tee -a gpurun_out/r04s/tune.txt; }
run X=1
run ANI_TEST_L2_CHUNK=1048576
run ANI_TEST_L2_CHUNK=4194304
run ANI_TEST_L2_CHUNK=8388608
run ANI_SUBBATCH_FRAGS=524288
run ANI_SUBBATCH_FRAGS=2097152
run X=1
