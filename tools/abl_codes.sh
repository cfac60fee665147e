exec < /dev/null
mkdir -p gpurun_out/r04t
for v in "" ablRANK2 ablSTAGE2 ablFLUSH2 ablVALU2 ""; do
  ANI_LIB_VARIANT=$v ANI_L2_OVERLAP=0 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('variant=[$v]', d['ms_per_step'], d['stage_ms_per_step_rank0']['msL2'], {x: k[x] for x in ('ani::k_l2_sim','ani::k_l2_codes')})" | tee -a gpurun_out/r04t/ablation.txt
done
