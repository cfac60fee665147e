#!/bin/bash
# Time-boxed GPU fuzz of the library against the oracle under several engine configurations (tests/parity_cases.py: fuzz), then the
# parity suite with poisoned pool memory.  usage: [FUZZ_LITE=1] tools/fuzz_round.sh <tag>   (lite: the first four legs + one poisoned suite, ~5 min)
TAG=${1:-fuzz}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
echo "== default"; timeout 240 python tools/fuzz_gpu.py 2>&1 | tail -4
echo "== streamed reference sets (1 resident chunk of <= 5000 minimizers)"; ANI_MAX_INDEX_MINIMIZERS=5000 ANI_MAX_RESIDENT_CHUNKS=1 timeout 300 python tools/fuzz_gpu.py 41 60 2>&1 | tail -2
echo "== tiny big-L1 groups, noise filter always on, tiny sub-batches"; ANI_TEST_L1_BIG_GROUP_HITS=20000 ANI_TEST_L1_BIG_GROUP_FRAGS=2 ANI_TEST_L1_FILTER_MIN=0 ANI_SUBBATCH_FRAGS=5 ANI_TEST_L2_CHUNK=9 timeout 300 python tools/fuzz_gpu.py 43 60 2>&1 | tail -2
echo "== host pool for every host loop (ANI_TEST_HOST_PAR_MIN_WORK=0, 5 threads)"; ANI_TEST_HOST_PAR_MIN_WORK=0 ANI_HOST_THREADS=5 timeout 300 python tools/fuzz_gpu.py 67 60 2>&1 | tail -2
[ -n "${FUZZ_LITE:-}" ] && { echo "== parity suite with poisoned pool memory (0xff)"; ANI_TEST_POOL_POISON=255 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3; echo "== lite round done"; exit 0; }
echo "== L2 overlap off"; ANI_TEST_L2_OVERLAP=0 timeout 300 python tools/fuzz_gpu.py 47 60 2>&1 | tail -2
echo "== small L1 classes off the table: hit cap 700 (class M from 701 hits on is skipped: everything beyond goes the batched path), wave kernel off"; ANI_TEST_L1_LDS_MAX=700 ANI_TEST_L1_TINY=0 timeout 300 python tools/fuzz_gpu.py 71 60 2>&1 | tail -2
echo "== parity suite with poisoned pool memory (0xff)"; ANI_TEST_POOL_POISON=255 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== parity suite with poisoned pool memory (0x5a)"; ANI_TEST_POOL_POISON=90 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
} | tee $OUT/fuzz.log
