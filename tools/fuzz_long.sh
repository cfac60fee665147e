#!/bin/bash
# A long default fuzz of the library against the oracle: many seeds, 600 iterations each (tests/parity_cases.py: fuzz).  usage: tools/fuzz_long.sh <tag> [first seed] [seeds]
TAG=${1:-fuzzlong}; S0=${2:-1000}; NS=${3:-20}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{ for ((i = 0; i < NS; i++)); do timeout 600 python tools/fuzz_gpu.py $((S0 + 37 * i)) 600 2>&1 | tail -2; done; } | tee $OUT/fuzz_long.log
echo "seeds ok: $(grep -c " ok " $OUT/fuzz_long.log)"
