#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args]   (PMC pass: counters only, kernel-trace, no other tracing)
TAG=$1; CTRS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-e2e --no-verify "$@" > $OUT/pmc.log 2>&1
cd $REPO
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" | tee $OUT/pmc_summary.txt
