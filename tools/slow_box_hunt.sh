#!/bin/bash
# Is this a box where fresh device memory is slow (20 - 40 us per MB)?  If so, measure what only such a box can tell: whether hipMalloc runs in
# parallel on several threads, and the command line end to end with and without the reservation.  On a fast box: leave after the probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/hunt; export TMPDIR=/tmp
T=gpurun_out/hunt/alloc_$(date +%s).txt
timeout 300 tools/ubench/alloc > $T 2>&1
us=$(awk '/ 4096 MB rep 1/ {gsub(/[()]/,""); for (i=1;i<=NF;i++) if ($i=="us/MB") {print $(i-1); exit}}' $T)
echo "hipMalloc of 4 GiB: $us us/MB"
if awk "BEGIN{exit !($us > 5)}"; then
  echo "== SLOW-ALLOCATION BOX"; cat $T
  E2E_REPS=4 timeout 900 python tools/e2e_probe.py 1000 default ANI_CLI_PREWARM=0 2>&1 | tee gpurun_out/hunt/e2e_slow_box.txt | cut -c1-1200
else
  rm -f $T; echo "fast box"
fi
