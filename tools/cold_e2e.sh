#!/bin/bash
# the command line as the FIRST device user of a fresh box (on a box whose driver wipes VRAM all of its memory is still dirty then): one cold run, then two more
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/cold
E2E_REPS=3 timeout 900 python tools/e2e_probe.py 1000 default 2>&1 | tee gpurun_out/cold/e2e_$(date +%s).txt | cut -c1-1500
