#!/bin/bash
# first thing on a fresh box: tools/ubench/alloc_par (parallel / piece-size behaviour of slow fresh memory), then the plain probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/hunt
timeout 600 tools/ubench/alloc_par 2>&1 | tee gpurun_out/hunt/alloc_par_$(date +%s).txt
