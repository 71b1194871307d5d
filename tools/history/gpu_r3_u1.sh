#!/bin/bash
# round 3, late: host-order switch A/B, gap reports around the SigLIP off-path excursions, marker cost probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python tools/probes/marker_cost.py > gpurun_out/u1_marker.txt 2>&1
timeout 600 tools/ab3.sh 2 LAP_BWD_PFX_FIRST=0 LAP_BWD_PFX_FIRST=1 > gpurun_out/u1_ab.txt 2>&1
GAPS_TARGET='gemm_pq_kernel<true, false, false>' timeout 300 tools/prof_gaps.sh u1_siglip --no-serve
LAP_BWD_PFX_FIRST=1 timeout 300 tools/prof_gaps.sh u1_pfx --no-serve
cat gpurun_out/u1_marker.txt gpurun_out/u1_ab.txt
