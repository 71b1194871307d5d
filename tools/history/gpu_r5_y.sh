#!/bin/bash
# round 5: what the third stream (SigLIP weight gradients + bias column sums) costs the step: ablation, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do for a in none nosigwgrad nosigwgrad,nosigbgrad; do ABL=$a timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | tee -a gpurun_out/r5y_wgrad_ablation.txt; done; done
