#!/bin/bash
# round 5: ablation bounds with zero (not garbage) stand-ins, twice interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
for a in none noexpert noexpert_elem noexpert_gemm noopt noattn; do ABL=$a timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=\|Error\|error" | tail -2 | tee -a gpurun_out/r5c_abl.txt; done
LAP_DUAL_STREAM=0 ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed 's/ABL=none/DUAL_STREAM=0/' | tee -a gpurun_out/r5c_abl.txt
done
