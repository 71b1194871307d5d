#!/bin/bash
# round 5: throttling the optimizer pass (fewer blocks per launch = less HBM demand per unit time), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
  for b in 4096 1024 512 256 128 64; do
    LAP_ADAMW_BLOCKS=$b ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/adamw_blocks=$b/" | tee -a gpurun_out/r5p_adamw_blocks.txt
  done
done
for l in 4 13 20; do for b in 256 128; do
  LAP_OPT_LOOKAHEAD=$l LAP_ADAMW_BLOCKS=$b ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/lookahead=$l adamw_blocks=$b/" | tee -a gpurun_out/r5p_adamw_blocks.txt
done; done
