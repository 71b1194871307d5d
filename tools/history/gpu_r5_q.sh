#!/bin/bash
# round 5: optimizer block cap, fine sweep around one block per CU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
  for b in 4096 160 192 224 256 288 320 384; do
    LAP_ADAMW_BLOCKS=$b ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/adamw_blocks=$b/" | tee -a gpurun_out/r5q_adamw_blocks.txt
  done
done
