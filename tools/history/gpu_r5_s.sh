#!/bin/bash
# round 5: knobs around the throttled optimizer (nontemporal accesses, release lookahead, pacing point, norm-pass blocks), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/$*/" | tee -a gpurun_out/r5s_knobs.txt; }
for r in 1 2; do
  run LAP_ADAMW_BLOCKS=240
  run LAP_ADAMW_BLOCKS=240 LAP_ADAMW_NT=0
  run LAP_ADAMW_BLOCKS=240 LAP_OPT_LOOKAHEAD=2
  run LAP_ADAMW_BLOCKS=240 LAP_OPT_LOOKAHEAD=5
  run LAP_ADAMW_BLOCKS=240 LAP_OPT_LOOKAHEAD=30
  run LAP_ADAMW_BLOCKS=240 LAP_OPT_PACE=layer
  run LAP_ADAMW_BLOCKS=240 LAP_SUMSQ_BLOCKS=240
  run LAP_ADAMW_BLOCKS=240 LAP_OPT_LOOKAHEAD=0
done
