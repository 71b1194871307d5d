#!/bin/bash
# round 5: adaRMS bank's optimizer update in second place (before) vs behind SigLIP's first three layer units (now), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" ABL=none timeout 300 python tools/probes/abl_step.py 8 2>&1 | grep -a "ABL=" | sed "s/ABL=none/$*/" | tee -a gpurun_out/r5w_ada_order.txt; }
for r in 1 2 3; do
  run LAP_OPT_ADA_FIRST=1
  run LAP_OPT_ADA_FIRST=0
done
