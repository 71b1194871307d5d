#!/bin/bash
# A/B of the adaRMS bank's place in the optimizer schedule now that the compute stream no longer waits for it (same box, interleaved)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
  for v in second tower; do
    echo -n "LAP_OPT_ADA_POS=$v  " >> gpurun_out/r6_ab_ada_pos.txt
    LAP_OPT_ADA_POS=$v python tools/step_only.py 8 3 2>/dev/null | tail -1 | cut -d'|' -f1 >> gpurun_out/r6_ab_ada_pos.txt
  done
done
