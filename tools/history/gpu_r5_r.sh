#!/bin/bash
# round 5: optimizer throttle, block size x block count around one block per CU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
  for v in "4096 256" "240 256" "256 256" "256 128" "512 128" "256 512" "128 512" "240 192" "480 128"; do set -- $v
    LAP_ADAMW_BLOCKS=$1 LAP_ADAMW_THREADS=$2 ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/adamw blocks=$1 threads=$2/" | tee -a gpurun_out/r5r_adamw.txt
  done
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "adamw" 2>&1 | tail -2
