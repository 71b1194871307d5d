#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for c in 0 1; do
    echo "LAP_SERVE_CHAIN=$c: $(LAP_SERVE_CHAIN=$c timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c90-330)" | tee -a gpurun_out/u8_serve.txt
  done
done
bash tools/gpu_r3_prof_serve.sh chain > gpurun_out/u8_prof.log 2>&1
tail -5 gpurun_out/u8_prof.log
