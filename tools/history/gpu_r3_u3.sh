#!/bin/bash
# round 3, late: the persistent denoise-layer chain — parity test, then serving latency with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve_chain or serve_skinny or attention_serve" > gpurun_out/u3_test.txt 2>&1
tail -15 gpurun_out/u3_test.txt
for i in 1 2; do
  for c in 0 1; do
    echo "LAP_SERVE_CHAIN=$c: $(LAP_SERVE_CHAIN=$c timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c1-260)" | tee -a gpurun_out/u3_serve.txt
  done
done
