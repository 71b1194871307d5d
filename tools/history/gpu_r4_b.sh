#!/bin/bash
# round 4, call B: the packed chain — bitwise tests, stage clock packed vs row-major, chunk latency packed vs row-major
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve_chain or serve_pack" 2>&1 | tail -5 | tee gpurun_out/r4_b_tests.txt
for p in 0 1; do PACKED=$p timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_b_clock.txt; done
for i in 1 2; do for p in 0 1; do
  echo "LAP_SERVE_PACKED=$p: $(LAP_SERVE_PACKED=$p timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c90-330)" | tee -a gpurun_out/r4_b_serve.txt
done; done
