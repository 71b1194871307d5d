#!/bin/bash
# round 4, call F: tensor-parallel chain — kernel tests, per-layer time, chunk latency TP vs flat; route tests again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve_chain" 2>&1 | tail -12 | tee gpurun_out/r4_f_tests.txt
for t in 0 1; do TPAR=$t timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | head -1 | tee -a gpurun_out/r4_f_clock.txt; done
for i in 1 2; do for t in 0 1; do
  echo "LAP_SERVE_TP=$t: $(LAP_SERVE_TP=$t timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c90-330)" | tee -a gpurun_out/r4_f_serve.txt
done; done
LAP_PARITY_REPORT=1 timeout 2400 python -m pytest tests/test_route_parity_gpu.py -x -q -m gpu -s 2>&1 | tail -25 | tee gpurun_out/r4_f_route.txt
timeout 2400 python -m pytest tests/test_model_parity_gpu.py -x -q -m gpu -k "b16" 2>&1 | tail -5 | tee gpurun_out/r4_f_b16.txt
