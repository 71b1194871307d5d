#!/bin/bash
# round 4, call C: (key run x query tile) serve attention — tests, stage clock, chunk latency
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve or attention" 2>&1 | tail -8 | tee gpurun_out/r4_c_tests.txt
PACKED=1 timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_c_clock.txt
for i in 1 2; do
  echo "packed: $(timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c90-330)" | tee -a gpurun_out/r4_c_serve.txt
done
timeout 1500 python -m pytest tests/test_model_parity_gpu.py -x -q -m gpu -k "sample or serve or chain or full_width" 2>&1 | tail -8 | tee gpurun_out/r4_c_tests2.txt
