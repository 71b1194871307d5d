#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve_chain or attention_serve" 2>&1 | tail -3 | tee gpurun_out/r4_d_tests.txt
PACKED=1 timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_d_clock.txt
echo "packed: $(timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | cut -c90-330)" | tee -a gpurun_out/r4_d_serve.txt
