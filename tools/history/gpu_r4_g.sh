#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "serve_chain" 2>&1 | tail -3
TPAR=1 timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_g_clock.txt
