#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/bench_prefill_gemm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_i_prefill_gemm.txt
