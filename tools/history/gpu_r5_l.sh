#!/bin/bash
# round 5: scalar-offset DMA stage of the generic GEMM kernel: bitwise tests, k-step floors, serve + train step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -4 | tee gpurun_out/r5l_tests.txt
timeout 500 python tools/probes/prefill_floor.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5l_prefill_floor.txt
timeout 600 python tools/bench_serve.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r5l_serve.txt
for r in 1 2; do ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | tee -a gpurun_out/r5l_step.txt; done
