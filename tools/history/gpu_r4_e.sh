#!/bin/bash
# round 4, call E: the new parity tests first (measurement mode for the A/B), then the whole GPU suite, then the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAP_PARITY_REPORT=1 timeout 2400 python -m pytest tests/test_route_parity_gpu.py -x -q -m gpu -s 2>&1 | tail -25 | tee gpurun_out/r4_e_route.txt
timeout 2400 python -m pytest tests/test_model_parity_gpu.py -x -q -m gpu -k "b16" 2>&1 | tail -15 | tee gpurun_out/r4_e_b16.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_train_loop_gpu.py -x -q -m gpu -k "benchmark_shapes or without_optimizer" 2>&1 | tail -15 | tee gpurun_out/r4_e_kern.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r4_e_all.txt
timeout 1500 python bench.py 2> gpurun_out/r4_e_bench.err | tee gpurun_out/r4_e_bench.json | cut -c1-600
