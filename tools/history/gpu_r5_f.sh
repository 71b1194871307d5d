#!/bin/bash
# round 5: small-tile k-loop reorder + one-block-per-row reduce_norm: bitwise tests, per-launch floors, serving latency
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_serving_tiles or fused_reduce_norm or gemm_tiles_all or two_phase or tail_split" 2>&1 | tail -5 | tee gpurun_out/r5f_tests.txt
timeout 500 python tools/probes/prefill_floor.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f_prefill_floor.txt
timeout 600 python tools/bench_serve.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r5f_serve.txt
timeout 1500 python -m pytest tests/test_model_parity_gpu.py -x -q -m gpu -k "sample_actions or hipgraph or graph" 2>&1 | tail -5 | tee -a gpurun_out/r5f_tests.txt
