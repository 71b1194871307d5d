#!/bin/bash
# round 3, serving stage 2: serving tiles, fused prefill consumers
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "serving_tiles or gelu_after or fused_reduce_norm or serve" > gpurun_out/r3_s2_kern.log 2>&1; tail -n 3 gpurun_out/r3_s2_kern.log
timeout 1500 python -m pytest tests/test_model_parity_gpu.py -q -x -k "sample or sampler or full_depth" > gpurun_out/r3_s2_par.log 2>&1; tail -n 3 gpurun_out/r3_s2_par.log
for v in "LAP_SERVE_FUSIONS=0 LAP_GEMM_NO_SERVING_TILES=1" "LAP_SERVE_FUSIONS=0" "LAP_SERVE_FUSIONS=1" "LAP_PREFILL_KS=2,2,8" "LAP_PREFILL_KS=5,4,8" "LAP_PREFILL_KS=4,4,6"; do
  echo "== $v"; env $v timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
done
