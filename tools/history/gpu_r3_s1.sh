#!/bin/bash
# round 3, serving stage 1: lean denoise attention + shared-mod skinny prologue + cached time modulations
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "serve or suffix_only or attention_hd256" > gpurun_out/r3_s1_kern.log 2>&1; tail -n 3 gpurun_out/r3_s1_kern.log
timeout 1200 python -m pytest tests/test_model_parity_gpu.py -q -x -k "sample or sampler" > gpurun_out/r3_s1_par.log 2>&1; tail -n 3 gpurun_out/r3_s1_par.log
for v in "LAP_SERVE_ATTN=0" "LAP_SERVE_ATTN=1" "LAP_SERVE_ATTN=1 LAP_SKINNY_NT=1"; do
  echo "== $v"; env $v timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
done
