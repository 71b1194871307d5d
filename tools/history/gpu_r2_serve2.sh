#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "serve_skinny" > gpurun_out/r2_sk_kern.log 2>&1
tail -n 3 gpurun_out/r2_sk_kern.log
timeout 900 python -m pytest tests/test_model_parity_gpu.py -q -x -k "full_width_sample" > gpurun_out/r2_par3.log 2>&1
tail -n 3 gpurun_out/r2_par3.log
bash tools/prof_serve.sh $1
head -n 16 gpurun_out/prof_serve_$1.md | cut -c1-150
