#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "serve_infos or serving_tiles" > gpurun_out/r3_s4_kern.log 2>&1; tail -n 3 gpurun_out/r3_s4_kern.log
timeout 1500 python -m pytest tests/test_model_parity_gpu.py tests/test_policy_gpu.py -q -x -k "sample or sampler or policy" > gpurun_out/r3_s4_par.log 2>&1; tail -n 3 gpurun_out/r3_s4_par.log
timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
