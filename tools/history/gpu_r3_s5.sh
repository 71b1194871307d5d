#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "geglu_epilogue or serving_tiles" 2>&1 | tail -n 2
timeout 1500 python -m pytest tests/test_model_parity_gpu.py -q -x -k "sample or sampler" 2>&1 | tail -n 2
timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
bash tools/prof_bench.sh r03a --no-serve; bash tools/prof_gaps.sh r03a --no-serve
