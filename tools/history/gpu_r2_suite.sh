#!/bin/bash
# the GPU suite + serve bench (+ optional train bench): gpurun_out/r2_suite_<tag>.log
tag=$1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r2_suite_$tag.log 2>&1
tail -n 6 gpurun_out/r2_suite_$tag.log
timeout 600 python tools/bench_serve.py 2>/dev/null | tail -n 1
if [ "$2" == "bench" ]; then timeout 900 python bench.py 2>/dev/null | tail -n 1 > gpurun_out/r2_bench_$tag.json; cat gpurun_out/r2_bench_$tag.json; fi
