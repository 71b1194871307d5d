#!/bin/bash
# round-2 profile set: train-step kernel stats + GEMM shapes, serving kernel stats, queue gaps
bash tools/prof_bench.sh r02a --no-serve
bash tools/prof_serve.sh r02d
bash tools/prof_gaps.sh r02a --no-serve
python tools/bench_serve_split.py 2>/dev/null | tail -n 1 > gpurun_out/r02_serve_split.json
python tools/gemm_ablate.py 2>/dev/null | grep -v amdgpu > gpurun_out/r02_gemm_ablate_prod.txt
cat gpurun_out/r02_serve_split.json
