#!/bin/bash
# round-3 profile set: train-step kernel stats + in-situ GEMM shapes, queue gaps, serving kernel stats + timeline, PMC passes, bench line
mkdir -p gpurun_out
bash tools/prof_bench.sh r03 --no-serve
bash tools/prof_gaps.sh r03 --no-serve
bash tools/gpu_r3_prof_serve.sh r03
( bash tools/pmc_gemm.sh fwd 17920 32768 2048 14 r03; bash tools/pmc_gemm.sh wgrad 17920 32768 2048 14 r03; bash tools/pmc_traffic.sh fwd 17920 32768 2048 14 ) > gpurun_out/r03_gemm_pmc_counters.txt 2>&1
tail -4 gpurun_out/r03_gemm_pmc_counters.txt
LAP_BENCH_SHAPES=1 timeout 1500 python bench.py 2> gpurun_out/r03_bench_shapes_isolated.txt > gpurun_out/r03_bench_line.json
tail -c 1500 gpurun_out/r03_bench_line.json
