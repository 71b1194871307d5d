#!/bin/bash
# round 4, closing measurement set after the attention-backward change: bench line, kernel stats, per-queue step breakdown, queue gaps, attention A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py 2> gpurun_out/r04b_bench.err | tee gpurun_out/r04b_bench_line.json | cut -c1-200
bash tools/prof_bench.sh r04b --no-serve
bash tools/prof_overlap.sh r04b lap_gemm_asm_nt_geglu
bash tools/prof_gaps.sh r04b --no-serve
bash tools/gpu_r4_attn.sh | tail -8
