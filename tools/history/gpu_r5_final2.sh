#!/bin/bash
# round 5, measurement set after the optimizer throttle (no PMC pass: the GEMM kernels did not change): bench line (+ isolated shapes), kernel stats + in-situ shapes, queue gaps, per-queue step breakdown,
# serving kernel stats + timeline, PMC counters of the dominant GEMM (SQ set on nt, L2-to-fabric traffic of nt_geglu), ablation bounds.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAP_BENCH_SHAPES=1 timeout 1500 python bench.py 2> gpurun_out/r05_bench_shapes_isolated.txt > gpurun_out/r05_bench_line.json
tail -c 600 gpurun_out/r05_bench_line.json
bash tools/prof_bench.sh r05 --no-serve
bash tools/prof_gaps.sh r05 --no-serve
bash tools/prof_overlap.sh r05 lap_gemm_asm_nt_geglu
bash tools/gpu_r3_prof_serve.sh r05
for r in 1 2; do for a in none noexpert noexpert_elem noexpert_gemm noopt; do ABL=$a timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | tee -a gpurun_out/r05_ablation_bounds.txt; done; done
