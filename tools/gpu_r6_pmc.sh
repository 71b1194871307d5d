#!/bin/bash
# round 6: PMC passes (own runs, kernel-trace only) — SQ set of the plain forward kernel, traffic of the gate|up + GeGLU kernel (bench.py's `traffic`), traffic of the
# fused GeGLU-backward data gradient against the plain one — and the ablation bounds of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( bash tools/pmc_gemm.sh fwd 17920 32768 2048 14 r06; bash tools/pmc_traffic_geglu.sh ) > gpurun_out/r06_gemm_pmc_counters.txt 2>&1
bash tools/pmc_traffic_gbwd.sh > gpurun_out/r06_geglu_bwd_pmc_traffic.txt 2>&1
python tools/bench_gbwd.py 2>/dev/null | tail -1 >> gpurun_out/r06_geglu_bwd_pmc_traffic.txt
for r in 1 2; do for a in none noexpert noopt; do ABL=$a timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | tee -a gpurun_out/r06_ablation_bounds.txt; done; done
