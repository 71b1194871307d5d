#!/bin/bash
# round 5, first contact: new loss-branch / policy tests, baseline bench line with the new JSON fields, ablation bounds, PMC of the dominant kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "loss_branches or policy_from_checkpoint" 2>&1 | tail -15 | tee gpurun_out/r5a_tests.txt
timeout 1500 python bench.py 2> gpurun_out/r5a_bench.err | tee gpurun_out/r5a_bench_line.json | cut -c1-400
for a in none noexpert noopt noattn noexpert,noopt; do ABL=$a timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep ABL= | tee -a gpurun_out/r5a_abl.txt; done
( bash tools/pmc_gemm.sh fwd 17920 32768 2048 14 r05; bash tools/pmc_traffic_geglu.sh ) > gpurun_out/r05_gemm_pmc_counters.txt 2>&1
tail -5 gpurun_out/r05_gemm_pmc_counters.txt
