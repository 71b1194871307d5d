#!/bin/bash
# round 6, measurement set (the GEMM / attention / serving kernels did not change this round: no PMC pass): bench line (+ isolated shapes), kernel stats +
# in-situ shapes, queue gaps, per-queue step breakdown, step phases and one SigLIP block / LLM layer each way, serving kernel stats + timeline.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LAP_BENCH_SHAPES=1 timeout 1500 python bench.py 2> gpurun_out/r06_bench_shapes_isolated.txt > gpurun_out/r06_bench_line.json
tail -c 400 gpurun_out/r06_bench_line.json
bash tools/prof_bench.sh r06 --no-serve
bash tools/prof_gaps.sh r06 --no-serve
bash tools/prof_overlap.sh r06 lap_gemm_asm_nt_geglu
rocprofv3 --kernel-trace -d gpurun_out/r06_tr -o r -- python tools/step_only.py 4 2 > gpurun_out/r06_tr.log 2>&1
db=$(find gpurun_out/r06_tr -name "*.db" | head -1)
python tools/prof_phases.py $db gpurun_out/r06_phases.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<72, 0>" gpurun_out/r06_siglip_fwd_block.txt
python tools/prof_timeline.py $db "attn_dma_kv_kernel<72>" gpurun_out/r06_siglip_bwd_block.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<256, 0>" gpurun_out/r06_llm_fwd_layer.txt
python tools/prof_timeline.py $db "attn_dma_kv_kernel<256>" gpurun_out/r06_llm_bwd_layer.txt
rm -rf gpurun_out/r06_tr
python tools/step_only.py 6 2 2>/dev/null | tail -1 > gpurun_out/r06_host_ahead.txt
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06s -o r -- python tools/bench_serve.py > gpurun_out/r06_serve.log 2>&1
tail -1 gpurun_out/r06_serve.log | cut -c1-300
db=$(find gpurun_out/prof_r06s -name "*.db" | head -1)
python tools/prof_timeline.py $db im2col_kernel gpurun_out/r06_serve_timeline.txt
python tools/prof_summary.py $db gpurun_out/r06_serve_stats.md 40 > /dev/null
rm -rf gpurun_out/prof_r06s
