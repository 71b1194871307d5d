#!/bin/bash
# round 6: kernel trace of the train step alone -> phase breakdown, one SigLIP block forward / backward, one LLM layer forward / backward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/step_only.py 6 2 > gpurun_out/r6_step_only.txt 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r6_tr -o r -- python tools/step_only.py 4 2 > gpurun_out/r6_tr.log 2>&1
db=$(find gpurun_out/r6_tr -name "*.db" | head -1)
python tools/prof_phases.py $db gpurun_out/r6_phases.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<72, 0>" gpurun_out/r6_siglip_fwd_block.txt
python tools/prof_timeline.py $db "attn_dma_kv_kernel<72>" gpurun_out/r6_siglip_bwd_block.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<256, 0>" gpurun_out/r6_llm_fwd_layer.txt
python tools/prof_timeline.py $db "attn_dma_kv_kernel<256>" gpurun_out/r6_llm_bwd_layer.txt
python tools/prof_timeline.py $db "fm_mix_kernel" gpurun_out/r6_full_step.txt
python tools/prof_gaps.py $db > gpurun_out/r6_gaps.txt 2>&1
rm -rf gpurun_out/r6_tr
