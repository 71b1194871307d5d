#!/bin/bash
# kernel trace of the default bench -> all-queue timeline of ONE backward layer (between two dK/dV attention launches): gpurun_out/layer_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/ly_$tag -o r -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serve > gpurun_out/ly_$tag.bench.log 2>&1
db=$(find gpurun_out/ly_$tag -name "*.db" | head -1)
python tools/prof_timeline.py $db "attn_dma_kv_kernel<256>" gpurun_out/layer_$tag.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<256, 0>" gpurun_out/layer_fwd_$tag.txt
rm -rf gpurun_out/ly_$tag
