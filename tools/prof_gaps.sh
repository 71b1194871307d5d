#!/bin/bash
# kernel trace of the default bench -> idle-gap report gpurun_out/gaps_<tag>.txt   (usage on the GPU box: tools/prof_gaps.sh <tag>)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/gaps_$tag -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/gaps_$tag.bench.log 2>&1
db=$(find gpurun_out/gaps_$tag -name "*.db" | head -1)
python tools/prof_gaps.py $db 20 fm_mix_kernel 2 ${GAPS_TARGET:-lap_gemm_asm_tn} > gpurun_out/gaps_$tag.txt 2>&1
rm -rf gpurun_out/gaps_$tag
tail -1 gpurun_out/gaps_$tag.bench.log | cut -c1-160
