#!/bin/bash
# kernel trace of the default bench -> overlap report gpurun_out/overlap_<tag>.txt   (usage on the GPU box: tools/prof_overlap.sh <tag> kernel...)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/ov_$tag -o r -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serve > gpurun_out/ov_$tag.bench.log 2>&1
db=$(find gpurun_out/ov_$tag -name "*.db" | head -1)
python tools/prof_overlap.py $db fm_mix_kernel "$@" > gpurun_out/overlap_$tag.txt 2>&1
rm -rf gpurun_out/ov_$tag
tail -1 gpurun_out/ov_$tag.bench.log | cut -c1-160
