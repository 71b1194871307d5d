#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> gpurun_out/prof_<tag>/ + markdown summary gpurun_out/prof_<tag>.md
# usage (on the GPU box, via tools/gpu_run.sh): tools/prof_bench.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.bench.log 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/prof_$tag.md 60 > /dev/null
python tools/prof_shapes.py $db > gpurun_out/prof_${tag}_shapes.txt 2>/dev/null
rm -f $db   # large; the summaries are what travels back
tail -1 gpurun_out/prof_$tag.bench.log | cut -c1-160
