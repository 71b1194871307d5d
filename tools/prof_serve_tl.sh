#!/bin/bash
# rocprofv3 kernel trace of tools/bench_serve.py -> gpurun_out/prof_serve_<tag>.md + the timeline of one replayed chunk
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_serve_$tag -o r -- python tools/bench_serve.py > gpurun_out/prof_serve_$tag.log 2>&1
db=$(find gpurun_out/prof_serve_$tag -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/prof_serve_$tag.md 40 > /dev/null
python tools/prof_timeline.py $db im2col_kernel gpurun_out/prof_serve_${tag}_timeline.txt
rm -f $db
tail -1 gpurun_out/prof_serve_$tag.log | cut -c1-200
