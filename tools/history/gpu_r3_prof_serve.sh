#!/bin/bash
# serve kernel stats + timeline of one replay -> gpurun_out/r3_serve_<tag>_{stats.md,timeline.txt}
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3s_$tag -o r -- python tools/bench_serve.py > gpurun_out/r3_serve_$tag.log 2>&1
tail -1 gpurun_out/r3_serve_$tag.log | cut -c1-300
db=$(find gpurun_out/prof_r3s_$tag -name "*.db" | head -1)
python tools/prof_timeline.py $db im2col_kernel gpurun_out/r3_serve_${tag}_timeline.txt
python tools/prof_summary.py $db gpurun_out/r3_serve_${tag}_stats.md 40 > /dev/null
rm -rf gpurun_out/prof_r3s_$tag
head -20 gpurun_out/r3_serve_${tag}_stats.md | cut -c1-160
