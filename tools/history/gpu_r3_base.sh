#!/bin/bash
# round-3 baseline: serve split, serve kernel timeline of one replay, default bench line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_serve_split.py > gpurun_out/r3_base_split.log 2>&1; tail -1 gpurun_out/r3_base_split.log
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3base -o r -- python tools/bench_serve.py > gpurun_out/r3_base_serve.log 2>&1
tail -1 gpurun_out/r3_base_serve.log | cut -c1-300
db=$(find gpurun_out/prof_r3base -name "*.db" | head -1)
python tools/prof_timeline.py $db im2col_kernel gpurun_out/r3_base_serve_timeline.txt
python tools/prof_summary.py $db gpurun_out/r3_base_serve_stats.md 40 > /dev/null
rm -rf gpurun_out/prof_r3base
timeout 900 python bench.py > gpurun_out/r3_base_bench.log 2>&1; tail -1 gpurun_out/r3_base_bench.log | cut -c1-600
