#!/bin/bash
# round 5: whole GPU suite + bench line (state after the prefill / gated-residual-backward changes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5k_gputests.txt
timeout 1500 python bench.py 2> gpurun_out/r5k_bench.err | tee gpurun_out/r5k_bench_line.json | cut -c1-300
