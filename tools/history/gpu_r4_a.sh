#!/bin/bash
# round 4, call A: per-CU pull-rate / XCD-local barrier probe + the serving baseline on this box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 tools/probes/cu_pull 400 > gpurun_out/r4_cu_pull.txt 2>&1
cat gpurun_out/r4_cu_pull.txt
timeout 300 python tools/bench_serve.py 2>&1 | tail -1 | tee gpurun_out/r4_serve_base.txt
