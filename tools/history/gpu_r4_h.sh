#!/bin/bash
# round 4, call H: train-step A/B of the stream layout (interleaved on one box): default | no second stream | no third stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "$1: $(env $2 timeout 600 python bench.py --no-serve --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["in_situ_event_timed"]["gemm_ms_per_step"])')" | tee -a gpurun_out/r4_h_ab.txt; }
for i in 1 2; do
  run default "X=1"
  run nodual "LAP_DUAL_STREAM=0"
  run nowg "LAP_WGRAD_STREAM=0"
  run nodual_nowg "LAP_DUAL_STREAM=0 LAP_WGRAD_STREAM=0"
done
