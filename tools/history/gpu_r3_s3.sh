#!/bin/bash
mkdir -p gpurun_out
for v in "LAP_PREFILL_KS=4,1,8,5" "LAP_PREFILL_KS=4,4,8,5" "LAP_PREFILL_KS=4,1,16,15" "LAP_PREFILL_KS=4,1,8,19" "LAP_PREFILL_KS=4,1,8,15" "LAP_PREFILL_KS=4,1,12,15" "LAP_PREFILL_KS=3,1,8,5"; do
  echo "== $v"; env $v timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
done
