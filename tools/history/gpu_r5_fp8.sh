#!/bin/bash
# round 5: fp8 (BASELINE config 5, N = 1) against bf16 on one box, interleaved, with the round's optimizer / gradient changes in both
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
  for d in bf16 fp8; do
    timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-serve --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['value'], 'samples/s', d['ms_per_step'], 'ms per step')" | tee -a gpurun_out/r05_fp8_vs_bf16.txt
  done
done
