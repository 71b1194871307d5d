#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_parity_gpu.py tests/test_policy_gpu.py -q -x -k "sample or sampler or policy" 2>&1 | tail -n 2
for v in "LAP_SERVE_OVERLAP=0" "LAP_SERVE_OVERLAP=1" "LAP_SERVE_OVERLAP=0" "LAP_SERVE_OVERLAP=1"; do
  echo "== $v"; env $v timeout 600 python tools/bench_serve_split.py 2>&1 | tail -1
done
LAP_SERVE_OVERLAP=1 timeout 600 python tools/bench_serve.py 2>&1 | tail -1 | cut -c1-300
