#!/bin/bash
# same-box A/B of the train step: library with the round-3 attention backward (liblap_hip_r3attn.so, built from commit 04bd495's
# attention.hip + today's other objects) against today's library; interleaved, 3 runs each
cd $GRAFT_REPO_ROOT/lap_amd
set -e; cp liblap_hip.so /tmp/new.so; cp liblap_hip_r3attn.so /tmp/old.so; set +e
cd ..
for i in 1 2 3; do for v in old new; do
  cp /tmp/$v.so lap_amd/liblap_hip.so
  echo "$v: $(timeout 600 python bench.py --no-serve --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms per step,", d["value"], "samples/s")')"
done; done | tee gpurun_out/r4_attn_step_ab.txt
cp /tmp/new.so lap_amd/liblap_hip.so
