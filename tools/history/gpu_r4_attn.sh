#!/bin/bash
# attention A/B (variant 1 = production, 2 = experimental instantiations) + per-kernel durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/probes/attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_attn_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/attn_ab_prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/attn_ab_prof -o r --output-format csv -- python tools/probes/attn_ab.py > /dev/null 2>&1
f=$(find gpurun_out/attn_ab_prof -name "*kernel_stats.csv" | head -1)
python - <<PY | tee -a gpurun_out/r4_attn_ab.txt
import csv
for r in csv.DictReader(open("$f")):
    if "attn" in r["Name"]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
rm -rf gpurun_out/attn_ab_prof
