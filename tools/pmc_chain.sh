#!/bin/bash
# HBM-side traffic of the persistent denoise step (serve_chain_kernel): FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots),
# kernel-trace only (gpurun refuses other trace domains next to --pmc).  One launch = 18 expert layers: 623 MB of weights + the cached keys.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcc_$c
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o r -- python tools/probes/chain_clock.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "serve_chain" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
v = [float(r["Counter_Value"]) for r in rows]
print("serve_chain_kernel $c per launch (raw counter units, KB):", round(sum(v) / len(v), 1), "launches", len(v), "min", min(v), "max", max(v))
PY
done
