#!/bin/bash
# HBM traffic of one GEMM shape: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmct_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o r -- python tools/bench_one_gemm.py $1 $2 $3 $4 $5 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "gemm_" in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
v = [float(r["Counter_Value"]) for r in rows]
print("$c per launch (raw counter units, KB):", sum(v) / len(v), "launches", len(v))
PY
done
