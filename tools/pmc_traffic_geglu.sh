#!/bin/bash
# L2-to-fabric traffic of the step's dominant kernel, lap_gemm_asm_nt_geglu at 17920 x 32768 x 2048: FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots), kernel-trace only (gpurun's rule).  Output format = tools/pmc_traffic.sh (bench.py parses it).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "# traffic kernel: lap_gemm_asm_nt_geglu 17920 x 32768 x 2048 (gate|up + GeGLU, writes gu [17920, 32768] and act [17920, 16384])"
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmctg_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o r -- python tools/bench_gfwd_once.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "nt_geglu" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
v = [float(r["Counter_Value"]) for r in rows]
print("$c per launch (raw counter units, KB):", sum(v) / len(v), "launches", len(v))
PY
done
