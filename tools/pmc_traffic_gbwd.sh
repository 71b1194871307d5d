#!/bin/bash
# L2-to-fabric traffic and duration of the fused GeGLU-backward data gradient (lap_gemm_asm_nn_geglu_bwd, 17920 x 16384 x 2048) against the plain data
# gradient of the same shape (lap_gemm_asm_nn): FETCH_SIZE and WRITE_SIZE in separate passes, kernel-trace only (gpurun's rule).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "# lap_gemm_asm_nn_geglu_bwd vs lap_gemm_asm_nn at 17920 x 16384 x 2048: what the fused epilogue moves (gate|up read 1.17 GB, d(gate|up) written 1.17 GB; the plain kernel writes d(act) 0.59 GB)"
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcgb_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o r -- python tools/bench_gbwd_once.py > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for kern in ("lap_gemm_asm_nn_geglu_bwd", "lap_gemm_asm_nn"):
    sel = [r for r in rows if r["Kernel_Name"].strip() == kern and r["Counter_Name"] == "$c"]
    v = [float(r["Counter_Value"]) for r in sel]
    d = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in sel] if sel and "End_Timestamp" in sel[0] else []
    print(f"{kern:28s} $c per launch (raw counter units, KB): {sum(v) / max(len(v), 1):.0f}  launches {len(v)}" + (f"  avg {sum(d) / len(d):.1f} us" if d else ""))
PY
  rm -rf $out
done
