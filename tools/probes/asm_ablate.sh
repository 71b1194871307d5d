#!/bin/bash
# timing ablations of the assembly GEMM loop (results are WRONG by construction): which part of the k-tile costs what
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v" 
  LAP_ASM_HSACO=tools/probes/variants/$v.hsaco timeout ${ABL_TIMEOUT:-120} python tools/bench_asm_gemm.py quick ${ABL_LAYOUTS:-nt} 2>&1 | grep -v "^$"
done
