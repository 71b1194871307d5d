#!/bin/bash
# times every variant on a few shapes: python tools/bench_asm_gemm.py quick
for f in tools/probes/variants/*.hsaco; do
  n=$(basename $f .hsaco)
  r=$(LAP_ASM_HSACO=$f timeout 200 python tools/bench_asm_gemm.py quick $LAYOUTS 2>/dev/null | awk '{printf "%s/%s ", $6, ($5=="True")?"ok":"BAD"}')
  echo "$n: $r"
done
