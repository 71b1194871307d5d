#!/bin/bash
# usage: asm_try.sh <variant> : bitwise tests + quick bench of a schedule variant of the assembly GEMM
v=$1
export LAP_ASM_HSACO=tools/probes/variants/$v.hsaco
echo "== $v: check"
timeout 300 python tools/bench_asm_gemm.py check 2>&1 | grep -v amdgpu.ids
echo "== $v: pytest"
timeout 600 python -m pytest tests -q -x -m gpu -k "assembly or asm" 2>&1 | tail -5
echo "== $v: quick"
timeout 300 python tools/bench_asm_gemm.py quick nt nn tn 2>&1 | grep -v amdgpu.ids
