#!/bin/bash
# round-6 probe: the 8-wave forward kernel (tools/gen_gemm_asm.py Kernel8) against the production 4-wave kernel: bitwise check, then timing with and without epilogue
cd $GRAFT_REPO_ROOT
V=tools/probes/variants
echo "== check (8-wave vs HIP tile, bitwise)"; LAP_ASM_HSACO=$V/nt8.hsaco LAP_ASM_NT_THREADS=512 timeout 300 python tools/bench_asm_gemm.py check nt 2>&1 | grep -v amdgpu
echo "== quick production 4-wave"; timeout 300 python tools/bench_asm_gemm.py quick nt 2>&1 | grep -v amdgpu
echo "== quick 8-wave (direct epilogue)"; LAP_ASM_HSACO=$V/nt8.hsaco LAP_ASM_NT_THREADS=512 timeout 300 python tools/bench_asm_gemm.py quick nt 2>&1 | grep -v amdgpu
echo "== quick 4-wave, no epilogue"; LAP_ASM_HSACO=$V/base_noepi.hsaco timeout 300 python tools/bench_asm_gemm.py quick nt 2>&1 | grep -v amdgpu
echo "== quick 8-wave, no epilogue"; LAP_ASM_HSACO=$V/nt8_noepi.hsaco LAP_ASM_NT_THREADS=512 timeout 300 python tools/bench_asm_gemm.py quick nt 2>&1 | grep -v amdgpu
