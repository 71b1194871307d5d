export LAP_ASM_HSACO=tools/probes/variants/ring.hsaco
echo "== ring: check tn"
timeout 120 python tools/bench_asm_gemm.py check tn 2>&1 | grep -v amdgpu.ids
echo "== ring: pytest"
timeout 300 python -m pytest tests -q -x -m gpu -k "assembly" 2>&1 | tail -3
echo "== ring: quick"
timeout 120 python tools/bench_asm_gemm.py quick tn 2>&1 | grep -v amdgpu.ids
export LAP_ASM_HSACO=tools/probes/variants/noring.hsaco
echo "== noring: quick"
timeout 120 python tools/bench_asm_gemm.py quick tn 2>&1 | grep -v amdgpu.ids
