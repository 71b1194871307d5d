export LAP_ASM_HSACO=tools/probes/variants/peel.hsaco
echo "== peel: check"
timeout 200 python tools/bench_asm_gemm.py check 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6}'
echo "== peel: pytest"
timeout 400 python -m pytest tests -q -x -m gpu -k "assembly" 2>&1 | tail -3
echo "== peel: quick"
timeout 200 python tools/bench_asm_gemm.py quick nt nn tn 2>&1 | grep -v amdgpu.ids
export LAP_ASM_HSACO=tools/probes/variants/nopeel.hsaco
echo "== nopeel: quick"
timeout 200 python tools/bench_asm_gemm.py quick nt nn tn 2>&1 | grep -v amdgpu.ids
