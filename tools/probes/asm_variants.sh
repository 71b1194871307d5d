#!/bin/bash
# builds schedule variants of the assembly GEMM into tools/probes/variants/<name>.hsaco: name:ENV=VAL,ENV=VAL ...
mkdir -p tools/probes/variants
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $(echo $envs | tr ',' ' ') python tools/gen_gemm_asm.py > /tmp/v_$name.s || exit 1
  /opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/v_$name.s -o /tmp/v_$name.o || exit 1
  /opt/rocm/lib/llvm/bin/ld.lld -shared /tmp/v_$name.o -o tools/probes/variants/$name.hsaco || exit 1
done
ls tools/probes/variants
