#!/bin/bash
# round-2 final profile set (current library: assembly GEMM kernels with tile counters, two model streams, paced optimizer): train-step kernel stats + GEMM shapes,
# queue gaps, serving kernel stats, PMC passes of the gate|up forward (assembly vs HIP tile 10) and its weight gradient
bash tools/prof_bench.sh r02g --no-serve
bash tools/prof_gaps.sh r02g --no-serve
bash tools/prof_serve.sh r02g
{
bash tools/pmc_gemm.sh fwd 17920 32768 2048 14 a
bash tools/pmc_gemm.sh fwd 17920 32768 2048 10 b
bash tools/pmc_gemm.sh wgrad 17920 32768 2048 14 c
bash tools/pmc_gemm.sh wgrad 17920 32768 2048 12 d
} > gpurun_out/r02g_gemm_pmc.txt 2>&1
bash tools/pmc_traffic.sh fwd 17920 32768 2048 14 >> gpurun_out/r02g_gemm_pmc.txt 2>&1
tail -n 14 gpurun_out/r02g_gemm_pmc.txt
