#!/bin/bash
# round-2 PMC passes (own runs, kernel-trace + pmc only): tile 10 forward, tile 12 wgrad / dgrad on the gate-up shapes; HBM-side bytes
{
bash tools/pmc_gemm.sh fwd 17920 32768 2048 10 a
bash tools/pmc_gemm.sh wgrad 17920 32768 2048 12 a
bash tools/pmc_gemm.sh wgrad 17920 32768 2048 10 b
bash tools/pmc_gemm.sh dgrad 17920 2048 32768 12 a
} > gpurun_out/r02_gemm_pmc.txt 2>&1
bash tools/pmc_traffic.sh wgrad 17920 32768 2048 12 >> gpurun_out/r02_gemm_pmc.txt 2>&1
tail -n 20 gpurun_out/r02_gemm_pmc.txt
