#!/bin/bash
# round-2 first GPU pass: per-layer parity report, the GPU suite, the bench line
mkdir -p gpurun_out
LAP_PARITY_REPORT=1 timeout 1500 python -m pytest tests/test_model_parity_gpu.py -q -x -s -k "full_depth or reference_config or graphed or loss_activations or two_layer" > gpurun_out/r2_par1.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2_all1.log 2>&1
timeout 900 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
tail -5 gpurun_out/r2_par1.log gpurun_out/r2_all1.log; cat gpurun_out/r2_bench1.json
