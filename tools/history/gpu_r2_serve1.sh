#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "serve_skinny" > gpurun_out/r2_sk_kern.log 2>&1
LAP_PARITY_REPORT=1 timeout 1500 python -m pytest tests/test_model_parity_gpu.py -q -x -s -k "full_width_sample or loss_activations or two_layer" > gpurun_out/r2_par2.log 2>&1
timeout 600 python tools/bench_serve.py > gpurun_out/r2_serve1.json 2> gpurun_out/r2_serve1.err
tail -n 8 gpurun_out/r2_sk_kern.log; grep -v "^  " gpurun_out/r2_par2.log | tail -n 12; cat gpurun_out/r2_serve1.json; tail -n 3 gpurun_out/r2_serve1.err
