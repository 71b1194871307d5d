#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_model_parity_gpu.py -q -x -s -k "full_depth or full_size_train" > gpurun_out/r3_t1_par.log 2>&1; grep -v "^  " gpurun_out/r3_t1_par.log | tail -n 12 | cut -c1-600
