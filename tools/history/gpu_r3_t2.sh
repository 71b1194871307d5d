#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_model_parity_gpu.py -q -x -k "not full_depth and not full_size_train" > gpurun_out/r3_t2_par.log 2>&1; tail -n 5 gpurun_out/r3_t2_par.log | cut -c1-400
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_train_loop_gpu.py tests/test_fsdp_gpu.py -q -x > gpurun_out/r3_t2_kern.log 2>&1; tail -n 5 gpurun_out/r3_t2_kern.log | cut -c1-400
bash tools/ab3.sh 2 "LAP_LM_ALL_ROWS=1" "LAP_X=1"
