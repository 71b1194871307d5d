#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_model_parity_gpu.py -q -x -k "loss_rows or train_step or vqa" > gpurun_out/r3_t3_par.log 2>&1; tail -n 3 gpurun_out/r3_t3_par.log | cut -c1-400
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_train_loop_gpu.py tests/test_fsdp_gpu.py tests/test_policy_gpu.py tests/test_full_size_gpu.py -q > gpurun_out/r3_t3_kern.log 2>&1; tail -n 5 gpurun_out/r3_t3_kern.log | cut -c1-400
bash tools/ab3.sh 2 "LAP_LM_ALL_ROWS=1 LAP_LM_NO_LO=1" "LAP_LM_NO_LO=1" "LAP_X=1"
