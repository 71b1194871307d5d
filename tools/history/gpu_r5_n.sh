#!/bin/bash
# round 5: bf16 weight-gradient buffers: kernel tests, model-level gradient parity, train-step tests, then interleaved A/B of the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_assembly_kernels_match or adamw" 2>&1 | tail -6 | tee gpurun_out/r5n_tests.txt
timeout 3000 python -m pytest tests/test_model_parity_gpu.py tests/test_train_loop_gpu.py tests/test_fsdp_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee -a gpurun_out/r5n_tests.txt
for r in 1 2 3; do for g in 0 1; do
  LAP_GRAD_BF16=$g ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/grad_bf16=$g/" | tee -a gpurun_out/r5n_ab.txt
done; done
