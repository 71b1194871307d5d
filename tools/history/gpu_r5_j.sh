#!/bin/bash
# round 5: action-expert stream — wide per-sample backward kernels + expert weight gradients off the suffix stream: tests, then interleaved A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rmsnorm or gated or norm" 2>&1 | tail -4 | tee gpurun_out/r5j_tests.txt
LAP_WGRAD_STREAM=sbe timeout 2400 python -m pytest tests/test_model_parity_gpu.py -x -q -m gpu -k "loss_activations_and_grads or full_width_two_layer or stop or train_step" 2>&1 | tail -4 | tee -a gpurun_out/r5j_tests.txt
for r in 1 2 3; do
  for v in "sb 0" "sb 1" "sbe 1" "sbe 0"; do set -- $v
    LAP_WGRAD_STREAM=$1 LAP_NORM_BWD_WIDE=$2 ABL=none timeout 300 python tools/probes/abl_step.py 6 2>&1 | grep -a "ABL=" | sed "s/ABL=none/wgrad=$1 wide=$2/" | tee -a gpurun_out/r5j_ab.txt
  done
done
