#!/bin/bash
# round 5: what the action-expert stream costs (ablation said 43 ms): serial-mode kernel durations + the default mode's all-queue layer timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "loss_branches" 2>&1 | tail -5 | tee gpurun_out/r5b_tests.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LAP_DUAL_STREAM=0 rocprofv3 --kernel-trace -d gpurun_out/ser -o r -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serve > gpurun_out/r5b_serial.bench.log 2>&1
db=$(find gpurun_out/ser -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r5b_serial_stats.md 70 > /dev/null
python tools/prof_timeline.py $db "attn_dma_kv_kernel<256" gpurun_out/r5b_serial_layer_bwd.txt
python tools/prof_timeline.py $db "attn_dma_q_kernel<256, 0>" gpurun_out/r5b_serial_layer_fwd.txt
rm -rf gpurun_out/ser
tail -1 gpurun_out/r5b_serial.bench.log | cut -c1-200
bash tools/prof_layer.sh r5b
tail -1 gpurun_out/ly_r5b.bench.log | cut -c1-200
