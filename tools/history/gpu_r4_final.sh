#!/bin/bash
# round 4, final measurement set: whole GPU suite, bench line, kernel stats + queue gaps of the train step, serving kernel stats + timeline,
# chain stage clocks (flat / tensor parallel), PMC traffic of the persistent denoise step, attention counters, fp8 vs bf16.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04_gputests.txt
timeout 1500 python bench.py 2> gpurun_out/r04_bench.err | tee gpurun_out/r04_bench_line.json | cut -c1-300
bash tools/prof_bench.sh r04 --no-serve
bash tools/prof_gaps.sh r04 --no-serve
bash tools/gpu_r3_prof_serve.sh r04
for t in 0 1; do TPAR=$t timeout 300 python tools/probes/chain_clock.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_chain_clock.txt; done
bash tools/pmc_chain.sh 2>&1 | tee gpurun_out/r04_serve_chain_pmc_traffic.txt
( bash tools/pmc_attn.sh -1; bash tools/pmc_attn.sh -1 bwd ) > gpurun_out/r04_attention_hd256_counters.txt 2>&1
tail -4 gpurun_out/r04_attention_hd256_counters.txt | cut -c1-300
( timeout 600 python tools/bench_fp8.py 2>&1 | grep -v amdgpu.ids
  for i in 1 2; do for d in bf16 fp8; do
    echo "bench.py --dtype $d: $(timeout 600 python bench.py --dtype $d --no-serve --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms per step,", d["value"], "samples/s")')"
  done; done ) 2>&1 | tee gpurun_out/r04_fp8_vs_bf16.txt
