#!/bin/bash
# round 5: where does a small-M GEMM's k-step go?  SQ wait breakdown + memory-side counters of 512 x 1152 x 4608 on the 64 x 64 tile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_gemm.sh fwd 512 1152 4608 17 r5g 2>&1 | tee gpurun_out/r5g_pmc.txt
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|SQ_[A-Z_0-9a-z]*\)\b" | sort -u > gpurun_out/r5g_counters.txt
wc -l gpurun_out/r5g_counters.txt
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"; do
  out=gpurun_out/pmcg_$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o r -- python tools/bench_one_gemm.py fwd 512 1152 4608 17 > /dev/null 2>&1
  python - <<PY 2>&1 | tee -a gpurun_out/r5g_pmc.txt
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
if not f: print("no output for: $set")
else:
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "gemm_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print({k: round(v / cnt[k] / 1e3, 1) for k, v in agg.items()}, "(thousands per launch)")
PY
done
