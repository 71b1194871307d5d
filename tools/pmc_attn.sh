#!/bin/bash
# PMC counters for the attention kernels (own run, kernel-trace only as gpurun requires): tools/pmc_attn.sh <variant> [bwd]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
out=gpurun_out/pmc_attn_$1_$RANDOM
rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o r -- python tools/attn_one.py $1 $2 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    if "attn" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    n = cnt[(k, "SQ_WAVE_CYCLES")]
    print(k, {c: round(v / n / 1e6, 3) for c, v in d.items()}, "launches", n)
PY
done
