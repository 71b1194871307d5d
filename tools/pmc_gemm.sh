#!/bin/bash
# PMC counters for one GEMM shape (own run, kernel-trace only as gpurun requires)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$1_$6
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -o r -- python tools/bench_one_gemm.py $1 $2 $3 $4 $5 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    if "gemm_" not in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]: continue
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
for k, d in agg.items():
    n = cnt[(k, "SQ_WAVE_CYCLES")]
    print("$1 $2x$3x$4 tile$5:", {c: round(v / n / 1e6, 2) for c, v in d.items()}, "launches", n)
    wc = d["SQ_WAVE_CYCLES"]
    print("   frac of wave-cycles: wait_any %.2f active %.2f wait_lds %.3f ; lds conflict/active %.3f ; mfma_busy/busy_cycles %.3f" % (d["SQ_WAIT_ANY"]/wc, d["SQ_ACTIVE_INST_ANY"]/wc, d["SQ_WAIT_INST_LDS"]/wc, d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1), d["SQ_VALU_MFMA_BUSY_CYCLES"]/max(d["SQ_BUSY_CYCLES"],1)))
PY
