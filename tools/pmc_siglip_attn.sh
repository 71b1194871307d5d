#!/bin/bash
# L2-to-fabric traffic of the SigLIP attention kernels (head size 72, fused qkv buffer: a head's row is 144 contiguous bytes at a 6912-byte
# row stride): FETCH_SIZE and WRITE_SIZE in separate passes, kernel-trace only (gpurun's rule), caches flushed between launches.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/probes/siglip_attn_once.py 6 2>&1 | grep -v amdgpu.ids
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcsa_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o r -- python tools/probes/siglip_attn_once.py 4 > /dev/null 2>&1
  python - <<PY
import collections, csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "$c" and "attn" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"$c per launch (raw counter units, KB) {k:60s} {sum(v) / len(v):12.1f}  launches {len(v)}")
PY
  rm -rf $out
done
