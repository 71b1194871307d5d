#!/bin/bash
# round 6: head-resident SigLIP attention: tests, kernel A/B (rocprofv3 averages), step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || (timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -15) > gpurun_out/r6_attn_tests.log 2>&1
cat > /tmp/attn72.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from lap_amd import hip
B, T, NH, HD = 64, 256, 16, 72
W = NH * HD
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * T, 3 * W, device="cuda", generator=g) * 0.7).bfloat16()
do = torch.randn(B * T, W, device="cuda", generator=g).bfloat16()
q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
for variant in (3, -1, 3, -1):
    hip.attention_set_variant(variant)
    for _ in range(20):
        (o, _), lse = hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD, scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0))
        dqkv = torch.empty_like(qkv)
        hip.attention_bwd([q], [k], [v], [o], [do], lse, [T], [T], B, NH, NH, HD, scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0),
                          dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]])
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/r6_attn72 -o r -- python /tmp/attn72.py > gpurun_out/r6_attn72.log 2>&1
db=$(find gpurun_out/r6_attn72 -name "*.db" | head -1)
python - "$db" > gpurun_out/r6_attn72_kernel_ab.txt <<'PY'
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
agg = {}
for n, s, e in rows:
    if "attn" in n:
        k = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]
        d, c = agg.get(k, (0, 0)); agg[k] = (d + e - s, c + 1)
print("# SigLIP attention at B = 32 (64 images x 16 heads x 256 tokens x 72, fused q|k|v rows), isolated, rocprofv3 --kernel-trace averages")
for k, (d, c) in sorted(agg.items()):
    print(f"{k:50s} calls {c:5d} avg {d / c / 1e3:8.1f} us")
PY
rm -rf gpurun_out/r6_attn72
for i in $STEP_ROUNDS; do
  for v in 3 -1; do
    echo -n "LAP_ATTN_VARIANT=$v  " >> gpurun_out/r6_ab_attn_step.txt
    LAP_ATTN_VARIANT=$v python tools/step_only.py 8 3 2>/dev/null | tail -1 | cut -d'|' -f1 >> gpurun_out/r6_ab_attn_step.txt
  done
done
