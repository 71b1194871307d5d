"""SigLIP attention forward + backward at the train step's shape (64 images x 256 tokens, 16 heads of 72, fused qkv buffer), a few launches:
the workload of tools/pmc_siglip_attn.sh (L2-to-fabric traffic per kernel) and a timing line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip

dev = "cuda"
B, NH, HD, T = 64, 16, 72, 256
W = NH * HD
qkv = (torch.randn(B * T, 3 * W, device=dev) * 0.3).bfloat16()
q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
kw = dict(scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0))
dqkv = torch.zeros_like(qkv)
do = (torch.randn(B * T, W, device=dev) * 0.3).bfloat16()
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GiB: flushes L2 and the Infinity Cache between launches
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
for i in range(n):
    big.zero_()
    ev[i][0].record()
    (o, _), lse = hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD, **kw)
    ev[i][1].record()
    hip.attention_bwd([q], [k], [v], [o], [do], lse, [T], [T], B, NH, NH, HD, dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]], **kw)
    ev[i][2].record()
torch.cuda.synchronize()
f = min(e[0].elapsed_time(e[1]) for e in ev[1:]) * 1e3
b = min(e[1].elapsed_time(e[2]) for e in ev[1:]) * 1e3
mb = B * T * W * 2 / 1e6
print(f"siglip attention, cold caches: fwd {f:.1f} us (q, k, v read + o written = {4 * mb:.0f} MB -> {4 * mb / f:.2f} TB/s), bwd {b:.1f} us")
