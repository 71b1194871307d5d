"""Is SigLIP's attention bound by the access pattern of the fused q|k|v rows (a head's row = 144 contiguous bytes at a 6912-byte stride)?
The same 1024 heads x 256 tokens x 72 once in the tower's token-major layout and once head-major (every head a contiguous [256, 72] block,
emulated as 1024 single-head images); rocprofv3 --kernel-trace averages tell the two launches apart by their grid.
usage: rocprofv3 --kernel-trace -d out -o r -- python tools/probes/siglip_attn_layout.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lap_amd import hip

B, T, NH, HD = 64, 256, 16, 72
W = NH * HD
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * T, 3 * W, device="cuda", generator=g) * 0.7).bfloat16()
do = torch.randn(B * T, W, device="cuda", generator=g).bfloat16()
q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
# head-major copies: [B * NH * T, 72]
hm = lambda x: x.reshape(B, T, NH, HD).permute(0, 2, 1, 3).contiguous().view(B * NH * T, HD)
qh, kh, vh, doh = hm(q), hm(k), hm(v), hm(do)
ev = lambda: torch.cuda.Event(enable_timing=True)
for name, run in (("token-major", 0), ("head-major", 1), ("token-major", 0), ("head-major", 1)):
    ts = []
    for it in range(12):
        a, b2, c = ev(), ev(), ev()
        a.record()
        if run == 0:
            (o, _), lse = hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD, scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0))
            b2.record()
            dqkv = torch.empty_like(qkv)
            hip.attention_bwd([q], [k], [v], [o], [do], lse, [T], [T], B, NH, NH, HD, scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0),
                              dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]])
        else:
            (o, _), lse = hip.attention_fwd([qh], [kh], [vh], [T], [T], B * NH, 1, 1, HD, scale=HD ** -0.5)
            b2.record()
            hip.attention_bwd([qh], [kh], [vh], [o], [doh], lse, [T], [T], B * NH, 1, 1, HD, scale=HD ** -0.5)
        c.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append((a.elapsed_time(b2) * 1e3, b2.elapsed_time(c) * 1e3))
    print(f"{name:12s} forward {sum(t[0] for t in ts) / len(ts):7.1f} us   backward (dQ + dK/dV launches) {sum(t[1] for t in ts) / len(ts):7.1f} us")
