"""First stage at which the chain differs from the separate launches (depth 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
DEV = "cuda"
def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)
B, S, Tp, D, NH, HD, H = 1, 50, 816, 1024, 8, 256, 4096
M = S
for depth in (1, 2):
    x = rnd(M, D, seed=1)
    mod = rnd(1, (2 * depth + 1) * 3 * D, scale=0.3, seed=2)
    W = [(rnd((NH + 2) * HD, D, scale=D ** -0.5, seed=10 + 4 * l), rnd(D, NH * HD, scale=(NH * HD) ** -0.5, seed=11 + 4 * l),
          rnd(2 * H, D, scale=D ** -0.5, seed=12 + 4 * l), rnd(D, H, scale=H ** -0.5, seed=13 + 4 * l)) for l in range(depth)]
    cache = [(rnd(Tp, HD, scale=0.25, seed=100 + l), rnd(Tp, HD, seed=200 + l)) for l in range(depth)]
    pos = (torch.arange(M, device=DEV, dtype=torch.int32) + Tp).view(1, M).contiguous()
    tab = hip.rope_table(pos, 1, M, M, 0, HD)
    kinfo = torch.full((B, Tp + S), 3 << 24, dtype=torch.int32, device=DEV); kinfo[:, Tp:] = (4 << 24) | 0x800001
    qinfo = torch.full((B, S), (6 << 24) | 0x800001, dtype=torch.int32, device=DEV)
    slot = lambda j: mod[:, j * 3 * D:(j + 1) * 3 * D]
    xx = x
    for l, (wqkv, wo, wgu, wd) in enumerate(W):
        q, k, v = hip.serve_qkv_rope(xx, slot(2 * l), 0, S, wqkv, tab, NH, HD, HD ** -0.5)
        o, _ = hip.attention_fwd([None, q], [cache[l][0], k], [cache[l][1], v], [0, S], [Tp, S], B, NH, 1, HD, qinfo, kinfo, need_lse=False)
        xa = hip.serve_proj_residual(o[1], wo, xx, slot(2 * l)[:, 2 * D:], 0, S)
        act = hip.serve_gate_up(xa, slot(2 * l + 1), 0, S, wgu)
        xx = hip.serve_proj_residual(act, wd, xa, slot(2 * l + 1)[:, 2 * D:], 0, S)
    ctr = hip.serve_chain_counters(DEV)
    keep = {}
    out = hip.serve_chain(x, mod, 3 * D, W, cache, tab, qinfo, kinfo, B, S, NH, HD, H, Tp, HD ** -0.5, ctr, keep=keep)
    torch.cuda.synchronize()
    print(f"depth {depth}: failed={hip.serve_chain_failed(ctr)}")
    for nm, ref in (("q", q), ("k", k), ("v", v), ("o", o[1]), ("xa", xa), ("act", act), ("out", xx)):
        got = out if nm == "out" else keep[nm]
        d = (got.float() - ref.float()).abs()
        nz = (d > 0).nonzero()
        print(f"  {nm:4s} equal={torch.equal(got, ref)} ndiff={int((d > 0).sum())} max={d.max().item():.4g}", ("first " + str(nz[0].tolist()) + " last " + str(nz[-1].tolist())) if len(nz) else "")
