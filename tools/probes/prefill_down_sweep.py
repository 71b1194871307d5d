"""Gemma-2B down projection of the serving prefill (560 x 2048 x 16384) as split-K partials + fused reduce / residual / RMSNorm:
tile x split sweep INCLUDING the consumer (more slabs cost the reduce pass more).  us per (GEMM + reduce) in a replayed graph.
Also the qkv (560 x 2560 x 2048) and out (560 x 2048 x 2048) projections with their consumers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip

dev = torch.device("cuda")
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()


def timed(fn, n=10, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


scratch = hip._gemm_scratch(dev)
M, N, K = 560, 2048, 16384
ws = [rnd(N, K) for _ in range(4)]      # rotate: every layer has its own weights (HBM-cold)
a, res = rnd(M, K), rnd(M, N)
gamma = torch.zeros(N, device=dev)
i = [0]
for tile, splits in ((5, (4, 8, 16)), (15, (4, 8, 16)), (19, (4, 8, 16)), (6, (4, 8, 16)), (16, (8, 16, 32))):
    line = [f"down t{tile}:"]
    for ks in splits:
        def fn():
            i[0] = (i[0] + 1) % 4
            part, k2 = hip.linear_partials(a, ws[i[0]], scratch, ksplit=ks, tile=tile)
            hip.fused_reduce_norm(part, k2, M, N, residual=res, norm=1, gamma=gamma)
        def fn_g():
            i[0] = (i[0] + 1) % 4
            hip.linear_partials(a, ws[i[0]], scratch, ksplit=ks, tile=tile)
        try:
            line.append(f"k{ks}: {timed(fn):6.1f} (gemm {timed(fn_g):6.1f})")
        except Exception as e:   # noqa: BLE001
            line.append(f"k{ks}: {type(e).__name__}")
    print("  ".join(line), flush=True)
