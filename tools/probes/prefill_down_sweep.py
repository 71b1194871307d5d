"""Serving prefill, Gemma down projection (560 x 2048 over K = 16384) as split-K partials + fused reduce / residual / next norm:
tile x split candidates, timed as hipGraph replays of 20 back-to-back pairs.  Also the qkv (560 x 2560 x 2048) and out (560 x 2048 x 2048)
projections' partial routes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()


def timed(fn, n=20, reps=10):
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


scratch = hip._gemm_scratch(torch.device(dev))
for name, (M, N, K) in {"down": (560, 2048, 16384), "out": (560, 2048, 2048), "qkv": (560, 2560, 2048)}.items():
    a, w, res = rnd(M, K), rnd(N, K) * 0.05, rnd(M, N)
    gamma = torch.randn(N, device=dev)
    ref = None
    line = [f"{name} {M}x{N}x{K}:"]
    for tile in (5, 6, 15, 19, 16, 18):
        for ks in (1, 2, 4, 8, 12, 16, 24, 32):
            if K // ks < 512 or ks * M * N > scratch.numel():
                continue
            def fn():
                part, k2 = hip.linear_partials(a, w, scratch, ksplit=ks, tile=tile)
                return hip.fused_reduce_norm(part, k2, M, N, residual=res, norm=1, gamma=gamma)
            try:
                t = timed(fn)
            except Exception as e:   # noqa: BLE001
                line.append(f"t{tile}/k{ks}:{type(e).__name__}")
                continue
            x, h = fn()
            if ref is None:
                ref = x.float().clone()
            err = ((x.float() - ref).abs().max() / ref.abs().max()).item()
            line.append(f"t{tile}/k{ks}:{t:.1f}us" + ("" if err < 2e-2 else f"(err {err:.1e})"))
    print(" ".join(line), flush=True)
