"""Tail-split cost model check: auto (model-chosen split of the last round) vs forced unsplit vs forced split counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
from tools.bench_kernels import timeit, rnd
for name, M, N, K in [("gemma out", 17920, 2048, 2048), ("qkv dgrad-like", 17920, 2048, 2560), ("down fwd", 17920, 2048, 16384),
                      ("sig qkv", 16384, 3456, 1152), ("sig out", 16384, 1152, 1152), ("sig fc1", 16384, 4304, 1152),
                      ("sig fc2", 16384, 1152, 4304), ("sig dqkv-like", 16384, 1152, 3456)]:
    a = rnd(M, K); w = rnd(N, K); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = [f"auto {timeit(lambda: hip.linear_fwd(a, w, out), iters=20) * 1e6:7.1f} us", f"unsplit {timeit(lambda: hip.linear_fwd(a, w, out, ksplit=1), iters=20) * 1e6:7.1f} us"]
    print(f"{name:16s} M={M} N={N} K={K}: " + " | ".join(res), flush=True)
