"""Timing ablations of the generic GEMM kernel's k-loop at the serving-prefill shapes (LAP_HIP_LIB_VARIANT=exp; results wrong by
construction): no in-loop LDS-DMA, no MFMA, neither.  us per launch in a replayed graph; K = 4608 so that the loop dominates."""
import os, sys
os.environ["LAP_HIP_LIB_VARIANT"] = "exp"
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


for M, N, K, tile in ((512, 1152, 4608, 17), (512, 3456, 4608, 16), (560, 2560, 4608, 6), (560, 32768, 2048, 15), (1600, 8192, 1024, 6)):
    a, w = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = [f"{M}x{N}x{K} t{tile}:"]
    for bits, label in ((0, "full"), (32, "no DMA"), (64, "no MFMA"), (96, "neither"), (8 + 16, "packed addr")):
        hip.call("lap_gemm_set_debug", bits)
        line.append(f"{label} {timed(lambda: hip.linear_fwd(a, w, out, tile=tile, ksplit=1)):6.1f}")
    hip.call("lap_gemm_set_debug", 0)
    print("  ".join(line) + "  us", flush=True)
