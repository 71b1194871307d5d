"""Timing probe (LAP_HIP_LIB_VARIANT=exp): the serving-prefill GEMM tiles with their operand tiles fetched 1 KiB contiguous per LDS-DMA
piece (what fragment-packed operand images would cost) against the row-major fetch (8 rows x 128 B per piece).  Results of the
packed variants are wrong by construction; us per launch in a replayed graph."""
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


for name, M, N, K, tile in (("sig_out", 512, 1152, 1152, 17), ("sig_qkv", 512, 3456, 1152, 16), ("sig_fc1", 512, 4352, 1152, 16), ("sig_fc2", 512, 1152, 4352, 17),
                            ("gem_qkv", 560, 2560, 2048, 16), ("gem_out", 560, 2048, 2048, 16), ("gem_gu", 560, 32768, 2048, 15), ("gem_down", 560, 2048, 16384, 16),
                            ("gem_gu6", 560, 32768, 2048, 6), ("gem_gu16", 560, 32768, 2048, 16)):
    a, w = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = [f"{name:9s} {M}x{N}x{K} t{tile}:"]
    for bits, label in ((0, "row-major"), (8, "B packed"), (24, "A+B packed")):
        hip.call("lap_gemm_set_debug", bits)
        line.append(f"{label} {timed(lambda: hip.linear_fwd(a, w, out, tile=tile, ksplit=1)):6.1f}")
    hip.call("lap_gemm_set_debug", 0)
    print("  ".join(line) + "  us", flush=True)
