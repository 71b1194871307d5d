"""Is the assembly GEMM loop's ~1.65 PF the chip's power envelope?  The production forward kernel on random operands (the benchmark's), on operands that are all
zeros, and on a constant: same instruction stream, same bytes, different toggle rates.  usage: python tools/probes/asm_power.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, N, K in ((8192, 8192, 8192), (17920, 32768, 2048)):
    for name, mk in (("random uniform", lambda *s: (torch.rand(*s, device=dev) - 0.5).bfloat16()), ("zeros", lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)),
                     ("constant 1.0", lambda *s: torch.ones(*s, device=dev, dtype=torch.bfloat16)), ("random normal x 0.02", lambda *s: (torch.randn(*s, device=dev) * 0.02).bfloat16())):
        a, b = mk(M, K), mk(N, K)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = min(timeit(lambda: hip.gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=14, ksplit=1)) for _ in range(2))
        print(f"nt {M} x {N} x {K}  {name:22s} {2.0 * M * N * K / t / 1e6:6.0f} TF/s")
