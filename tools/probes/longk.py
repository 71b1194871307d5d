"""two full rounds (16384 x 2048 outputs) of the long contractions: assembly kernel vs the HIP tiles, padded A rows as in the engine"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()
def timeit(fn, n=10):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for lay, M, N, K, pa in (("nn", 16384, 2048, 32768, 64), ("nn", 17920, 2048, 32768, 64), ("nt", 16384, 2048, 16384, 64), ("nt", 17920, 2048, 16384, 64), ("nn", 1536, 2048, 32768, 64), ("nt", 1536, 2048, 16384, 64)):
    a = rnd(M, K + pa)[:, :K]
    b = rnd(K, N) if lay == "nn" else rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=True, b_kc=lay == "nt")
    line = f"{lay} {M} {N} {K}:"
    for name, extra in (("auto", {}), ("asm", dict(tile=14, ksplit=1)), ("hip", dict(tile=5, ksplit=1))):
        if name == "asm" and M % 256: continue
        try:
            t = min(timeit(lambda: hip.gemm(a, b, out, **kw, **extra)) for _ in range(2))
            line += f"  {name} {t:7.1f} us {2.0*M*N*K/t/1e6:5.0f} TF"
        except Exception as e:
            line += f"  {name} n/a"
    print(line, flush=True)
