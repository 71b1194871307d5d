"""weight-gradient product of gate|up (dW [32768, 2048] = dgu^T x over 17920 rows): padded vs unpadded rows of dgu, f32 output"""
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
for (M, N, K) in ((32768, 2048, 17920), (2048, 16384, 17920)):
    for pa, pb in ((0, 0), (64, 0), (0, 64), (64, 64)):
        a = rnd(K, M + pa)[:, :M]; b = rnd(K, N + pb)[:, :N]
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
        kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=False, b_kc=False)
        t = min(timeit(lambda: hip.gemm(a, b, out, **kw)) for _ in range(2))
        print(f"tn {M} {N} {K} pad A {pa} B {pb}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
