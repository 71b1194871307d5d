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
# weight gradient of gate|up: dW[32768, 2048] = dy[17920, 32768]^T x[17920, 2048]
M, N, K = 32768, 2048, 17920
x = rnd(K, N)
for pad in (0, 64, 128, 256, 512, 1024):
    dy = rnd(K, M + pad)[:, :M]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    for tile in (12, 14):
        t = timeit(lambda: hip.gemm(dy, x, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=N, ldc=N, a_kc=False, b_kc=False, tile=tile, ksplit=1))
        print(f"wgrad pad {pad:5d} tile {tile}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# data gradient of gate|up: dx[17920, 2048] = dy[17920, 32768] W[32768, 2048]
M, N, K = 17920, 2048, 32768
w = rnd(K, N)
for pad in (0, 64, 256):
    dy = rnd(M, K + pad)[:, :K]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(dy, w, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=N, ldc=N, a_kc=True, b_kc=False, tile=5))
    print(f"dgrad pad {pad:5d} tile 12 + tail split: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# forward of gate|up writes gu[17920, 32768]: does the output stride matter?
M, N, K = 17920, 32768, 2048
a, b = rnd(M, K), rnd(N, K)
for pad in (0, 64, 256):
    out = torch.empty(M, N + pad, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N + pad, tile=14, ksplit=1))
    print(f"fwd out pad {pad:5d} asm: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
