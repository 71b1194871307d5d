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
# weight gradient of the down projection: dW[2048, 16384] = dy[17920, 2048]^T act[17920, 16384]
M, N, K = 2048, 16384, 17920
dy = rnd(K, M)
for pad in (0, 64, 256):
    x = rnd(K, N + pad)[:, :N]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    for tile in (12, 14):
        t = timeit(lambda: hip.gemm(dy, x, out, M=M, N=N, K=K, lda=M, ldb=x.stride(0), ldc=N, a_kc=False, b_kc=False, tile=tile, ksplit=1))
        print(f"down wgrad act pad {pad:4d} tile {tile}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# forward of the down projection reads act[17920, 16384] as A (K = 16384): tile 5 (tail split) with residual
M, N, K = 17920, 2048, 16384
w = rnd(N, K); res = rnd(M, N)
for pad in (0, 64, 256):
    a = rnd(M, K + pad)[:, :K]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldb=K, ldc=N, residual=res, ldr=N))
    print(f"down fwd act pad {pad:4d}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# data gradient of the down projection writes dact[17920, 16384] (N = 16384), reads W[2048, 16384]
M, N, K = 17920, 16384, 2048
dy = rnd(M, K); w = rnd(K, N)
for pad in (0, 64):
    out = torch.empty(M, N + pad, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(dy, w, out, M=M, N=N, K=K, lda=K, ldb=N, ldc=N + pad, a_kc=True, b_kc=False))
    print(f"down dgrad out pad {pad:4d}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# forward of gate|up reads x[17920, 2048] and W[32768, 2048] (K = 2048: 4 KB strides)
M, N, K = 17920, 32768, 2048
for pad in (0, 64):
    a = rnd(M, K + pad)[:, :K]; b = rnd(N, K + pad)[:, :K]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(a, b, out, M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N))
    print(f"gate-up fwd operand pad {pad:4d}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# weight gradient of gate|up with BOTH operands padded
M, N, K = 32768, 2048, 17920
for pa, pb in ((64, 0), (64, 64)):
    dy = rnd(K, M + pa)[:, :M]; x = rnd(K, N + pb)[:, :N]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    t = timeit(lambda: hip.gemm(dy, x, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=x.stride(0), ldc=N, a_kc=False, b_kc=False))
    print(f"gate-up wgrad pads {pa} {pb}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
