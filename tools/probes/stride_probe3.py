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
# data gradient of the down projection: dact[17920, 16384] = dy[17920, 2048] W[2048, 16384]  (B rows: 32 KiB)
M, N, K = 17920, 16384, 2048
dy = rnd(M, K)
for pad in (0, 64):
    w = rnd(K, N + pad)[:, :N]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (12, 14):
        t = timeit(lambda: hip.gemm(dy, w, out, M=M, N=N, K=K, lda=K, ldb=w.stride(0), ldc=N, a_kc=True, b_kc=False, tile=tile, ksplit=1))
        print(f"down dgrad W pad {pad:3d} tile {tile}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# weight gradients whose operands have 4 KiB rows: qkv dW[2560, 2048] = dqkv[17920, 2560]^T h[17920, 2048]
M, N, K = 2560, 2048, 17920
for pa, pb in ((0, 0), (0, 64), (64, 64)):
    dy = rnd(K, M + pa)[:, :M]; x = rnd(K, N + pb)[:, :N]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    t = timeit(lambda: hip.gemm(dy, x, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=x.stride(0), ldc=N, a_kc=False, b_kc=False))
    print(f"qkv wgrad pads {pa} {pb}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
# gate|up data gradient: W[32768, 2048] rows are 4 KiB
M, N, K = 17920, 2048, 32768
dy = rnd(M, K + 64)[:, :K]
for pad in (0, 64):
    w = rnd(K, N + pad)[:, :N]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(dy, w, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=w.stride(0), ldc=N, a_kc=True, b_kc=False))
    print(f"gate-up dgrad W pad {pad:3d}: {t:8.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
