"""The 128-column remainders of SigLIP's N = 1152 products (lap_gemm_bf16_ex's N cut): which tile runs [16384 x 128 x K] fastest?
Forward layout (fc2 forward: + f32 bias + residual, ldc = 1152) and data-gradient layout (qkv / fc1 data gradients), event-timed, isolated.
usage: python tools/probes/remainder_tiles.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lap_amd import hip

dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M = 16384


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K in (4352, 3456, 1152):
    x = rnd(M, K); w = rnd(1152, K); bias = torch.randn(1152, device=dev); res = rnd(M, 1152); out = torch.empty(M, 1152, dtype=torch.bfloat16, device=dev)
    for tile, ks in ((-1, 0), (6, 1), (6, 2), (6, 4), (0, 1), (0, 0), (16, 1), (18, 1), (17, 1)):
        try:
            t = timed(lambda: hip.gemm(x, w[1024:], out[:, 1024:], M=M, N=128, K=K, lda=K, ldb=K, ldc=1152, bias=bias[1024:], residual=res[:, 1024:], ldr=1152,
                                       tile=tile, ksplit=ks))
            print(f"fwd  K={K:5d} tile {tile:3d} ksplit {ks}: {t:7.1f} us")
        except Exception as e:  # noqa: BLE001
            print(f"fwd  K={K:5d} tile {tile:3d} ksplit {ks}: {type(e).__name__}")
    dy = rnd(M, K); wt = rnd(K, 1152); dx = torch.empty(M, 1152, dtype=torch.bfloat16, device=dev)
    for tile, ks in ((-1, 0), (6, 1), (6, 2), (6, 4), (0, 1), (0, 0), (12, 1), (12, 0)):
        try:
            t = timed(lambda: hip.gemm(dy, wt[:, 1024:], dx[:, 1024:], M=M, N=128, K=K, lda=K, ldb=1152, ldc=1152, a_kc=True, b_kc=False, tile=tile, ksplit=ks))
            print(f"dgrad K={K:5d} tile {tile:3d} ksplit {ks}: {t:7.1f} us")
        except Exception as e:  # noqa: BLE001
            print(f"dgrad K={K:5d} tile {tile:3d} ksplit {ks}: {type(e).__name__}")
