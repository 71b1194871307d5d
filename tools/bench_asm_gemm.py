"""The assembly forward-layout GEMM (csrc/gemm_nt_asm.s) against the HIP kernels: bitwise check + timing.
Usage: python tools/bench_asm_gemm.py [check|bench]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()

def asm(a, b, out):
    M, K = a.shape; N = b.shape[0]
    hip.call("lap_gemm_nt_asm", hip._p(a), hip._p(b), hip._p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), hip._stream())
    return out

def timeit(fn, n=10):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "check"
shapes = [(256, 512, 256), (512, 256, 384), (1024, 768, 1152), (2304, 1280, 512), (4096, 4096, 4096)] if mode == "check" else \
         [(17920, 32768, 2048), (17920, 2048, 16384), (17920, 2048, 2048), (17920, 2560, 2048), (8192, 8192, 8192), (4096, 4096, 4096), (16384, 3584, 1152)]
for M, N, K in shapes:
    a, b = rnd(M, K), rnd(N, K)
    ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16); out = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    hip.gemm(a, b, ref, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=10)
    asm(a, b, out)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    err = (out.float() - ref.float()).abs().max().item()
    line = f"{M:>6} {N:>6} {K:>6}: bitwise {same} max|diff| {err:.3g}"
    if mode != "check" or same:
        t0 = timeit(lambda: hip.gemm(a, b, ref, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=10))
        t1 = timeit(lambda: asm(a, b, out))
        fl = 2.0 * M * N * K
        line += f" | tile10 {t0:8.1f} us {fl/t0/1e6:6.0f} TF/s | asm {t1:8.1f} us {fl/t1/1e6:6.0f} TF/s"
    else:
        bad = (out != ref).nonzero()
        line += f" | first mismatches {bad[:4].tolist()} n_bad {bad.shape[0]}"
    print(line, flush=True)
