"""forward products with a residual (out / down projections), isolated: time per launch (run twice: default and LAP_GEMM_NO_ASM_RES=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = []
for name, M, N, K, pad, biased in (("gemma out", 17920, 2048, 2048, 0, False), ("gemma down (padded act)", 17920, 2048, 16384, 64, False),
                                  ("gemma down 16384 rows", 16384, 2048, 16384, 64, False),
                                  ("siglip out", 16384, 1152, 1152, 0, True), ("siglip fc2 K=4352", 16384, 1152, 4352, 0, True)):
    a = rnd(M, K + pad)[:, :K]; w = rnd(N, K); r = rnd(M, N); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    b = torch.randn(N, device=dev) if biased else None
    fn = lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldb=K, ldc=N, bias=b, residual=r, ldr=N)
    t = min(timed(fn), timed(fn))
    forced = ""
    if os.environ.get("BENCH_FORCE_ASM") and M % 256 == 0:
        fn2 = lambda: hip.gemm(a, w, out, M=M, N=N, K=K, lda=a.stride(0), ldb=K, ldc=N, bias=b, residual=r, ldr=N, tile=14, ksplit=1)
        try:
            t2 = min(timed(fn2), timed(fn2)); forced = f" [asm forced {t2:7.1f} us]"
        except Exception as e:
            forced = " [asm n/a]"
    res.append(f"{name} {M}x{N}x{K}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s){forced}")
print(("NO_ASM_RES " if os.environ.get("LAP_GEMM_NO_ASM_RES") else "asm res    ") + "\n           ".join(res))
