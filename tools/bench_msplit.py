"""560-tile products (M = 17920 x N = 2048) of the train step, isolated: time per launch (run twice: default and LAP_GEMM_NO_MSPLIT=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M, N = 17920, 2048
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = []
for name, K, kind in (("gate-up dgrad", 32768, "nn"), ("qkv dgrad", 2560, "nn"), ("out dgrad", 2048, "nn"), ("plain fwd", 2048, "nt"), ("plain fwd", 16384, "nt")):
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if kind == "nn":
        a, w = rnd(M, K), rnd(K, N)
        fn = lambda: hip.linear_dgrad(a, w, out)
    else:
        a, w = rnd(M, K), rnd(N, K)
        fn = lambda: hip.linear_fwd(a, w, out)
    t = min(timed(fn), timed(fn))
    res.append(f"{name} K={K}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s)")
print(("NO_MSPLIT " if os.environ.get("LAP_GEMM_NO_MSPLIT") else "msplit    ") + " | ".join(res))
