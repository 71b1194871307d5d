"""SigLIP-width products (N = 1152 = 4.5 tiles of 256) at B = 32: time per launch, isolated (run twice: default and LAP_GEMM_NO_NSPLIT=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M = 16384
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = []
for name, K, kind in (("out fwd", 1152, "ntbr"), ("fc2 fwd", 4304, "ntbr"), ("qkv dgrad", 3456, "nn"), ("fc1 dgrad", 4304, "nn"), ("head-like fwd", 1152, "ntb")):
    N = 1152
    if kind == "nn":
        a, w = rnd(M, K), rnd(K, N); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: hip.linear_dgrad(a, w, out)
    else:
        a, w = rnd(M, K), rnd(N, K); b = torch.randn(N, device=dev); r = rnd(M, N) if kind == "ntbr" else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: hip.linear_fwd(a, w, out, bias=b, residual=r)
    t = timed(fn)
    res.append(f"{name} K={K}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s)")
print(("NO_NSPLIT " if os.environ.get("LAP_GEMM_NO_NSPLIT") else "nsplit    ") + " | ".join(res))
