"""fp8 vs bf16 GEMM at the LAP-3B shapes (forward layout)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for name, m, n, k in (("gateup fwd", 17920, 32768, 2048), ("down fwd", 17920, 2048, 16384), ("gateup dgrad", 17920, 2048, 32768),
                      ("down dgrad", 17920, 16384, 2048), ("qkv fwd", 17920, 2560, 2048), ("square 8192", 8192, 8192, 8192)):
    a = rnd(m, k); w = rnd(n, k); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    a8, sa = hip.quantize_fp8(a); w8, w8t, sw = hip.quantize_fp8_weight(w)
    tb = timeit(lambda: hip.linear_fwd(a, w, out))
    tf = timeit(lambda: hip.gemm_fp8(a8, sa, w8, sw, out))
    tq = timeit(lambda: hip.quantize_fp8(a))
    tw = timeit(lambda: hip.quantize_fp8_weight(w))
    fl = 2 * m * n * k
    print(f"{name:13s} bf16 {tb*1e3:7.3f} ms {fl/tb/1e12:5.0f} TF | fp8 {tf*1e3:7.3f} ms {fl/tf/1e12:5.0f} TF | quant act {tq*1e3:.3f} ms, weight {tw*1e3:.3f} ms", flush=True)
