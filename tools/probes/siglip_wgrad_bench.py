"""SigLIP weight gradients (the third stream's GEMMs: dWt[out][in] = dy^T x over 16384 token rows) alone on the chip: the production route
(hip.linear_wgrad -> lap_gemm_wgrad_bf16: the ping-pong HIP tile with a K split + the reduce) per shape, and tile / split sweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip

dev = "cuda"
rows = 16384
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
tot = 0.0
for name, out, inn in (("wqkv", 3456, 1152), ("wo", 1152, 1152), ("w1", 4352, 1152), ("w2", 1152, 4352)):
    dy = (torch.randn(rows, out, device=dev) * 0.1).bfloat16()
    x = (torch.randn(rows, inn, device=dev) * 0.5).bfloat16()
    g = torch.zeros(out, inn, dtype=torch.bfloat16, device=dev)
    ss = torch.zeros(1, device=dev)
    fl = 2.0 * rows * out * inn
    t = timeit(lambda: hip.linear_wgrad_sumsq(dy, x, g, ss))
    tot += t
    line = f"{name:5s} dWt[{out}][{inn}] over {rows} rows: production route {t:7.1f} us = {fl / t / 1e6:6.0f} TF/s |"
    g32 = torch.zeros(out, inn, dtype=torch.float32, device=dev)
    for tile, ks in ((12, 2), (12, 3), (12, 4), (12, 6), (12, 8), (10, 4), (10, 8)):
        try:
            t2 = timeit(lambda: hip.linear_wgrad(dy, x, g32, accum=False, ksplit=ks, tile=tile))
            line += f" t{tile}x{ks} {t2:6.1f}"
        except Exception as e:  # noqa: BLE001
            line += f" t{tile}x{ks} n/a"
    print(line, flush=True)
print(f"sum of the four: {tot:.1f} us per layer, {tot * 27 / 1e3:.2f} ms per step; at 1.2 PF: {2.0 * rows * (3456 * 1152 + 1152 * 1152 + 2 * 4352 * 1152) / 1.2e15 * 1e6:.1f} us per layer")
