"""Micro-benchmark of the skinny-M fused projections at the LAP-3B action-expert shapes: every call uses a different
weight matrix out of a > 256 MB rotation, so the stream really comes from HBM as in the denoise loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).bfloat16()
D, NH, HD, H = 1024, 8, 256, 4096


def timeit(fn, n=36, reps=10):
    """`n` dependent-free launches captured into one HIP graph (host launch cost out of the picture), replayed `reps` times."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for i in range(3): fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n): fn(i)
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): gr.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


for M in (50, 16):
    x = rnd(M, D); mod = rnd(1, 3 * D, sc=0.3)
    pos = (torch.arange(M, device=dev, dtype=torch.int32) + 560).view(1, M).contiguous()
    tab = hip.rope_table(pos, 1, M, M, 0, HD)
    nw = 36
    wq = [rnd((NH + 2) * HD, D, sc=0.03) for _ in range(nw)]
    wg = [rnd(2 * H, D, sc=0.03) for _ in range(nw)]
    wo = [rnd(D, NH * HD, sc=0.03) for _ in range(nw)]
    wd = [rnd(D, H, sc=0.03) for _ in range(nw)]
    a2 = rnd(M, NH * HD); a4 = rnd(M, H)
    res = {}
    res["qkv"] = timeit(lambda i: hip.serve_qkv_rope(x, mod, 0, M, wq[i % nw], tab, NH, HD, 0.0625))
    res["gate_up"] = timeit(lambda i: hip.serve_gate_up(x, mod, 0, M, wg[i % nw]))
    for ft in (1, 2):
        hip.serve_set_variant(ft)
        res[f"o_ft{ft}"] = timeit(lambda i: hip.serve_proj_residual(a2, wo[i % nw], x, mod[:, 2 * D:], 0, M))
        res[f"down_ft{ft}"] = timeit(lambda i: hip.serve_proj_residual(a4, wd[i % nw], x, mod[:, 2 * D:], 0, M))
    hip.serve_set_variant(1)
    # old path pieces for comparison
    scratch = hip._gemm_scratch(torch.device(dev))
    res["old_gemm_partials_down"] = timeit(lambda i: hip.linear_partials(a4, wd[i % nw], scratch))
    print(f"M={M}: " + "  ".join(f"{k} {v:.2f}us" for k, v in res.items()), flush=True)

# ---- where does a block's time go?  M = 16 (one token tile: 64 blocks, one per CU): weights rotating (HBM) vs one
# weight matrix re-used (L2 / MALL warm after the first replay)
M = 16
x = rnd(M, D); mod = rnd(1, 3 * D, sc=0.3); a2 = rnd(M, NH * HD); a4 = rnd(M, H)
wo = [rnd(D, NH * HD, sc=0.03) for _ in range(36)]; wd = [rnd(D, H, sc=0.03) for _ in range(36)]
for name, a, ws in (("o", a2, wo), ("down", a4, wd)):
    cold = timeit(lambda i: hip.serve_proj_residual(a, ws[i % 36], x, mod[:, 2 * D:], 0, M))
    warm = timeit(lambda i: hip.serve_proj_residual(a, ws[0], x, mod[:, 2 * D:], 0, M))
    print(f"M=16 {name}: rotating weights {cold:.2f} us, same weights {warm:.2f} us", flush=True)
