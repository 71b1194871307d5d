"""Where do the serving-prefill GEMM launches spend their ~10 us?  us per launch in a replayed graph of 20 back-to-back calls:
K sweep at fixed M x N (fixed cost vs per-k-step cost), N sweep, and the floor of a trivial kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip

dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()


def timed(fn, n=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


x1 = torch.zeros(64, device=dev)
print(f"trivial kernel (cast of 64 floats): {timed(lambda: hip.cast_f32_to_bf16(x1)):.2f} us per launch")
for M, N, tile in ((512, 1152, 17), (512, 1152, 16), (512, 3456, 16), (560, 2048, 16), (560, 2560, 6)):
    line = [f"M{M} N{N} t{tile}:"]
    for K in (64, 256, 512, 1152, 2048, 4608):
        a, w = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        # rotate over 8 weight copies so that the weights are HBM-cold like in the model (each layer has its own)
        ws = [rnd(N, K) for _ in range(8)]
        i = [0]
        def fn():
            i[0] = (i[0] + 1) % 8
            hip.linear_fwd(a, ws[i[0]], out, tile=tile, ksplit=1)
        t_cold = timed(fn)
        t_hot = timed(lambda: hip.linear_fwd(a, w, out, tile=tile, ksplit=1))
        line.append(f"K{K}: {t_hot:5.1f}/{t_cold:5.1f}")
    print("  ".join(line) + "   (us hot / rotating weights)", flush=True)
# elementwise floors
x = rnd(512, 1152)
g_, b_ = torch.ones(1152, device=dev), torch.zeros(1152, device=dev)
print(f"layernorm_fwd 512x1152: {timed(lambda: hip.layernorm_fwd(x, g_, b_)):.2f} us")
x2 = rnd(560, 2048)
sc = torch.zeros(2048, device=dev)
print(f"rmsnorm_fwd 560x2048: {timed(lambda: hip.rmsnorm_fwd(x2, scale=sc, save_rstd=False)):.2f} us")
