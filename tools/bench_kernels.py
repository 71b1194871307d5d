"""Micro-benchmarks of the hot kernels at LAP-3B shapes (GPU box). Prints TF/s or GB/s per kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, dtype=torch.bfloat16):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(dtype)


def bench_gemms():
    M = 17920  # 32 x 560 prefix rows
    shapes = [
        ("qkv   fwd", "fwd", M, 2560, 2048), ("out   fwd", "fwd", M, 2048, 2048),
        ("gateup fwd", "fwd", M, 32768, 2048), ("down  fwd", "fwd", M, 2048, 16384),
        ("lm head dgrad", "dgrad", 1504, 2048, 257152), ("prefill down", "fwd", 816, 2048, 16384),
        ("gateup dgrad", "dgrad", M, 2048, 32768), ("down  dgrad", "dgrad", M, 16384, 2048),
        ("gateup wgrad", "wgrad", M, 32768, 2048), ("down  wgrad", "wgrad", M, 2048, 16384),
        ("siglip fc1", "fwd", 16384, 4304, 1152), ("siglip qkv", "fwd", 16384, 3456, 1152),
        ("expert gateup", "fwd", 1600, 8192, 1024), ("lm head", "fwd", 1504, 257152, 2048),
        ("square 4096", "fwd", 4096, 4096, 4096), ("square 8192", "fwd", 8192, 8192, 8192),
    ]
    for name, kind, m, n, k in shapes:
        if kind == "fwd":
            a = rnd(m, k); w = rnd(n, k); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            fn = lambda: hip.linear_fwd(a, w, out)
        elif kind == "dgrad":
            a = rnd(m, k); w = rnd(k, n); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            fn = lambda: hip.linear_dgrad(a, w, out)
        else:
            dy = rnd(m, n); x = rnd(m, k); out = torch.empty(n, k, dtype=torch.float32, device=dev)
            fn = lambda: hip.linear_wgrad(dy, x, out)
        res = []
        for tile in ((10, 13) if kind == 'fwd' else (10, 12)):
            if kind == "fwd":
                fn = lambda: hip.linear_fwd(a, w, out, tile=tile)
            elif kind == "dgrad":
                fn = lambda: hip.linear_dgrad(a, w, out, tile=tile)
            else:
                fn = lambda: hip.linear_wgrad(dy, x, out, tile=tile)
            t = timeit(fn, iters=10)
            res.append(f"t{tile}: {t*1e3:7.3f} ms {2*m*n*k/t/1e12:6.0f} TF")
        # calibration only: the vendor library (torch.matmul -> hipBLASLt) on the same product
        if kind == "fwd":
            vf = lambda: torch.matmul(a, w.t(), out=out)
        elif kind == "dgrad":
            vf = lambda: torch.matmul(a, w, out=out)
        else:
            o16 = torch.empty(n, k, dtype=torch.bfloat16, device=dev)
            vf = lambda: torch.matmul(dy.t(), x, out=o16)
        t = timeit(vf, iters=10)
        res.append(f"hipBLASLt: {t*1e3:7.3f} ms {2*m*n*k/t/1e12:6.0f} TF")
        print(f"gemm {name:14s} {kind:5s} M={m} N={n} K={k}: " + " | ".join(res), flush=True)



def bench_attn():
    B, NH, HD, Tp, S = 32, 8, 256, 560, 50
    q0 = rnd(B, Tp, NH * HD); q1 = rnd(B, S, NH * HD); k0 = rnd(B, Tp, HD); k1 = rnd(B, S, HD); v0 = rnd(B, Tp, HD); v1 = rnd(B, S, HD)
    fl = 4 * B * NH * (Tp + S) ** 2 * HD
    t = timeit(lambda: hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD))
    print(f"attn fwd gemma B32 T610 hd256: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF/s")
    (o0, o1), lse = hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD)
    t = timeit(lambda: hip.attention_bwd([q0, q1], [k0, k1], [v0, v1], [o0, o1], [q0, q1], lse, [Tp, S], [Tp, S], B, NH, 1, HD))
    print(f"attn bwd gemma: {t*1e3:.3f} ms {2.5*fl/t/1e12:.1f} TF/s (2.5x fwd flops)")
    B, NH, HD, T = 64, 16, 72, 256
    q = rnd(B, T, NH * HD); k = rnd(B, T, NH * HD); v = rnd(B, T, NH * HD)
    fl = 4 * B * NH * T * T * HD
    t = timeit(lambda: hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD))
    print(f"attn fwd siglip B64 T256 hd72: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF/s")
    (o, _), lse = hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD)
    t = timeit(lambda: hip.attention_bwd([q], [k], [v], [o], [q], lse, [T], [T], B, NH, NH, HD))
    print(f"attn bwd siglip: {t*1e3:.3f} ms {2.5*fl/t/1e12:.1f} TF/s")


def bench_mem():
    rows, D = 17920, 2048
    x = rnd(rows, D); sc = torch.zeros(D, device=dev)
    t = timeit(lambda: hip.rmsnorm_fwd(x, scale=sc))
    print(f"rmsnorm fwd: {t*1e6:.1f} us {2*rows*D*2/t/1e9:.0f} GB/s")
    y, rstd = hip.rmsnorm_fwd(x, scale=sc); ds = torch.zeros(D, device=dev)
    t = timeit(lambda: hip.rmsnorm_bwd(x, x, rstd, scale=sc, dscale=ds))
    print(f"rmsnorm bwd: {t*1e6:.1f} us {3*rows*D*2/t/1e9:.0f} GB/s")
    xs = rnd(16384, 1152); gam = torch.ones(1152, device=dev); bet = torch.zeros(1152, device=dev)
    t = timeit(lambda: hip.layernorm_fwd(xs, gam, bet))
    print(f"layernorm fwd (16384 x 1152): {t*1e6:.1f} us {2*16384*1152*2/t/1e9:.0f} GB/s")
    ys, mean_, rstd_ = hip.layernorm_fwd(xs, gam, bet)
    dg = torch.zeros(1152, device=dev); db = torch.zeros(1152, device=dev)
    t = timeit(lambda: hip.layernorm_bwd(xs, xs, gam, mean_, rstd_, dg, db))
    print(f"layernorm bwd: {t*1e6:.1f} us {3*16384*1152*2/t/1e9:.0f} GB/s")
    gu = rnd(rows, 32768)
    t = timeit(lambda: hip.geglu_fwd(gu))
    print(f"geglu fwd: {t*1e6:.1f} us {rows*16384*2*3/t/1e9:.0f} GB/s")
    act = hip.geglu_fwd(gu)
    t = timeit(lambda: hip.geglu_bwd(gu, act))
    print(f"geglu bwd: {t*1e6:.1f} us {rows*16384*2*5/t/1e9:.0f} GB/s")
    pos = torch.arange(610, device=dev, dtype=torch.int32)[None].repeat(32, 1).contiguous()
    qkv = rnd(rows, 2560)
    t = timeit(lambda: hip.rope_split_fwd(qkv, pos, 32, 560, 610, 0, 8, 256, 0.0625))
    print(f"rope fwd: {t*1e6:.1f} us {rows*2560*2*2/t/1e9:.0f} GB/s")
    n = 400_000_000
    p = torch.zeros(n, device=dev); g = torch.ones(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); e = torch.zeros(n, device=dev)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    sc8 = torch.tensor([1.0, 1e-4, 0.1, 0.05, 0.99, 1.0, 0, 0], device=dev)
    t = timeit(lambda: hip.adamw_ema(p, m, v, e, g, p16, sc8, 0.9, 0.95, 1e-8, 1e-4, 1.0), iters=5)
    print(f"adamw+ema: {t*1e3:.2f} ms {n*(5*4+4*4+2)/t/1e9:.0f} GB/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "mem"]
    if "gemm" in which: bench_gemms()
    if "attn" in which: bench_attn()
    if "mem" in which: bench_mem()
