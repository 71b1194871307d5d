"""Where does a k-tile of the production GEMM kernel (tile 10) go?  Normal vs no in-loop LDS-DMA vs no MFMA, on the
LAP-3B shapes.  Needs a LAP_GEMM_EXPERIMENTAL=1 build (python -m lap_amd.build --force with the env set)."""
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


TILE = int(os.environ.get("TILE", "10"))
for name, kind, m, n, k in (("gateup fwd", "fwd", 17920, 32768, 2048), ("square 8192", "fwd", 8192, 8192, 8192),
                            ("gateup dgrad", "dgrad", 17920, 2048, 32768), ("gateup wgrad", "wgrad", 17920, 32768, 2048)):
    if kind == "fwd":
        a = rnd(m, k); w = rnd(n, k); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); fn = lambda: hip.linear_fwd(a, w, out, tile=TILE, ksplit=1)
    elif kind == "dgrad":
        a = rnd(m, k); w = rnd(k, n); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); fn = lambda: hip.linear_dgrad(a, w, out, tile=TILE, ksplit=1)
    else:
        dy = rnd(m, n); x = rnd(m, k); out = torch.empty(n, k, dtype=torch.float32, device=dev); fn = lambda: hip.linear_wgrad(dy, x, out, tile=TILE, ksplit=1)
    res = []
    TILE = int(os.environ.get("TILE", "10"))
    for bits, label in ((0, "normal"), (1, "no DMA"), (2, "no MFMA"), (3, "neither"), (4, "no setprio")):
        hip.call_noexcept = None
        hip._fn["lap_gemm_set_debug"](bits)
        t = timeit(fn)
        K = k if kind != "wgrad" else m
        tiles = ((m if kind != "wgrad" else n) + 255) // 256 * (((n if kind != "wgrad" else k) + 255) // 256)
        rounds = (tiles + 255) // 256
        res.append(f"{label}: {t*1e3:.3f} ms ({2*m*n*k/t/1e12:.0f} TF, {t*1e6/rounds/(K/64):.2f} us per k-tile)")
    hip._fn["lap_gemm_set_debug"](0)
    print(f"{name:14s} " + " | ".join(res), flush=True)
