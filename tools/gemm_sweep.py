"""Isolated sweep of (tile, ksplit) choices over the train step's weak GEMM shapes (tools/gemm_census.py finds them).
Usage: python tools/gemm_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()

def run(M, N, K, al, bl, od, tile, ksplit, bias=False, res=False):
    a = rnd(M, K) if al == "k" else rnd(K, M)
    b = rnd(N, K) if bl == "k" else rnd(K, N)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if od == "f32" else torch.bfloat16)
    kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=al == "k", b_kc=bl == "k", tile=tile, ksplit=ksplit)
    if ksplit > 1:   # explicit split: two-phase needs scratch, which hip.gemm only lends on ksplit == 0 -> call the C ABI directly
        sc = hip._gemm_scratch(dev)
        flags = hip.GEMM_OUT_F32 if od == "f32" else 0
        fn = lambda: hip.call("lap_gemm_bf16_ex", hip._p(a), hip._p(b), hip._p(out), None, None, M, N, K, kw["lda"], kw["ldb"], N, 0, 1.0,
                              int(kw["a_kc"]), int(kw["b_kc"]), flags, tile, ksplit, hip._p(sc), sc.numel() * 4)
    else:
        if bias: kw["bias"] = torch.zeros(N, device=dev, dtype=torch.float32)
        if res: kw["residual"] = rnd(M, N); kw["ldr"] = N
        fn = lambda: hip.gemm(a, b, out, **kw)
    try:
        for _ in range(3): fn()
    except Exception as e:   # noqa: BLE001
        return None
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 * 1e3

SHAPES = [  # M, N, K, A, B, out
    (16384, 4304, 1152, "k", "k", "bf16"), (16384, 1152, 4304, "k", "k", "bf16"), (16384, 3456, 1152, "k", "k", "bf16"),
    (16384, 1152, 1152, "k", "k", "bf16"), (16384, 1152, 4304, "k", "n", "bf16"), (16384, 4304, 1152, "k", "n", "bf16"),
    (16384, 1152, 3456, "k", "n", "bf16"), (16384, 1152, 1152, "k", "n", "bf16"),
    (3456, 1152, 16384, "m", "n", "f32"), (1152, 1152, 16384, "m", "n", "f32"), (1152, 4304, 16384, "m", "n", "f32"), (4304, 1152, 16384, "m", "n", "f32"),
    (17920, 2048, 2048, "k", "k", "bf16"), (17920, 2048, 2048, "k", "n", "bf16"), (17920, 2560, 2048, "k", "k", "bf16"), (17920, 2048, 2560, "k", "n", "bf16"),
    (1600, 1024, 8192, "k", "n", "bf16"), (1600, 1024, 4096, "k", "k", "bf16"), (1600, 1024, 2048, "k", "k", "bf16"), (1600, 1024, 2560, "k", "n", "bf16"),
    (1600, 8192, 1024, "k", "k", "bf16"), (1600, 2560, 1024, "k", "k", "bf16"), (1600, 4096, 1024, "k", "n", "bf16"), (1600, 2048, 1024, "k", "n", "bf16"),
    (1024, 4096, 1600, "m", "n", "f32"), (2560, 1024, 1600, "m", "n", "f32"), (1024, 2048, 1600, "m", "n", "f32"), (8192, 1024, 1600, "m", "n", "f32"),
]
CH = [(-1, 0), (5, 1), (6, 1), (5, 2), (6, 2), (6, 4), (5, 4)]
print(f"{'M':>6} {'N':>6} {'K':>6} A B  out | " + " ".join(f"t{t}/s{k}".rjust(9) for t, k in CH) + "   (us; TF/s of the best)")
for M, N, K, al, bl, od in SHAPES:
    ts = [run(M, N, K, al, bl, od, t, k) for t, k in CH]
    best = min(x for x in ts if x)
    print(f"{M:>6} {N:>6} {K:>6} {al} {bl} {od:>4} | " + " ".join(("%9.1f" % x) if x else "        -" for x in ts) + f"   best {2.0*M*N*K/best/1e6:6.0f} auto {2.0*M*N*K/ts[0]/1e6:6.0f}")
