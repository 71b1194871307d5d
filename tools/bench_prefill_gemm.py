"""Serving-prefill GEMMs (M = 512 SigLIP rows / 560 Gemma rows, forward layout): the library's automatic choice (split-K + reduce
where it picks one) against explicit tile candidates, timed as hipGraph replays of 20 back-to-back calls (what the sampler graph sees).
usage: bench_prefill_gemm.py [shape-name ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip

dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
SHAPES = {   # name: (M, N, K, bias, gelu, residual)
    "sig_qkv": (512, 3456, 1152, True, False, False), "sig_out": (512, 1152, 1152, True, False, True),
    "sig_fc1": (512, 4304, 1152, True, True, False), "sig_fc2": (512, 1152, 4304, True, False, True),
    "sig_head": (512, 2048, 1152, True, False, False),
    "gem_qkv": (560, 2560, 2048, False, False, False), "gem_out": (560, 2048, 2048, False, False, True),
    "gem_gu": (560, 32768, 2048, False, False, False), "gem_down": (560, 2048, 16384, False, False, True),
}
CANDS = {"sig_qkv": [-1, 16, 18, 6], "sig_out": [-1, 17, 16, 18], "sig_fc1": [-1, 16, 18, 6], "sig_fc2": [-1, 17, 16], "sig_head": [-1, 16, 17, 18],
         "gem_qkv": [-1, 16, 18, 6], "gem_out": [-1, 16, 17, 18], "gem_gu": [-1, 15, 10, 6], "gem_down": [-1, 15, 16]}
# (round 4: the same tiles with 5 - 8 LDS stages instead of 3 - 4 were measured too — 12.1 vs 12.4 us, 17.2 vs 17.2 us ...: these
#  launches are not bound by k-tiles in flight; profiles/r04_prefill_gemm_ring_depth.txt)


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


for name in (sys.argv[1:] or SHAPES):
    M, N, K, hb, gelu, hr = SHAPES[name]
    a, w = rnd(M, K), rnd(N, K)
    bias = torch.randn(N, device=dev) if hb else None
    res = rnd(M, N) if hr else None
    ref = None
    out_line = [f"{name:9s} {M}x{N}x{K}"]
    for tile in CANDS[name]:
        for ks in ((0,) if tile == -1 else (1, 2, 4) if K >= 2048 and tile != 15 else (1, 2) if tile != 15 else (1,)):
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            # the fused epilogue with GELU only exists unsplit; split runs need scratch (hip.gemm lends it when ksplit == 0)
            def fn():
                if ks > 1:
                    sc = hip._gemm_scratch(a.device)
                    hip.call("lap_gemm_bf16_ex", hip._p(a), hip._p(w), hip._p(out), hip._p(bias), hip._p(res), M, N, K, K, K, N, N if hr else 0, 1.0, 1, 1,
                             (hip.GEMM_GELU if gelu else 0) | (hip.GEMM_BIAS_F32 if hb else 0), tile, ks, hip._p(sc), sc.numel() * 4)
                else:
                    hip.linear_fwd(a, w, out, bias=bias, residual=res, gelu=gelu, tile=tile, ksplit=ks)
            try:
                t = timed(fn)
            except Exception as e:   # noqa: BLE001
                out_line.append(f"t{tile}/k{ks}: {type(e).__name__}")
                continue
            if ref is None:
                ref = out.clone()
                err = 0.0
            else:
                err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
            out_line.append(f"t{tile}/k{ks}: {t:6.1f}us{'' if err < 2e-3 else f' ERR {err:.1e}'}")
    print("  ".join(out_line), flush=True)
