"""The row-panel GEMM (csrc/serve_panel.hip) against the prefill's present launches, us per launch in a replayed graph of 20 calls,
weights rotating over 8 copies (HBM-cold as in the model, where every layer has its own)."""
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


scratch = hip._gemm_scratch(torch.device(dev))
for name, M, N, K, tile, norm, ks in (("sig_qkv", 512, 3456, 1152, 16, 2, 1), ("sig_out", 512, 1152, 1152, 17, 0, 1), ("sig_fc1", 512, 4352, 1152, 16, 2, 1),
                                      ("sig_fc2", 512, 1152, 4352, 6, 0, 4), ("gem_qkv", 560, 2560, 2048, 6, 1, 1), ("gem_out", 560, 2048, 2048, 16, 0, 1),
                                      ("gem_down", 560, 2048, 16384, 19, 0, 8)):
    a = rnd(M, K)
    ws = [rnd(N, K) * 0.05 for _ in range(8)]
    wps = [hip.serve_pack_weight(w, hip.PACK_PLAIN) for w in ws]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    i = [0]
    def old():
        i[0] = (i[0] + 1) % 8
        if ks > 1:
            hip.linear_partials(a, ws[i[0]], scratch, ksplit=(None if name == "sig_fc2" else ks), tile=tile)
        else:
            hip.linear_fwd(a, ws[i[0]], out, tile=tile, ksplit=1)
    def norm_old():
        if norm == 2: hip.layernorm_fwd(a, gam, bet)
        elif norm == 1: hip.rmsnorm_fwd(a, scale=bet, save_rstd=False)
    line = [f"{name:8s} {M}x{N}x{K}: present {timed(old):6.1f}" + (f" + norm {timed(norm_old):4.1f}" if norm else "")]
    for nt in (1, 2, 3, 4):
        def new():
            i[0] = (i[0] + 1) % 8
            if ks > 1:
                hip.panel_partials(a, wps[i[0]], N, scratch, ks, nt=nt)
            else:
                hip.panel_linear(a, wps[i[0]], N, norm=norm, gamma=gam if norm else None, beta=bet if norm == 2 else None, out=out, nt=nt)
        try:
            line.append(f"nt{nt} {timed(new):6.1f}")
        except Exception as e:
            line.append(f"nt{nt} fail")
    print("  ".join(line) + "  us", flush=True)
