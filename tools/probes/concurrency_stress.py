"""Does a kernel give the same bits when another stream keeps the chip busy?  For each candidate: solo result, then the same
launch while a background stream runs weight-gradient GEMMs (what the off-path gradient stream / a collective would do).

python tools/probes/concurrency_stress.py [rows]
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
dev = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
W, MLP, T, NH = 1152, 4304, 256, 16
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
rndf = lambda *s: torch.randn(*s, device=dev, generator=g)
dx, a, h, y2, o, y = rnd(rows, W), rnd(rows, MLP), rnd(rows, MLP), rnd(rows, W), rnd(rows, W), rnd(rows, W)
w2, w1, wo, wqkv = rnd(W, MLP), rnd(MLP, W), rnd(W, W), rnd(3 * W, W)
dh, dqkv = rnd(rows, MLP), rnd(rows, 3 * W)
gam, mean, rstd = rndf(W), rndf(rows) * 0.01, rndf(rows).abs() + 0.5
bg_out = [torch.empty(W, MLP, device=dev), torch.empty(MLP, W, device=dev), torch.empty(W, W, device=dev), torch.empty(3 * W, W, device=dev)]
side = torch.cuda.Stream()

def background():
    with torch.cuda.stream(side):
        for _ in range(3):
            hip.linear_wgrad(dx, a, bg_out[0]); hip.linear_wgrad(dh, y2, bg_out[1]); hip.linear_wgrad(dx, o, bg_out[2]); hip.linear_wgrad(dqkv, y, bg_out[3])

def ln_bwd():
    d = dx.clone(); dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev)
    hip.layernorm_bwd(y2, y, gam, mean, rstd, dg, db, dx=d, accum_dx=True)
    return d
def colsum():
    out = torch.zeros(MLP, device=dev); hip.colsum(dh, out); return out
cands = {
    "dgrad w2 [rows,W]x[W,MLP]": lambda: hip.linear_dgrad(dx, w2),
    "dgrad w1 [rows,MLP]x[MLP,W]": lambda: hip.linear_dgrad(dh, w1),
    "dgrad wo": lambda: hip.linear_dgrad(dx, wo),
    "dgrad wqkv": lambda: hip.linear_dgrad(dqkv, wqkv),
    "fwd fc1 + bias": lambda: hip.linear_fwd(y2, w1, bias=gam.new_zeros(MLP)),
    "fwd fc2 + res": lambda: hip.linear_fwd(a, w2, residual=dx),
    "gelu_bwd": lambda: hip.gelu_bwd(h, dh),
    "layernorm_bwd accum": ln_bwd,
    "colsum (atomics? f32)": colsum,
    "wgrad w1 (main)": lambda: hip.linear_wgrad(dh, y2, torch.empty(MLP, W, device=dev)),
}
for name, f in cands.items():
    ref = f().clone(); torch.cuda.synchronize()
    again = f().clone(); torch.cuda.synchronize()
    solo_same = torch.equal(ref, again)
    bad = 0; worst = 0.0
    for rep in range(20):
        background()
        out = f()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, ((out.float() - ref.float()).norm() / ref.float().norm()).item())
    print(f"{name:32s} solo-repeatable {solo_same}  mismatches under load {bad}/20  worst rel {worst:.1e}", flush=True)
