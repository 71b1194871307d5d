"""layernorm_bwd against background GEMMs by layout and tile: which kernel family disturbs it?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
dev = "cuda:0"
rows, W, MLP = 1536, 1152, 4304
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
rndf = lambda *s: torch.randn(*s, device=dev, generator=g)
dx, y2, y, dh = rnd(rows, W), rnd(rows, W), rnd(rows, W), rnd(rows, MLP)
gam, mean, rstd = rndf(W), rndf(rows) * 0.01, rndf(rows).abs() + 0.5
w1 = rnd(MLP, W)
outW = torch.empty(MLP, W, device=dev); outWb = torch.empty(MLP, W, device=dev, dtype=torch.bfloat16)
outF = torch.empty(rows, MLP, device=dev, dtype=torch.bfloat16); outD = torch.empty(rows, W, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
def ln_bwd():
    d = torch.empty_like(dx); dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev)
    hip.layernorm_bwd(y2, y, gam, mean, rstd, dg, db, dx=d, accum_dx=False)
    return d
def tn(tile, ksplit, f32=True):
    o = outW if f32 else outWb
    return lambda: hip.gemm(dh, y2, o, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=tile, ksplit=ksplit)
loads = {"TN auto": lambda: hip.linear_wgrad(dh, y2, outW)}
for t in (6, 5, 10, 12, 2):
    loads[f"TN tile {t} ksplit 1 f32"] = tn(t, 1)
loads["TN tile 6 ksplit 1 bf16 out"] = tn(6, 1, False)
loads["NT tile 6 (fwd fc1)"] = lambda: hip.linear_fwd(y2, w1, out=outF, tile=6)
loads["NN tile 6 (dgrad w1)"] = lambda: hip.linear_dgrad(dh, w1, out=outD, tile=6)
loads["NN tile 10"] = lambda: hip.linear_dgrad(dh, w1, out=outD, tile=10)
ref = ln_bwd().clone(); torch.cuda.synchronize()
for name, bg in loads.items():
    try:
        bg(); torch.cuda.synchronize()
    except Exception as e:
        print(f"{name:32s} rejected ({e})"); continue
    bad = 0
    for rep in range(40):
        with torch.cuda.stream(side):
            for _ in range(6):
                bg()
        r = ln_bwd(); torch.cuda.synchronize()
        bad += not torch.equal(r, ref)
    print(f"{name:32s} mismatches {bad}/40", flush=True)
