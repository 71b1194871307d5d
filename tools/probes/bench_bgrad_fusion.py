"""LayerNorm backward with / without the fused column sums of dx, against the column-sum kernel it replaces (and the GELU
backward + column-sum pair, whose fused form — a strip kernel, 199 us against 85 + 24 — was dropped)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from lap_amd import hip
dev="cuda:0"; rows, W, MLP = 16384, 1152, 4304
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
x, dy, h, dh = rnd(rows, W), rnd(rows, W), rnd(rows, MLP), rnd(rows, MLP)
gam = torch.randn(W, device=dev); mean = torch.randn(rows, device=dev) * 0.01; rstd = torch.randn(rows, device=dev).abs() + 0.5
dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev); sums = torch.zeros(W, device=dev); bsum = torch.zeros(MLP, device=dev)
d = rnd(rows, W)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
print("ln_bwd accum           %.1f us" % t(lambda: hip.layernorm_bwd(x, dy, gam, mean, rstd, dg, db, dx=d, accum_dx=True)))
print("ln_bwd accum + dxsum   %.1f us" % t(lambda: hip.layernorm_bwd(x, dy, gam, mean, rstd, dg, db, dx=d, accum_dx=True, dxsum=sums)))
print("colsum [16384,1152]    %.1f us" % t(lambda: hip.colsum(d, sums)))
print("gelu_bwd               %.1f us" % t(lambda: hip.gelu_bwd(h, dh)))
print("colsum [16384,4304]    %.1f us" % t(lambda: hip.colsum(dh, bsum)))
