"""LayerNorm / RMSNorm backward at the train step's shapes: time and effective bandwidth (x, dy, dx read + dx written)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from lap_amd import hip
dev = "cuda:0"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
rows, W = 16384, 1152
x, dy, d = rnd(rows, W), rnd(rows, W), rnd(rows, W)
gam = torch.randn(W, device=dev); mean = torch.randn(rows, device=dev) * 0.01; rstd = torch.randn(rows, device=dev).abs() + 0.5
dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev); sums = torch.zeros(W, device=dev)
us = t(lambda: hip.layernorm_bwd(x, dy, gam, mean, rstd, dg, db, dx=d, accum_dx=True, dxsum=sums))
print("layernorm_bwd 16384 x 1152 accum + dxsum  %.1f us  %.2f TB/s" % (us, 4 * rows * W * 2 / us / 1e6))
rows, W = 17920, 2048
x, dy, d = rnd(rows, W), rnd(rows, W), rnd(rows, W)
sc = torch.randn(W, device=dev); rstd = torch.randn(rows, device=dev).abs() + 0.5; dsc = torch.zeros(W, device=dev)
us = t(lambda: hip.rmsnorm_bwd(x, dy, rstd, scale=sc, dx=d, dscale=dsc, accum_dx=True))
print("rmsnorm_bwd 17920 x 2048 accum            %.1f us  %.2f TB/s" % (us, 4 * rows * W * 2 / us / 1e6))
us = t(lambda: hip.rmsnorm_fwd(x, scale=sc, save_rstd=True))
print("rmsnorm_fwd 17920 x 2048                  %.1f us  %.2f TB/s" % (us, 2 * rows * W * 2 / us / 1e6))
rows, W = 16384, 1152
x = rnd(rows, W); gam = torch.randn(W, device=dev); bet = torch.randn(W, device=dev)
us = t(lambda: hip.layernorm_fwd(x, gam, bet))
print("layernorm_fwd 16384 x 1152                %.1f us  %.2f TB/s" % (us, 2 * rows * W * 2 / us / 1e6))
