"""What a big GEMM costs when the optimizer kernel runs beside it: time of N back-to-back GEMM launches alone, and with an AdamW + EMA pass
over a layer-sized unit (110 M parameters) looping on a second stream for the whole time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()
n = 110_000_000
p, m, v, ema, g = (torch.zeros(n, device=dev) for _ in range(5))
p16 = torch.zeros(n, device=dev, dtype=torch.bfloat16)
scal = torch.ones(8, device=dev)
side = torch.cuda.Stream()
def opt_loop(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            hip.adamw_ema(p, m, v, ema, g, p16, scal, 0.9, 0.95, 1e-8, 1e-4, 1.0)
def run(fn, reps, with_opt):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    if with_opt: opt_loop(with_opt)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
# one optimizer pass alone
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
opt_loop(2); torch.cuda.synchronize(); s.record(side); opt_loop(10); e.record(side); torch.cuda.synchronize()
t_opt = s.elapsed_time(e) / 10 * 1e3
print(f"adamw+ema over {n/1e6:.0f} M parameters alone: {t_opt:.0f} us ({n*38/t_opt/1e6:.2f} TB/s)", flush=True)
cases = [("nt gate-up fwd", True, True, 17920, 32768, 2048, torch.bfloat16, 0), ("nn down dgrad", True, False, 17920, 16384, 2048, torch.bfloat16, 0),
         ("tn gate-up wgrad", False, False, 32768, 2048, 17920, torch.float32, 64), ("tn down wgrad", False, False, 2048, 16384, 17920, torch.float32, 64)]
for name, a_kc, b_kc, M, N, K, dt, pad in cases:
    a = rnd(M, K) if a_kc else rnd(K, M + pad)[:, :M]
    b = rnd(N, K) if b_kc else rnd(K, N)
    out = torch.empty(M, N, device=dev, dtype=dt)
    kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=a_kc, b_kc=b_kc)
    fn = lambda: hip.gemm(a, b, out, **kw)
    reps = 12
    alone = min(run(fn, reps, 0) for _ in range(2))
    k = int(alone * reps * 1.6 / t_opt) + 2           # enough optimizer passes to cover the GEMM loop
    both = min(run(fn, reps, k) for _ in range(2))
    print(f"{name:18s} alone {alone:7.1f} us ({2.0*M*N*K/alone/1e6:5.0f} TF)   beside the optimizer {both:7.1f} us ({both/alone:.2f} x)", flush=True)
# the other side: optimizer passes per unit time while GEMMs run back to back
name, a_kc, b_kc, M, N, K, dt, pad = cases[0]
a, b = rnd(M, K), rnd(N, K); out = torch.empty(M, N, device=dev, dtype=dt)
kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_kc=True, b_kc=True)
torch.cuda.synchronize()
for _ in range(40): hip.gemm(a, b, out, **kw)          # ~80 ms of GEMMs on the compute stream
s.record(side); opt_loop(30); e.record(side); torch.cuda.synchronize()
x = s.elapsed_time(e) / 30 * 1e3
print(f"optimizer pass beside back-to-back gate-up forwards: {x:.0f} us ({x/t_opt:.2f} x);  serial-equivalent of the overlapped work: {1639.6/2486.8 + t_opt/x:.2f}")
