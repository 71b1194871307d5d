import os, sys
sys.path.insert(0, "/root/repo")
import torch
from lap_amd import hip
dev="cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
V, D, R = 257152, 2048, 1024
dl, pl = rnd(R, V), rnd(R, D)
g = torch.empty(V, D, device=dev)
t = timed(lambda: hip.linear_wgrad(dl, pl, g))
ref = torch.empty(V, D, device=dev)
os.environ["X"]="1"
hip.gemm(dl, pl, ref, M=V, N=D, K=R, lda=V, ldb=D, ldc=D, a_kc=False, b_kc=False, tile=12, ksplit=1)
print(f"table wgrad {t:.1f} us  {2.0*V*D*R/t/1e6:.0f} TF/s  equal to the HIP tile: {torch.equal(g, ref)}  rel {((g-ref).norm()/ref.norm()).item():.2e}")
