"""Gate|up projection + GeGLU of the Gemma-2B MLP at B = 32 (17920 rows): one launch vs two, isolated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M, F, K = 17920, 16384, 2048
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
x, w = rnd(M, K), rnd(2 * F, K) * 0.05
t1 = min(timed(lambda: hip.linear_geglu_train(x, w)) for _ in range(2))
def two():
    gu = hip.linear_fwd(x, w)
    return hip.geglu_fwd(gu, pad=True)
t2 = min(timed(two) for _ in range(2))
tg = min(timed(lambda: hip.linear_fwd(x, w)) for _ in range(2))
print(f"fused {t1:7.1f} us | gate-up {tg:7.1f} + geglu_fwd {t2 - tg:7.1f} = {t2:7.1f} us")
