"""Attention micro-benchmark at the lap_bench shapes, per kernel variant (tools only)."""
import sys
import torch
from lap_amd import hip

dev = "cuda"
def rnd(*s): return (torch.randn(*s, device=dev) * 0.3).bfloat16()
def timeit(fn, iters=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

sys.path.insert(0, "tests")
from test_kernels_gpu import _lap_infos
B, NH, HD, Tp, S = 32, 8, 256, 560, 50
q0 = rnd(B, Tp, NH * HD); q1 = rnd(B, S, NH * HD); k0 = rnd(B, Tp, HD); k1 = rnd(B, S, HD); v0 = rnd(B, Tp, HD); v1 = rnd(B, S, HD)
qi, ki = _lap_infos(B, Tp, S, 48, 5, dev)
fl = 4 * B * NH * (Tp + S) ** 2 * HD
for var in (0, 1, 1):
    hip.attention_set_variant(var)
    for masked in (False, True):
        a = (qi, ki) if masked else (None, None)
        t = timeit(lambda: hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD, *a))
        (o0, o1), lse = hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD, *a)
        tb = timeit(lambda: hip.attention_bwd([q0, q1], [k0, k1], [v0, v1], [o0, o1], [q0, q1], lse, [Tp, S], [Tp, S], B, NH, 1, HD, *a))
        print(f"variant {var} masked={masked}: fwd {t*1e6:.0f} us {fl/t/1e12:.0f} TF | bwd {tb*1e6:.0f} us {2.5*fl/tb/1e12:.0f} TF", flush=True)
