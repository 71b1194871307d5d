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

# SigLIP: 64 images x 256 tokens, 16 heads of 72, no mask, fused qkv buffer
B, NH, HD, T = 64, 16, 72, 256
W = NH * HD
qkv = rnd(B * T, 3 * W)
q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
fl = 4 * B * NH * T * T * HD
for var in (0, 1, 1):
    hip.attention_set_variant(var)
    kw = dict(scale=HD ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0))
    t = timeit(lambda: hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD, **kw))
    (o, _), lse = hip.attention_fwd([q], [k], [v], [T], [T], B, NH, NH, HD, **kw)
    dqkv = torch.zeros_like(qkv)
    tb = timeit(lambda: hip.attention_bwd([q], [k], [v], [o], [o], lse, [T], [T], B, NH, NH, HD, dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]], **kw))
    print(f"siglip hd72 variant {var}: fwd {t*1e6:.0f} us {fl/t/1e12:.0f} TF | bwd {tb*1e6:.0f} us {2.5*fl/tb/1e12:.0f} TF", flush=True)
