"""Forward attention, HD = 256, MQA: two query heads per wave (attn_dma_qg_kernel) against one (attn_dma_q_kernel<256, 0>) at the train
step's shape (B = 32, 8 heads, 1 kv head, 560 prefix + 50 suffix tokens, the LAP mask): bitwise comparison and us per launch.
Needs a library with the parked kernel built in (tools/probes/attention_qg.hpp says how): lap_attention_set_variant 3 / 4 select the two.
Result: profiles/r06_attention_two_heads_per_wave_ab.txt (bitwise equal, 209 vs 172 - 182 us: slower)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip

dev = "cuda"
B, NH, NKV, HD, Tp, S = 32, 8, 1, 256, 560, 50
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).bfloat16().to(dev)
q0, q1 = rnd(B * Tp, NH * HD, sc=HD ** -0.25), rnd(B * S, NH * HD, sc=HD ** -0.25)
k0, k1 = rnd(B * Tp, HD, sc=HD ** -0.25), rnd(B * S, HD, sc=HD ** -0.25)
v0, v1 = rnd(B * Tp, HD), rnd(B * S, HD)
qinfo = torch.zeros(B, Tp + S, dtype=torch.int32); kinfo = torch.zeros(B, Tp + S, dtype=torch.int32)
n_lang, n_pad = 40, 6
for b in range(B):
    npad = (n_pad + b) % (n_pad + 1)
    nq = Tp - n_lang - npad
    for t in range(Tp):
        if t < nq:
            qinfo[b, t] = (3 << 24); kinfo[b, t] = (1 << 24)
        elif t < nq + n_lang:
            kk = t - nq + 1
            qinfo[b, t] = (3 << 24) | kk; kinfo[b, t] = (2 << 24) | kk
    qinfo[b, Tp:] = (5 << 24) | 0xFFFFFF; kinfo[b, Tp:] = (4 << 24)
qinfo, kinfo = qinfo.to(dev), kinfo.to(dev)


def run():
    return hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, NKV, HD, qinfo, kinfo)


def timed(n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
for name, var in (("one head per wave", 3), ("two heads per wave", 4)):
    hip.attention_set_variant(var)
    (o0, o1), lse = run()
    res[name] = (o0.clone(), o1.clone(), lse.clone())
    ts = [timed() for _ in range(3)]
    print(f"{name}: {' / '.join(f'{t:.1f}' for t in ts)} us per launch", flush=True)
hip.attention_set_variant(-1)
a, b = res["one head per wave"], res["two heads per wave"]
print("bitwise equal:", all(torch.equal(x, y) for x, y in zip(a, b)))
