"""A/B of the HD = 256 attention backward at the benchmark shapes (LAP block mask): delta = rowsum(dO o O) inside the dQ launch
(variant 1, production) against the separate delta pass in front (variant 2); outputs compared, forward / backward timed with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lap_amd import hip
from test_kernels_gpu import _lap_infos

dev = "cuda"
def rnd(*s): return (torch.randn(*s, device=dev) * 0.3).bfloat16()
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

B, NH, HD, Tp, S = 32, 8, 256, 560, 50
torch.manual_seed(0)
q0 = rnd(B, Tp, NH * HD); q1 = rnd(B, S, NH * HD); k0 = rnd(B, Tp, HD); k1 = rnd(B, S, HD); v0 = rnd(B, Tp, HD); v1 = rnd(B, S, HD)
d0 = rnd(B, Tp, NH * HD); d1 = rnd(B, S, NH * HD)
qi, ki = _lap_infos(B, Tp, S, 48, 5, dev)
fl = 4 * B * NH * (Tp + S) ** 2 * HD
ref = {}
for rep in range(2):
    for var in (2, 1):
        hip.attention_set_variant(var)
        fwd = lambda: hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD, qi, ki)
        (o0, o1), lse = fwd()
        bwd = lambda: hip.attention_bwd([q0, q1], [k0, k1], [v0, v1], [o0, o1], [d0, d1], lse, [Tp, S], [Tp, S], B, NH, 1, HD, qi, ki)
        g = bwd()
        flat = [o0, o1, lse] + [t for grp in g for t in grp]
        if var == 2: ref = [t.clone() for t in flat]
        else:
            same = [bool(torch.equal(a, b)) for a, b in zip(ref, flat)]
            worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ref, flat))
            print(f"variant {var} (delta inside the dQ launch) vs 2 (separate delta pass): bit-equal per tensor {same}, worst abs diff {worst:.3e}", flush=True)
        tf = timeit(fwd); tb = timeit(bwd)
        if rep == 1 and var == 3:
            for abl in ():
                hip.attention_set_variant(abl)
                print(f"ablation variant {abl} (3: no DMA wait, 4: no DMA wait, no barrier; wrong results): fwd {timeit(fwd):.1f} us", flush=True)
        print(f"variant {var}: fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF) | bwd (delta + dkdv + reduce + dq) {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF)", flush=True)
