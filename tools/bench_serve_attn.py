"""Serving attention (batch 1, 50 suffix queries over 816 cached + 50 fresh keys): latency vs key-split count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
def rnd(*s): return (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
B, NH, HD, Pn, S = 1, 8, 256, 816, 50
q = rnd(B, S, NH * HD); ck = rnd(B, Pn, HD); cv = rnd(B, Pn, HD); k = rnd(B, S, HD); v = rnd(B, S, HD)
def run(ns):
    return hip.attention_fwd([None, q], [ck, k], [cv, v], [0, S], [Pn, S], B, NH, 1, HD, need_lse=False, nsplit_hint=ns)
for ns in (None, 1, 2, 3, 4, 7, 14):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): run(ns)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(50): run(ns)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"nsplit {ns}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per attention (graph of 50)")
