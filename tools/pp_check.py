"""Race screen for the ping-pong GEMM (tile 8): bit-equality with tile 2 (the 8-wave single-phase kernel) (same f32 accumulation order) on large shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
def rnd(*s): return (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
bad = 0
for rep in range(3):
    for (m, n, k) in [(4096, 4096, 4096), (17920, 2048, 2048), (2000, 3000, 1096), (1504, 2048, 16384), (520, 392, 200), (256, 256, 64), (256, 256, 128), (304, 264, 192)]:
        a = rnd(m, k); w = rnd(n, k)
        e = torch.equal(hip.linear_fwd(a, w, tile=8, ksplit=1), hip.linear_fwd(a, w, tile=2, ksplit=1)); bad += not e
        w2 = rnd(k, n)
        e2 = torch.equal(hip.linear_dgrad(a, w2, tile=8, ksplit=1), hip.linear_dgrad(a, w2, tile=2, ksplit=1)); bad += not e2
        dy = rnd(k, m); x = rnd(k, n)   # wgrad: contraction over k rows
        g8 = torch.empty(m, n, device=dev); g5 = torch.empty(m, n, device=dev)
        hip.linear_wgrad(dy, x, g8, tile=8, ksplit=1); hip.linear_wgrad(dy, x, g5, tile=2, ksplit=1)
        e3 = torch.equal(g8, g5); bad += not e3
        print(rep, (m, n, k), e, e2, e3, flush=True)
print("BAD" if bad else "ALL_EQUAL", bad)
