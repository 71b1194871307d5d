"""Run the lap_bench attention fwd (+bwd) a few times for one kernel variant (profiling driver)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from lap_amd import hip
from test_kernels_gpu import _lap_infos
var = int(sys.argv[1]); bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.3).bfloat16()
B, NH, HD, Tp, S = 32, 8, 256, 560, 50
q0 = rnd(B, Tp, NH * HD); q1 = rnd(B, S, NH * HD); k0 = rnd(B, Tp, HD); k1 = rnd(B, S, HD); v0 = rnd(B, Tp, HD); v1 = rnd(B, S, HD)
qi, ki = _lap_infos(B, Tp, S, 48, 5, dev)
hip.attention_set_variant(var)
for _ in range(5):
    (o0, o1), lse = hip.attention_fwd([q0, q1], [k0, k1], [v0, v1], [Tp, S], [Tp, S], B, NH, 1, HD, qi, ki)
    if bwd:
        hip.attention_bwd([q0, q1], [k0, k1], [v0, v1], [o0, o1], [q0, q1], lse, [Tp, S], [Tp, S], B, NH, 1, HD, qi, ki)
torch.cuda.synchronize()
