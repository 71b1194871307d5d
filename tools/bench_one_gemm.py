import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
kind, m, n, k, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
if kind == "fwd":
    a = rnd(m, k); w = rnd(n, k); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); fn = lambda: hip.linear_fwd(a, w, out, tile=tile)
elif kind == "dgrad":
    a = rnd(m, k); w = rnd(k, n); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev); fn = lambda: hip.linear_dgrad(a, w, out, tile=tile)
else:
    dy = rnd(m, n); x = rnd(m, k); out = torch.empty(n, k, dtype=torch.float32, device=dev); fn = lambda: hip.linear_wgrad(dy, x, out, tile=tile)
for _ in range(5): fn()
torch.cuda.synchronize()
