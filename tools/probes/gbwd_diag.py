import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = "cuda"
def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*s, generator=g) * 2 - 1).bfloat16().to(dev)
M, N, K = 256, 512, 512
dy, w = rnd(M, K, seed=1), rnd(K, N, seed=2)
gu = (rnd(M, 2 * N, seed=3) * 6.0)
dgu = hip.linear_dgrad_geglu_bwd(dy, w, gu)
dact = hip.linear_dgrad(dy, w, tile=12, ksplit=1)
ref = hip.geglu_bwd(gu, dact)
g, u = gu[:, :N].float(), gu[:, N:].float()
da = dact.float()
k0, k1 = 0.7978845608028654, 0.044715
t = torch.tanh(k0 * (g + k1 * g ** 3))
gelu = 0.5 * g * (1 + t)
gp = 0.5 * (1 + t) + 0.5 * g * (1 - t * t) * k0 * (1 + 3 * k1 * g * g)
want = torch.cat([da * u * gp, da * gelu.bfloat16().float()], 1)
for name, a in (("fused", dgu), ("two-launch", ref)):
    for hname, sl in (("d(gate)", slice(0, N)), ("d(up)", slice(N, 2 * N))):
        x, y = a[:, sl].float(), want[:, sl]
        wb = y.bfloat16().float()
        print(f"{name:10s} {hname:8s}: equal to bf16(f32 formula) {(x == wb).float().mean().item():.4f}  max rel diff {((x - y).abs() / (y.abs() + 1e-20)).max().item():.3e}  rel_err {((x - y).norm() / y.norm()).item():.3e}")
eq = (dgu == ref)
print("fused == two-launch: gate", eq[:, :N].float().mean().item(), "up", eq[:, N:].float().mean().item())
bad = (~eq).nonzero()[:5]
for r, c in bad.tolist():
    print(r, c, dgu[r, c].item(), ref[r, c].item(), want[r, c].item(), "gate", gu[r, c % N].item(), "up", gu[r, N + c % N].item(), "dact", dact[r, c % N].item())
