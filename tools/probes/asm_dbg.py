import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
M, N, K = 256, 512, 128
def run(a, b):
    out = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
    hip.call("lap_gemm_nt_asm", hip._p(a), hip._p(b), hip._p(out), M, N, K, K, K, N, hip._stream())
    torch.cuda.synchronize()
    return out.float()
ones = lambda r: torch.ones(r, K, device=dev, dtype=torch.bfloat16)
# (a) column mapping: B[n, :] = n % 16 + 1 (k-uniform), A = ones / K
a = ones(M); b = ((torch.arange(N, device=dev) % 16 + 1).float()[:, None].expand(N, K)).bfloat16().contiguous()
o = run(a, b) / K
exp = (torch.arange(N, device=dev) % 16 + 1).float()[None, :].expand(M, N)
ok = (o == exp)
print("col test: fraction ok", ok.float().mean().item())
blk = ok.view(16, 16, 32, 16).permute(0, 2, 1, 3).reshape(16, 32, 256).all(-1)
for r in range(16): print("".join("#" if x else "." for x in blk[r].tolist()))
print("row 0, cols 56..80:", o[0, 56:80].tolist())
print("row 70, cols 0..20:", o[70, 0:20].tolist())
# (b) row mapping
a = ((torch.arange(M, device=dev) % 16 + 1).float()[:, None].expand(M, K)).bfloat16().contiguous(); b = ones(N)
o = run(a, b) / K
exp = (torch.arange(M, device=dev) % 16 + 1).float()[:, None].expand(M, N)
ok = (o == exp)
print("row test: fraction ok", ok.float().mean().item())
blk = ok.view(16, 16, 32, 16).permute(0, 2, 1, 3).reshape(16, 32, 256).all(-1)
for r in range(16): print("".join("#" if x else "." for x in blk[r].tolist()))
# (c) k mapping: A[m, k] = 1 for all; B[n, k] = (k == n % 128)
a = ones(M); b = torch.zeros(N, K, device=dev); b[torch.arange(N), torch.arange(N) % K] = 1; b = b.bfloat16()
o = run(a, b)
print("k test: fraction == 1:", (o == 1).float().mean().item(), " values:", torch.unique(o)[:10].tolist())
