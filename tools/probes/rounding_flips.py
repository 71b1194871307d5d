"""How far apart are two bf16 GEMM outputs that differ only in their f32 summation order?  (where the
prefix stream's teacher-forced distance comes from.)  engine tile order vs torch's f32 matmul, both rounded to bf16 once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
torch.manual_seed(0)
M = 1120
for K, N in ((1024, 1024), (2048, 2048), (4096, 1024), (16384, 2048)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    e = hip.linear_fwd(a, w).float()
    r32 = a.float() @ w.float().t()
    t = r32.bfloat16().float()
    flips = (e != t).float().mean().item()
    print(f"K={K:6d}: elements that round differently {flips:.3%}; relative L2 between the two bf16 outputs {((e - t).norm() / t.norm()).item():.2e}; "
          f"either one vs the f32 product {((t - r32).norm() / r32.norm()).item():.2e}")
