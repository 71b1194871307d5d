"""One process that launches the production gate|up + GeGLU kernel (lap_gemm_asm_nt_geglu) at the benchmark shape a few times:
the target of tools/pmc_traffic_geglu.sh (rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M, F, K = 17920, 16384, 2048
x, w = rnd(M, K), rnd(2 * F, K) * 0.05
for _ in range(5):
    hip.linear_geglu_train(x, w)
torch.cuda.synchronize()
