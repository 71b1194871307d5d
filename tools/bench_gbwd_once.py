"""One process that launches the fused down-projection data gradient + GeGLU backward (lap_gemm_asm_nn_geglu_bwd) and, for comparison, the plain data
gradient of the same shape (lap_gemm_asm_nn) a few times: the target of tools/pmc_traffic_gbwd.sh (rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
M, N, K = 17920, 16384, 2048
dy, w = rnd(M, K), rnd(K, N)
gu = (rnd(M, 2 * N + 64) * 4)[:, :2 * N]
for _ in range(5):
    hip.linear_dgrad_geglu_bwd(dy, w, gu)
for _ in range(5):
    hip.linear_dgrad(dy, w)
torch.cuda.synchronize()
