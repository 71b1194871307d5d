"""Calibration probe: which hipBLASLt kernels torch.matmul picks for the NT shapes (kernel names encode tile / wave layout)."""
import torch
dev = "cuda"
for (m, n, k) in [(17920, 32768, 2048), (8192, 8192, 8192), (17920, 2048, 2048)]:
    a = (torch.rand(m, k, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(n, k, device=dev) * 2 - 1).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())
torch.cuda.synchronize()
