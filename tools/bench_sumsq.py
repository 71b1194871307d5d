import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
for n in (133_000_000, 528_000_000, 12_345_677):
    x = torch.randn(n, device="cuda"); out = torch.zeros(1, device="cuda")
    for _ in range(3): hip.sumsq_f32(x, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out.zero_(); s.record()
    for _ in range(10): hip.sumsq_f32(x, out)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    ref = (x.double() ** 2).sum().item() * 10
    print(f"n={n}: {n*4/t/1e12:.2f} TB/s, rel err {abs(out.item()-ref)/ref:.2e}")
