"""(Isolated, the 256x256 tile + quadrant tail wins 9-14 % on the 560-tile shapes with K <= 2560; in the train step the
same dispatch rule was worth 0.17 ms of 309.7 — not adopted.)  560-tile data-gradient shapes with a short contraction: auto vs the 256x256 tile with its quadrant tail (interleaved, best of 4)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from lap_amd import hip
dev = "cuda:0"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.3).bfloat16()
def t(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for name, M, N, K in (("out dgrad 17920x2048x2048", 17920, 2048, 2048), ("qkv dgrad 17920x2048x2560", 17920, 2048, 2560)):
    dy, w = rnd(M, K), rnd(K, N)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    cfgs = ((-1, 0), (6, 1), (5, 0), (12, 1))
    best = {c: 1e9 for c in cfgs}
    for _ in range(4):
        for tile, ks in cfgs:
            best[(tile, ks)] = min(best[(tile, ks)], t(lambda: hip.linear_dgrad(dy, w, out=out, tile=tile, ksplit=ks)))
    print(name)
    for (tile, ks), us in best.items():
        print(f"    tile {tile:2d}/ks {ks}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF")
