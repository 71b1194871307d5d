import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
M, N, K = 256, 512, 128
a = (torch.arange(M * K, device=dev) % 251).float().view(M, K).bfloat16()     # distinct-ish values per (row, k)
b = ((torch.arange(N * K, device=dev) * 7) % 241).float().view(N, K).bfloat16()
out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)     # tile 0 writes its dump at C + 0 (tile (0,0)); tile 1 at C + 512 B
hip.call("lap_gemm_nt_asm", hip._p(a), hip._p(b), hip._p(out), M, N, K, K, K, N, hip._stream())
torch.cuda.synchronize()
raw = out.view(-1)[: 65536 // 2].clone()      # 64 KB from the C base: note tile 1's dump starts 512 B later and overlaps; fine for a first look
def image(x):   # expected swizzled image of the first k-tile: [256 rows][8 chunks][8 bf16]
    t = x[:256, :64].reshape(256, 8, 8)
    img = torch.empty_like(t)
    for r in range(256):
        sw = (r >> 1) & 7
        for c in range(8):
            img[r, c ^ sw] = t[r, c]
    return img.reshape(-1)
ea, eb = image(a), image(b)
ga, gb = raw[:16384], raw[16384:32768]
for name, e, g in (("A", ea, ga), ("B", eb, gb)):
    ok = (e == g).view(32, 512).all(-1)      # per 1 KB piece
    print(name, "pieces ok:", "".join("#" if x else "." for x in ok.tolist()))
    bad = (~ok).nonzero().flatten().tolist()
    if bad:
        p = bad[0]
        print("  piece", p, "got", g.view(32, 512)[p][:16].tolist(), "\n  exp", e.view(32, 512)[p][:16].tolist())
print("nonzero elements in the 64 KB dump:", int((raw != 0).sum()), " in whole out:", int((out != 0).sum()))
nz = (out.view(-1) != 0).nonzero().flatten()
if nz.numel():
    print("first nonzero offsets (elements):", nz[:8].tolist(), "values", out.view(-1)[nz[:8]].tolist())
    # which 1 KB pieces (512 elements) of the dump hold anything
    pieces = torch.unique(nz // 512)
    print("pieces with data:", pieces[:64].tolist())
