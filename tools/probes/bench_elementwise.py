"""Effective HBM bandwidth of the bandwidth-bound kernels at the train step's shapes (isolated): who is below ~5 TB/s?"""
import sys, torch
sys.path.insert(0, "/root/repo")
from lap_amd import hip
dev = "cuda:0"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
def t(f, n=30):
    for _ in range(4): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def show(name, us, nbytes, per_step):
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s   x{per_step:3d}/step = {us * per_step / 1e3:5.2f} ms", flush=True)
R, W, MLP = 16384, 1152, 4304
x, h = rnd(R, W), rnd(R, MLP); gam, bet = torch.randn(W, device=dev), torch.randn(W, device=dev)
show("layernorm_fwd 16384x1152", t(lambda: hip.layernorm_fwd(x, gam, bet)), 2 * R * W * 2, 55)
show("gelu_fwd 16384x4304", t(lambda: hip.gelu_fwd(h)), 2 * R * MLP * 2, 27)
show("gelu_bwd 16384x4304", t(lambda: hip.gelu_bwd(h, h)), 3 * R * MLP * 2, 27)
o1 = torch.zeros(W, device=dev); o4 = torch.zeros(MLP, device=dev); qkv = rnd(R, 3 * W); o3 = torch.zeros(3 * W, device=dev)
show("colsum 16384x1152", t(lambda: hip.colsum(x, o1)), R * W * 2, 0)
show("colsum 16384x3456", t(lambda: hip.colsum(qkv, o3)), R * 3 * W * 2, 27)
show("colsum 16384x4304", t(lambda: hip.colsum(h, o4)), R * MLP * 2, 27)
T, B, NH, HD = 560, 32, 8, 256
rows = B * T
q3 = rnd(rows, (NH + 2) * HD); pos = torch.arange(T, device=dev, dtype=torch.int32)[None].repeat(B, 1).contiguous()
show("rope_split_fwd 17920 rows", t(lambda: hip.rope_split_fwd(q3, pos, B, T, T, 0, NH, HD, HD ** -0.5)), 2 * rows * (NH + 2) * HD * 2, 18)
dq, dk, dv = rnd(rows, NH * HD), rnd(rows, HD), rnd(rows, HD)
show("rope_split_bwd 17920 rows", t(lambda: hip.rope_split_bwd(dq, dk, dv, pos, B, T, T, 0, NH, HD, HD ** -0.5)), 2 * rows * (NH + 2) * HD * 2, 18)
gu = rnd(rows, 2 * 16384); da = rnd(rows, 16384)
show("geglu_fwd 17920x32768 (padded out)", t(lambda: hip.geglu_fwd(gu, pad=True)), 3 * rows * 16384 * 2, 18)
show("geglu_bwd 17920x32768 (padded out)", t(lambda: hip.geglu_bwd(gu, da, pad=True)), 5 * rows * 16384 * 2, 18)
x2 = rnd(rows, 2048); sc = torch.randn(2048, device=dev)
show("rmsnorm_fwd 17920x2048", t(lambda: hip.rmsnorm_fwd(x2, scale=sc, save_rstd=True)), 2 * rows * 2048 * 2, 37)
Rl, V = 1504, 257152
lg = torch.randn(Rl, V, device=dev); tg = torch.randint(0, V, (Rl,), device=dev, dtype=torch.int32)
m = torch.full((Rl,), -3e38, device=dev); l = torch.zeros(Rl, device=dev); tl = torch.zeros(Rl, device=dev); w = torch.ones(Rl, device=dev)
dl = torch.empty(Rl, V, device=dev, dtype=torch.bfloat16)
show("ce_chunk_update 1504x257152 f32", t(lambda: hip.ce_chunk_update(lg, tg, m, l, tl, 0), 10), Rl * V * 4, 1)
hip.ce_chunk_update(lg, tg, m, l, tl, 0)
show("ce_chunk_grad 1504x257152", t(lambda: hip.ce_chunk_grad(lg, tg, m, l, w, dl, 0), 10), Rl * V * 6, 1)
xf = torch.randn(rows, 2048, device=dev)
show("cast_f32_to_bf16 17920x2048", t(lambda: hip.cast_f32_to_bf16(xf)), rows * 2048 * 6, 0)
