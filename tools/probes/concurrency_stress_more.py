"""More victims for the co-running-GEMM check (see concurrency_stress.py): attention forward / backward for both head sizes,
GeGLU backward, RoPE split, the cross-entropy kernels — bitwise against their solo results while 128 x 128 GEMM blocks of another
stream share the CUs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
dev = "cuda:0"
rows, W, MLP = 1536, 1152, 4304
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
dh, y2 = rnd(rows, MLP), rnd(rows, W)
outW = torch.empty(MLP, W, device=dev)
side = torch.cuda.Stream()
# SigLIP attention: 6 images x 256 tokens, 16 heads of 72
N, T, H, hd = 6, 256, 16, 72
qkv = rnd(N * T, 3 * W); do = rnd(N * T, W)
def attn72_fwd():
    (o, _), lse = hip.attention_fwd([qkv[:, :W]], [qkv[:, W:2 * W]], [qkv[:, 2 * W:]], [T], [T], N, H, H, hd, scale=hd ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0), need_lse=True)
    return o, lse
o72, lse72 = attn72_fwd()
def attn72_bwd():
    dqkv = torch.empty_like(qkv)
    hip.attention_bwd([qkv[:, :W]], [qkv[:, W:2 * W]], [qkv[:, 2 * W:]], [o72], [do], lse72, [T], [T], N, H, H, hd, scale=hd ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0),
                      dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]])
    return dqkv
# Gemma attention, one segment: 4 samples x 560 tokens, 8 heads of 256, one K/V head
B, Tq, NH, HD = 4, 560, 8, 256
q, k, v, dO = rnd(B * Tq, NH * HD), rnd(B * Tq, HD), rnd(B * Tq, HD), rnd(B * Tq, NH * HD)
def attn256_fwd():
    (o, _), lse = hip.attention_fwd([q], [k], [v], [Tq], [Tq], B, NH, 1, HD, need_lse=True)
    return o, lse
o256, lse256 = attn256_fwd()
def attn256_bwd():
    dq, dk, dv = hip.attention_bwd([q], [k], [v], [o256], [dO], lse256, [Tq], [Tq], B, NH, 1, HD)
    return torch.cat([dq[0].flatten(), dk[0].flatten(), dv[0].flatten()])
gu = rnd(rows, 2 * 2048); dact = rnd(rows, 2048)
pos = torch.arange(Tq, device=dev, dtype=torch.int32)[None].repeat(B, 1).contiguous()
qkvg = rnd(B * Tq, (NH + 2) * HD)
victims = {
    "attention fwd hd 72": lambda: attn72_fwd()[0], "attention bwd hd 72": attn72_bwd,
    "attention fwd hd 256": lambda: attn256_fwd()[0], "attention bwd hd 256": attn256_bwd,
    "geglu_bwd": lambda: hip.geglu_bwd(gu, dact), "geglu_fwd": lambda: hip.geglu_fwd(gu),
    "rope_split_fwd (q)": lambda: hip.rope_split_fwd(qkvg, pos, B, Tq, Tq, 0, NH, HD, HD ** -0.5)[0],
}
for name, f in victims.items():
    ref = f().clone(); torch.cuda.synchronize()
    again = f().clone(); torch.cuda.synchronize()
    solo = torch.equal(ref, again)
    bad = 0
    for rep in range(30):
        with torch.cuda.stream(side):
            for _ in range(8):
                hip.gemm(dh, y2, outW, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=6, ksplit=1)
        out = f(); torch.cuda.synchronize()
        bad += not torch.equal(out, ref)
    print(f"{name:24s} solo-repeatable {solo}  mismatches under load {bad}/30", flush=True)
