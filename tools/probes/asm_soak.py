"""Race screen of the assembly GEMM kernels: every kernel, shapes with several tiles per block and partial rounds, the same launch repeated
many times with other launches in between — all outputs must be bit-identical to the first (a stale LDS slot, an early read of a DMA piece
or a counted wait that is one too lenient shows up as a rare differing tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
dev = "cuda"
torch.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).bfloat16()
REPS = int(os.environ.get("SOAK_REPS", "40"))
side = torch.cuda.Stream()
noise_a, noise_b = rnd(4096, 4096), rnd(4096, 4096)
noise_o = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
big = torch.zeros(256 << 20, device=dev, dtype=torch.float32)
def disturb(i):
    if i % 3 == 0:      # a GEMM on another stream, a bandwidth hog on this one
        with torch.cuda.stream(side):
            hip.linear_fwd(noise_a, noise_b, noise_o)
    if i % 4 == 1:
        big.add_(1.0)
def soak(name, fn):
    ref = fn()
    ref = [r.clone() for r in (ref if isinstance(ref, tuple) else (ref,))]
    bad = 0
    for i in range(REPS):
        disturb(i)
        out = fn()
        out = out if isinstance(out, tuple) else (out,)
        bad += sum(0 if torch.equal(o, r) else 1 for o, r in zip(out, ref))
    torch.cuda.synchronize()
    print(f"{name:44s} {REPS} launches: {'identical' if bad == 0 else str(bad) + ' DIFFERENT'}", flush=True)
    return bad
tot = 0
for M, N, K in ((4352, 4096, 512), (17920, 2560, 2048), (2304, 8192, 1152)):
    x, w, wn = rnd(M, K), rnd(N, K), rnd(K, N)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tot += soak(f"nt {M}x{N}x{K}", lambda: hip.gemm(x, w, o, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=14, ksplit=1).clone())
    tot += soak(f"nn {M}x{N}x{K}", lambda: hip.gemm(x, wn, o, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_kc=False, tile=14, ksplit=1).clone())
    res = rnd(M, N); bias = torch.randn(N, device=dev)
    tot += soak(f"nt+res {M}x{N}x{K}", lambda: hip.gemm(x, w, o, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, residual=res, ldr=N, tile=14, ksplit=1).clone())
    tot += soak(f"nt+bias {M}x{N}x{K}", lambda: hip.gemm(x, w, o, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, tile=14, ksplit=1).clone())
    tot += soak(f"nt+bias+res {M}x{N}x{K}", lambda: hip.gemm(x, w, o, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, residual=res, ldr=N, tile=14, ksplit=1).clone())
    tot += soak(f"nt+bias+gelu {M}x{N}x{K}", lambda: hip.linear_bias_gelu_train(x, w, bias))
    h = rnd(M, N) * 4
    tot += soak(f"nn+dgelu {M}x{N}x{K}", lambda: hip.linear_dgrad_gelu_bwd(x, wn, h))
for M, N, K in ((4096, 2048, 17920), (2048, 4352, 2304), (8192, 1024, 1024)):      # weight gradients: C [M, N] = a^T b over K rows
    a, b = rnd(K, M + 64)[:, :M], rnd(K, N)
    o = torch.empty(M, N, device=dev, dtype=torch.float32)
    tot += soak(f"tn {M}x{N}x{K}", lambda: hip.gemm(a, b, o, M=M, N=N, K=K, lda=a.stride(0), ldb=N, ldc=N, a_kc=False, b_kc=False, tile=14, ksplit=1).clone())
for M, F, K in ((4352, 2048, 512), (17920, 16384, 2048)):
    x, w = rnd(M, K), rnd(2 * F, K) * 0.05
    tot += soak(f"nt+geglu {M}x{2*F}x{K}", lambda: hip.linear_geglu_train(x, w))
    dy, wd = rnd(M, K), rnd(K, F)
    gu = (rnd(M, 2 * F + 64) * 4)[:, :2 * F]
    tot += soak(f"nn+dgeglu {M}x{F}x{K}", lambda: hip.linear_dgrad_geglu_bwd(dy, wd, gu))
print("TOTAL differing outputs:", tot)
