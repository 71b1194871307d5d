"""The assembly GEMM kernels (csrc/gemm_asm_kernels.s) against the HIP kernels: bitwise check + timing, all three layouts.
Usage: python tools/bench_asm_gemm.py [check|bench|quick] [nt|nn|tn ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
dev = torch.device("cuda:0")
rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()

def timeit(fn, n=10):
    for _ in range(3): fn()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "check"
layouts = sys.argv[2:] or ["nt", "nn", "tn"]
SH = {"check": [(256, 512, 512), (512, 256, 640), (1024, 768, 1152), (2304, 1280, 512), (4096, 4096, 4096), (9216, 8192, 512)],
      "quick": {"nt": [(17920, 32768, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (17920, 2560, 2048)],
                "nn": [(17920, 16384, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (17920, 2048, 32768)],
                "tn": [(32768, 2048, 17920), (2048, 16384, 17920), (4096, 4096, 4096), (8192, 8192, 8192)]},
      "bench": {"nt": [(17920, 32768, 2048), (17920, 2048, 16384), (17920, 2048, 2048), (17920, 2560, 2048), (8192, 8192, 8192), (4096, 4096, 4096), (16384, 3584, 1152)],
                "nn": [(17920, 16384, 2048), (17920, 2048, 32768), (17920, 2048, 2560), (17920, 2048, 2048), (8192, 8192, 8192), (4096, 4096, 4096)],
                "tn": [(32768, 2048, 17920), (2048, 16384, 17920), (2560, 2048, 17920), (2048, 2048, 17920), (8192, 8192, 8192), (4096, 4096, 4096)]}}
for lay in layouts:
    a_kc, b_kc = lay[0] == "n", lay[1] == "t"
    f32 = lay == "tn"
    ref_tile = 10 if lay == "nt" else 12
    shapes = SH[mode] if mode == "check" else SH[mode][lay]
    for M, N, K in shapes:
        a = rnd(M, K) if a_kc else rnd(K, M)
        b = rnd(N, K) if b_kc else rnd(K, N)
        dt = torch.float32 if f32 else torch.bfloat16
        ref = torch.empty(M, N, device=dev, dtype=dt); out = torch.full((M, N), 7.0, device=dev, dtype=dt)
        kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=a_kc, b_kc=b_kc, ksplit=1)
        hip.gemm(a, b, ref, tile=ref_tile, **kw)
        hip.gemm(a, b, out, tile=14, **kw)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        fl = 2.0 * M * N * K
        if mode == "quick":
            t1 = min(timeit(lambda: hip.gemm(a, b, out, tile=14, **kw)) for _ in range(2))
            print(f"{lay} {M} {N} {K} {same} {fl/t1/1e6:.0f} TF/s", flush=True)
            continue
        line = f"{lay} {M:>6} {N:>6} {K:>6}: bitwise {same} max|diff| {(out.float() - ref.float()).abs().max().item():.3g}"
        if mode != "check" or same:
            kw0 = dict(kw); kw0.pop("ksplit")
            t0 = timeit(lambda: hip.gemm(a, b, ref, **dict(kw0, tile=5 if mode == "bench" else ref_tile)))     # bench: the library's own choice (may split)
            t1 = timeit(lambda: hip.gemm(a, b, out, tile=14, **kw))
            line += f" | hip {t0:8.1f} us {fl/t0/1e6:6.0f} TF/s | asm {t1:8.1f} us {fl/t1/1e6:6.0f} TF/s"
        else:
            bad = (out != ref).nonzero()
            line += f" | first mismatches {bad[:4].tolist()} n_bad {bad.shape[0]}"
        print(line, flush=True)
