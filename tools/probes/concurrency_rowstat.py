"""Minimal form of the layernorm_bwd disturbance: per-row sums (with / without the cross-lane butterfly, with / without an LDS
allocation) while 128 x 128 GEMM blocks share the CUs."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "canary.so"))
so.rowstat_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda:0"
rows, W, MLP = 1536, 1152, 4304
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
x, dy, dh, y2 = rnd(rows, W), rnd(rows, W), rnd(rows, MLP), rnd(rows, W)
gam = torch.randn(W, device=dev, generator=g)
outW = torch.empty(MLP, W, device=dev)
side = torch.cuda.Stream()
def stat(mode, lds):
    out = torch.zeros(rows * 128, device=dev)
    so.rowstat_launch(x.data_ptr(), dy.data_ptr(), gam.data_ptr(), out.data_ptr(), rows, mode, lds, torch.cuda.current_stream().cuda_stream)
    return out
for mode, lds, label in ((0, 0, "butterfly, no LDS"), (0, 36864, "butterfly, 36 KB LDS"), (1, 0, "per-lane partials, no LDS"), (1, 36864, "per-lane partials, 36 KB LDS")):
    ref = stat(mode, lds).clone(); torch.cuda.synchronize()
    bad = 0; nrows = 0
    for rep in range(100):
        with torch.cuda.stream(side):
            for _ in range(6):
                hip.gemm(dh, y2, outW, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=6, ksplit=1)
        r = stat(mode, lds); torch.cuda.synchronize()
        ne = (r != ref)
        if ne.any():
            bad += 1; nrows += int(ne.view(rows, -1).any(1).sum())
    print(f"{label:32s} launches with mismatches {bad}/100, rows affected {nrows}", flush=True)
