"""layernorm_bwd with debug taps under a co-running 128 x 128 GEMM: is a lane's partial sum wrong, or the butterfly's result?
(The answer: the partial sum of s1 in lanes 48..55, before any cross-lane step — and only in a build with packed-f32 VALU ops.)

for v in "" -fno-slp-vectorize; do hipcc -shared -fPIC --offload-arch=gfx950 -O3 -std=c++17 $v -Iinclude tools/probes/lnb_debug.hip -o tools/probes/lnb_debug${v:+_noslp}.so; done
python tools/probes/concurrency_lnb_debug.py lnb_debug.so          -> 86-100 of 100 launches differ
python tools/probes/concurrency_lnb_debug.py lnb_debug_noslp.so    -> 0 of 100
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1] if len(sys.argv) > 1 else "lnb_debug.so"))
so.lnb_debug_launch.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2
dev = "cuda:0"
rows, W, MLP = 1536, 1152, 4304
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
rndf = lambda *s: torch.randn(*s, device=dev, generator=g)
y2, y, dh = rnd(rows, W), rnd(rows, W), rnd(rows, MLP)
gam, mean, rstd = rndf(W), rndf(rows) * 0.01, rndf(rows).abs() + 0.5
outW = torch.empty(MLP, W, device=dev)
side = torch.cuda.Stream()
def run():
    d = torch.empty(rows, W, device=dev, dtype=torch.bfloat16); dg = torch.zeros(W, device=dev); db = torch.zeros(W, device=dev)
    dbg = torch.zeros(rows, 64, 4, device=dev)
    so.lnb_debug_launch(y2.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), d.data_ptr(), dg.data_ptr(), db.data_ptr(),
                        rows, W, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return d, dbg
ref, rdbg = [t.clone() for t in run()]; torch.cuda.synchronize()
shown = 0; fails = 0
for rep in range(100):
    with torch.cuda.stream(side):
        for _ in range(6):
            hip.gemm(dh, y2, outW, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=6, ksplit=1)
    d, dbg = run(); torch.cuda.synchronize()
    if not torch.equal(d, ref) or not torch.equal(dbg, rdbg):
        fails += 1
        if shown < 6:
            shown += 1
            badrows = (dbg != rdbg).view(rows, -1).any(1).nonzero().flatten().tolist()
            row = badrows[0] if badrows else (d != ref).any(1).nonzero().flatten()[0].item()
            pre_bad = (dbg[row, :, 0] != rdbg[row, :, 0]).nonzero().flatten().tolist(), (dbg[row, :, 1] != rdbg[row, :, 1]).nonzero().flatten().tolist()
            post = dbg[row, :, 2], rdbg[row, :, 2]
            print(f"rep {rep}: rows with debug diffs {len(badrows)}, row {row}: lanes with wrong pre-butterfly s1 {pre_bad[0][:8]} s2 {pre_bad[1][:8]}; "
                  f"post-butterfly s1 (all lanes equal? {bool((post[0] == post[0][0]).all())}) got {post[0][0].item():+.6f} ref {post[1][0].item():+.6f} "
                  f"diff*D {(post[0][0] - post[1][0]).item() * W:+.5f}", flush=True)
            print("    got ", [round(v, 4) for v in dbg[row, 48:56, 0].tolist()]); print("    ref ", [round(v, 4) for v in rdbg[row, 48:56, 0].tolist()])
            print("    got-ref", [round(v, 4) for v in (dbg[row, 48:56, 0] - rdbg[row, 48:56, 0]).tolist()])
            gvrow = (y[row].float() * gam)
            print("    gv at cols 384.. (lane 48 p0)", [round(v, 4) for v in gvrow[384:392].tolist()], " cols 896.. (lane 48 p1)", [round(v, 4) for v in gvrow[896:904].tolist()],
                  " sums", round(gvrow[384:392].sum().item(), 4), round(gvrow[896:904].sum().item(), 4))
print("launches with any difference:", fails, "of 100")
