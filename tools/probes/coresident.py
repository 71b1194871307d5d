"""GEMM time with another kernel resident on some CUs (what a collective does to the persistent assembly blocks).

hipcc -shared -fPIC --offload-arch=gfx950 tools/probes/spin.hip -o tools/probes/spin.so && python tools/probes/coresident.py
Prints ms per GEMM alone, with 16/32 CU-owning spinners, and with 256 CU-sharing spinners, for the asm route and tile 5.
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "spin.so"))
so.spin_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
dev = "cuda:0"
M, N, K = 17920, 16384, 2048
a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
sink = torch.zeros(4, device=dev, dtype=torch.int32)
side = torch.cuda.Stream()
def run(tile, blocks, lds, reps=8):
    torch.cuda.synchronize()
    if blocks:
        so.spin_launch(blocks, 256, lds, 40000, sink.data_ptr(), side.cuda_stream)
        torch.cuda._sleep(2000000)                    # let the spinners land first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm(a, b, c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=tile)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for tile, name in ((14, "asm"), (5, "hip tile 5")):
    run(tile, 0, 0)
    print(name, "alone %.3f ms | 16 owners %.3f | 32 owners %.3f | 64 owners %.3f | 256 sharers %.3f | 1024 sharers %.3f" % (
        run(tile, 0, 0), run(tile, 16, 48 << 10), run(tile, 32, 48 << 10), run(tile, 64, 48 << 10), run(tile, 256, 0), run(tile, 1024, 0)), flush=True)
