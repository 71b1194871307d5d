"""Do the 128x128 GEMM blocks leave a co-resident kernel's registers and LDS alone?  (follow-up to concurrency_stress5.py)
hipcc -shared -fPIC --offload-arch=gfx950 tools/probes/canary.hip -o tools/probes/canary.so && python tools/probes/concurrency_canary.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lap_amd import hip
so = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "canary.so"))
so.canary_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
dev = "cuda:0"
rows, W, MLP = 1536, 1152, 4304
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
dh, y2, w1 = rnd(rows, MLP), rnd(rows, W), rnd(MLP, W)
outW = torch.empty(MLP, W, device=dev); outD = torch.empty(rows, W, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
big = rnd(4096, 4096); wN = rnd(4096, 4096); outN = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
ORDER = sys.argv[1] if len(sys.argv) > 1 else "canary-first"
loads = {"none": lambda: None,
         "TN tile 6": lambda: hip.gemm(dh, y2, outW, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=6, ksplit=1),
         "NN tile 6": lambda: hip.linear_dgrad(dh, w1, out=outD, tile=6),
         "NT asm": lambda: hip.linear_fwd(big, wN, out=outN),
         "TN tile 10": lambda: hip.gemm(dh, y2, outW, M=MLP, N=W, K=rows, lda=MLP, ldb=W, ldc=W, a_kc=False, b_kc=False, tile=10, ksplit=1)}
for name, bg in loads.items():
    for kind, label, lds in ((0, "vgpr canary", 0), (1, "lds canary 36 KB", 36864), (2, "shuffle canary", 0), (2, "shuffle canary 36 KB", 36864)):
        err = torch.tensor([0, 0, 0x7fffffff, 0], device=dev, dtype=torch.int32)
        for rep in range(10):
            if ORDER == "canary-first":     # resident when the GEMM blocks arrive
                so.canary_launch(kind, 512, 300, lds, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
            with torch.cuda.stream(side):
                for _ in range(4):
                    bg()
            if ORDER != "canary-first":     # many short canaries that move into the slots GEMM waves have just left
                so.canary_launch(kind, 16384, 10, lds, err.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        e = err.tolist()
        print(f"background {name:10s} {label:22s} vgpr errors {e[0]}  lds errors {e[1]}  shuffle errors {e[3]}", flush=True)
