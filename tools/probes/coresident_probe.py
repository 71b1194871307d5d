"""Do the optimizer's waves run INSIDE the persistent assembly GEMM's blocks, or do the two kernels take turns?
Stream A: N gate|up forward products (17920 x 32768 x 2048, lap_gemm_asm_nt).  Stream B: AdamW + EMA launches over a 110 M parameter unit
(bf16 gradients).  Timed alone and started together (HIP events per stream); LAP_ADAMW_BLOCKS is read once per process, so the driver
runs one process per setting.
  both ~ max(A, B): the pass lives beside the GEMM's waves (co-resident);  both ~ A + B: they serialise.
usage: python tools/probes/coresident_probe.py            (driver: 240 / 256 / 480 / 1024 / 4096 blocks)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def worker():
    import torch

    from lap_amd import hip

    dev = "cuda:0"
    M, N, K = 17920, 32768, 2048
    x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    n = 110_100_480
    p, m, v, ema = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    g = (torch.randn(n, device=dev) * 1e-3).bfloat16()
    p16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    sc = torch.tensor([1.0, 1e-4, 0.1, 0.05, 0.999, 1.0, 0.0, 0.0], device=dev)
    NG, NA = 12, int(os.environ.get("PROBE_NA", "6"))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def gemms():
        for _ in range(NG):
            hip.linear_fwd(x, w, out=y)

    def opts():
        for _ in range(NA):
            hip.adamw_ema(p, m, v, ema, g, p16, sc, 0.9, 0.95, 1e-8, 1e-4, 1.0)

    def run(fa, fb):
        torch.cuda.synchronize()
        ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        start = torch.cuda.Event()
        start.record()
        sa.wait_event(start); sb.wait_event(start)
        with torch.cuda.stream(sa):
            ea0.record()
            if fa: fa()
            ea1.record()
        with torch.cuda.stream(sb):
            eb0.record()
            if fb: fb()
            eb1.record()
        torch.cuda.synchronize()
        return ea0.elapsed_time(ea1), eb0.elapsed_time(eb1)

    for _ in range(2):
        run(gemms, opts)
    a_alone = min(run(gemms, None)[0] for _ in range(3))
    b_alone = min(run(None, opts)[1] for _ in range(3))
    both = [run(gemms, opts) for _ in range(3)]
    ab = min(max(t) for t in both)
    a_in, b_in = min(t[0] for t in both), min(t[1] for t in both)
    print(f"adamw blocks {os.environ.get('LAP_ADAMW_BLOCKS', '240'):>5s}: {NG} GEMMs alone {a_alone:7.2f} ms | {NA} optimizer launches alone {b_alone:7.2f} ms "
          f"({NA * n * 36 / b_alone / 1e9:.2f} TB/s) | together: GEMMs {a_in:7.2f}, optimizer {b_in:7.2f}, both done after {ab:7.2f} ms "
          f"(sum {a_alone + b_alone:6.2f}, max {max(a_alone, b_alone):6.2f})", flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_WORKER") == "1":
        worker()
    else:
        for blocks in ("240", "256", "480", "1024", "4096"):
            env = dict(os.environ, PROBE_WORKER="1", LAP_ADAMW_BLOCKS=blocks)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
