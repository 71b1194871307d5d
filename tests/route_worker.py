"""Worker of tests/test_route_parity_gpu.py, run as `python -m tests.route_worker <out.pt>`: ONE full LAP-3B train step (config
`lap_bench`, B = 32, the benchmark's synthetic batch, fixed noise / time) under whatever route switches the environment sets
(LAP_GEMM_NO_ASM, LAP_FUSE_*, LAP_FOLD_SUMSQ ... are read once per process, hence a process per route).  Writes the loss, the
gradient norm, the assembly kernels' launch counts and a strided sample (<= 2^18 elements) of every gradient tensor."""
import os
import sys

import torch


def main(out_path: str) -> None:
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from bench import synthetic_batch
    from lap_amd import hip
    from lap_amd.config import get_config
    from lap_amd.train import TrainingStepRunner, init_train_state

    tc = get_config("lap_bench")
    cfg = tc.model
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("ROUTE_WORKER_BATCH", "32"))
    state = init_train_state(tc, device=dev)
    runner = TrainingStepRunner(tc)
    batch = synthetic_batch(cfg, B, dev, seed=0)
    g = torch.Generator(device="cpu").manual_seed(1)
    noise = torch.randn(B, cfg.action_horizon, cfg.action_dim, generator=g).to(dev)
    time = (torch.rand(B, generator=g) * 0.999 + 0.001).to(dev)
    before = hip.gemm_asm_launch_counts()
    state, info = runner(0, state, batch, 0, noise=noise, time=time)
    torch.cuda.synchronize()
    state.model.comm.synchronize()
    ran = {k: v - before[k] for k, v in hip.gemm_asm_launch_counts().items()}
    ps = state.model.ps
    sample, norms = {}, {}
    for name in ps.names():
        gr = ps.g(name).detach().reshape(-1)
        k = max(1, gr.numel() // 262144)
        sample[name] = gr[::k][:262144].float().cpu().clone()
        norms[name] = float(gr.double().norm())
    torch.save({"loss": float(info["loss"]), "grad_norm": float(info["grad_norm"]), "ran": ran, "sample": sample, "norms": norms,
                "lang_loss": float(info.get("lang_loss", float("nan"))), "action_loss": float(info.get("action_loss", float("nan")))}, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
