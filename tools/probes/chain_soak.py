"""Soak of the one-launch denoise step (csrc/serve_chain.hpp): (a) 1500 replays of the captured batch-1 sampler, every result compared with the
first; (b) the same while another stream keeps the chip busy with persistent 256-block assembly GEMMs (blocks that own whole CUs for ~2 ms) —
the chain's blocks must wait for CUs, the bounded barrier waits must not expire; prints the latency distribution and the failure flag."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
from lap_amd.config import get_config
from lap_amd.model import LAP
from lap_amd.serve import GraphedSampler

cfg = get_config("lap_bench").model
dev = "cuda"
model = LAP(cfg, seed=0, device=dev, with_grads=False)
g = GraphedSampler(model, 1, 10)
gen = torch.Generator(device="cpu").manual_seed(0)
for k in g.obs.images:
    g.obs.images[k].copy_(torch.rand(1, 224, 224, 3, generator=gen) * 2 - 1)
g.obs.tokenized_prompt.copy_(torch.randint(0, cfg.vocab_size, g.obs.tokenized_prompt.shape, generator=gen, dtype=torch.int32))
g.noise.copy_(torch.randn(1, cfg.action_horizon, cfg.action_dim, generator=gen).to(dev))
g.capture()
g.graph.replay(); torch.cuda.synchronize()
ref = g.out.clone()
assert (model.serve_chain and model._chain_ctr is not None) or os.environ.get("LAP_SERVE_CHAIN") == "0"


def run(n, label):
    lat, bad = [], 0
    for i in range(n):
        t0 = time.perf_counter()
        g.graph.replay(); torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
        if not torch.equal(g.out, ref):
            bad += 1
    lat.sort()
    print(f"{label}: {n} chunks, wrong results {bad}, barrier timeouts {model.serve_chain_failed()}, latency ms min {lat[0]:.2f} median {lat[n // 2]:.2f} "
          f"p99 {lat[int(n * 0.99)]:.2f} max {lat[-1]:.2f}", flush=True)


run(int(os.environ.get("SOAK_N", "1500")), "alone")
# (b) a second stream saturating the chip with the training step's biggest assembly GEMM
side = torch.cuda.Stream()
a = torch.randn(17920, 2048, device=dev, dtype=torch.bfloat16)
w = torch.randn(32768, 2048, device=dev, dtype=torch.bfloat16) * 0.02
stop = torch.zeros(1, device=dev)
import threading
flag = {"go": True}


def hammer():
    with torch.cuda.stream(side):
        while flag["go"]:
            for _ in range(8):
                hip.linear_fwd(a, w)
            side.synchronize()


th = threading.Thread(target=hammer); th.start()
time.sleep(0.5)
run(int(os.environ.get("SOAK_BESIDE", "300")), "beside persistent 256-block GEMMs on another stream")
flag["go"] = False; th.join()
run(200, "alone again")
