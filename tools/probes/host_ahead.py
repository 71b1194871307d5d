"""Is the host running ahead of the GPU in the B = 32 train step?  Host time of each runner() call (no synchronisation between
calls) next to the synchronised step time, then a cProfile of three calls sorted by own time.
usage: python tools/probes/host_ahead.py"""
import cProfile
import dataclasses
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

from bench import synthetic_batch
from lap_amd.config import get_config
from lap_amd.train import TrainingStepRunner, init_train_state

B = 32
dev = torch.device("cuda", 0)
tc = dataclasses.replace(get_config("lap_bench"), batch_size=B, fsdp_devices=1)
state = init_train_state(tc, device=dev, world_size=1, rank=0, use_fsdp=False)
runner = TrainingStepRunner(tc)
batches = [synthetic_batch(tc.model, B, dev, seed=i) for i in range(2)]
for i in range(3):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for i in range(6):
    a = time.perf_counter()
    state, info = runner(0, state, batches[i % 2], state.step)
    host.append((time.perf_counter() - a) * 1e3)
issued = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) * 1e3
print("host ms per runner() call:", " ".join(f"{h:.1f}" for h in host), f"| all issued after {issued:.1f} ms, GPU done after {total:.1f} ms ({total / 6:.1f} per step)")
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    state, info = runner(0, state, batches[i % 2], state.step)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
