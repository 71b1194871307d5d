"""The B = 32 LAP-3B train step alone (no meter pass, no serving leg, no CPU baseline): what rocprofv3 traces for the phase / layer timelines.
usage: python tools/step_only.py [steps] [warmup]     prints ms per step"""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

from bench import synthetic_batch
from lap_amd.config import get_config
from lap_amd.train import TrainingStepRunner, init_train_state

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(os.environ.get("STEP_BATCH", "32"))
dev = torch.device("cuda", 0)
tc = dataclasses.replace(get_config("lap_bench"), batch_size=B, fsdp_devices=1)
state = init_train_state(tc, device=dev, world_size=1, rank=0, use_fsdp=False)
runner = TrainingStepRunner(tc)
batches = [synthetic_batch(tc.model, B, dev, seed=i) for i in range(2)]
for i in range(warm):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for i in range(steps):
    a = time.perf_counter()
    state, info = runner(0, state, batches[i % 2], state.step)
    host.append((time.perf_counter() - a) * 1e3)
issued = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) * 1e3
print(f"{dt / steps:.2f} ms per step | host ms per runner() call: " + " ".join(f"{h:.1f}" for h in host) + f" | all issued after {issued:.1f} ms of {dt:.1f}")
