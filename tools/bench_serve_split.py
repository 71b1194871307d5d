"""Batch-1 action chunk: prefill vs denoise split (graph replay of sample_actions with 10 and with 1 denoise step)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd.config import get_config
from lap_amd.model import LAP
from lap_amd.serve import GraphedSampler

cfg = get_config("lap_bench").model
model = LAP(cfg, seed=0, device="cuda", with_grads=False)
gen = torch.Generator(device="cpu").manual_seed(0)
res = {}
for steps in (10, 1):
    g = GraphedSampler(model, 1, steps)
    for k in g.obs.images:
        g.obs.images[k].copy_(torch.rand(1, 224, 224, 3, generator=gen) * 2 - 1)
    g.obs.tokenized_prompt.copy_(torch.randint(0, cfg.vocab_size, g.obs.tokenized_prompt.shape, generator=gen, dtype=torch.int32))
    g.noise.copy_(torch.randn(1, cfg.action_horizon, cfg.action_dim, generator=gen))
    g.capture()
    for _ in range(3): g.graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.graph.replay()
    torch.cuda.synchronize(); res[steps] = (time.perf_counter() - t0) / 20 * 1e3
step = (res[10] - res[1]) / 9
print(json.dumps({"chunk_ms": round(res[10], 3), "denoise_step_ms": round(step, 4), "per_layer_us": round(step / 18 * 1e3, 2),
                  "prefill_ms": round(res[1] - step, 3)}))
