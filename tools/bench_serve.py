"""Batch-1 action-chunk latency of LAP-3B on one MI355X (BASELINE.json config 4): prefix prefill + 10 denoise steps."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd.config import get_config
from lap_amd.model import LAP
from lap_amd.serve import GraphedSampler

cfg = get_config(sys.argv[1] if len(sys.argv) > 1 else "lap_bench").model
dev = "cuda"
model = LAP(cfg, seed=0, device=dev, with_grads=False)
g = GraphedSampler(model, 1, 10)
gen = torch.Generator(device="cpu").manual_seed(0)
for k in g.obs.images:
    g.obs.images[k].copy_(torch.rand(1, 224, 224, 3, generator=gen) * 2 - 1)
g.obs.tokenized_prompt.copy_(torch.randint(0, cfg.vocab_size, g.obs.tokenized_prompt.shape, generator=gen, dtype=torch.int32))
noise = torch.randn(1, cfg.action_horizon, cfg.action_dim, generator=gen).to(dev)
g.noise.copy_(noise)

def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

eager = lambda: model.sample_actions(0, g.obs, num_steps=10, noise=g.noise)
for _ in range(2): eager()
t_eager = timeit(eager, 5)
ref = eager().clone()
g.capture()
t_graph = timeit(g.graph.replay, 20)
def one():
    g.graph.replay(); torch.cuda.synchronize()
t_one = timeit(one, 20)
same = torch.equal(ref, g.out)
bytes_min = 4.79e9 + 10 * 0.86e9 + 10 * 10.3e6
print(json.dumps({"metric": "batch-1 action-chunk latency LAP-3B bf16 (prefill + 10 denoise steps)", "eager_ms": round(t_eager, 3),
                  "hipgraph_ms": round(t_graph, 3), "hipgraph_one_at_a_time_ms": round(t_one, 3), "graph_equals_eager": bool(same), "hbm_floor_ms_at_6.29TBps": round(bytes_min / 6.29e12 * 1e3, 2),
                  "achieved_GBps_vs_algorithmic_bytes": round(bytes_min / (t_graph * 1e-3) / 1e9, 1),
                  "prompt_len": cfg.max_token_len, "action_horizon": cfg.action_horizon}))
