import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import lap_oracle as O
from tests.common import debug_model_cfg, make_inputs, oracle_cfg, rel, to_observation
from lap_amd.model import LAP
cfg = debug_model_cfg(); oc = oracle_cfg(cfg); P = O.init_params(oc, seed=7)
B = 2
obs, actions, noise, time = make_inputs(cfg, B=B, ragged=False)
c32 = {}; l32, m32 = O.compute_loss(P, oc, obs, actions, noise, time, collect=c32)
c16 = {}; l16, m16 = O.compute_loss(P, dataclasses.replace(oc, emulate_bf16=True), obs, actions, noise, time, collect=c16)
model = LAP(cfg, params=P, device="cuda")
col = {}
loss, _ = model.compute_loss(0, to_observation(obs, "cuda"), actions.cuda(), noise=noise.cuda(), time=time.cuda(), collect=col)
print("loss", loss.item(), l32.item(), l16.item())
T = model.n_img_tok; L = cfg.max_token_len; Pn = 2 * T + L
def show(name, a, b32, b16):
    print(f"{name:28s} eng-vs-f32 {rel(a, b32):.3e}   bf16oracle-vs-f32 {rel(b16, b32):.3e}")
# image tower: engine batches images [img0 batch | img1 batch]; oracle collects only first image key
n = B * T
show("img/stem", col["img/stem"][:n].view(B, T, -1), c32["img/stem"], c16["img/stem"])
show("img/block00", col["img/block00"][:n].view(B, T, -1), c32["img/block00"], c16["img/block00"])
show("img/out", col["img/out"][:n].view(B, T, -1), c32["img/out"], c16["img/out"])
pt32, _, _ = O.embed_prefix(P, oc, obs)
show("x0_in", col["x0_in"].view(B, Pn, -1), pt32, pt32)
st32, _, _, cond32 = O.embed_suffix(P, oc, (time[:, None, None] * noise + (1 - time[:, None, None]) * actions), time)
show("x1_in", col["x1_in"].view(B, cfg.action_horizon, -1), st32, st32)
print("pos equal", torch.equal(col["pos"].cpu().long(), c32["positions"]))
for l in range(oc.vlm.depth):
    show(f"layer{l} x0", col[f"llm/layer{l:02d}/x0"].view(B, Pn, -1), c32[f"llm/layer{l:02d}/x0"], c16[f"llm/layer{l:02d}/x0"])
    show(f"layer{l} x1", col[f"llm/layer{l:02d}/x1"].view(B, cfg.action_horizon, -1), c32[f"llm/layer{l:02d}/x1"], c16[f"llm/layer{l:02d}/x1"])
show("pre1", col["pre1"].view(B, cfg.action_horizon, -1), c32["llm/out1"], c16["llm/out1"])
show("v_t", col["v_t"], m32["v_t"], m16["v_t"])
print("per-sample lang", col["per_sample_lang"].cpu(), m32["per_sample_lang"])
print("per-sample act", col["per_sample_action"].cpu(), m32["per_sample_action"])
