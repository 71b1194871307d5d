"""Golden vectors (tests/golden/lap_debug_v1.npz, made by tools/make_golden.py from the CPU oracle).
CPU: the oracle reproduces them (pins the oracle).  GPU: the HIP engine matches them through the C ABI."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import lap_oracle as O
from tests.common import debug_model_cfg, make_inputs, oracle_cfg, rel, to_observation

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lap_debug_v1.npz"))


def _setup():
    cfg = debug_model_cfg()
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=int(G["param_seed"]))
    obs, actions, noise, time = make_inputs(cfg, B=int(G["batch"]), ragged=True, seed=int(G["input_seed"]))
    return cfg, oc, P, obs, actions, noise, time


def test_oracle_reproduces_golden():
    cfg, oc, P, obs, actions, noise, time = _setup()
    assert np.array_equal(P["PaliGemma/llm/layers/attn/q_einsum/w"][0, 0, :4, :4].numpy(), G["p_q0"])
    col = {}
    loss, m = O.compute_loss(P, oc, obs, actions, noise, time, collect=col)
    assert abs(loss.item() - float(G["loss"])) < 1e-4 * abs(float(G["loss"]))
    assert np.allclose(m["per_sample_lang"].numpy(), G["per_sample_lang"], rtol=1e-4)
    assert np.allclose(m["per_sample_action"].numpy(), G["per_sample_action"], rtol=1e-4)
    assert np.array_equal(col["positions"].numpy(), G["positions"])
    assert np.array_equal(col["mask"].sum(-1).numpy(), G["mask_rowsum"])
    assert rel(col["img/out"], torch.from_numpy(G["img_tokens"])) < 1e-4
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"}
    assert rel(O.sample_actions(P, oc, so, noise, num_steps=10), torch.from_numpy(G["sampled_actions"])) < 1e-4
    # padding rows of a sample attend to nothing; valid rows attend to at least one key
    valid = torch.cat([obs["image_masks"][k][:, None].expand(-1, 16) for k in cfg.image_keys] + [obs["tokenized_prompt_mask"]], 1)
    rs = torch.from_numpy(G["mask_rowsum"])[:, :valid.shape[1]]
    assert (rs[valid] > 0).all() and (rs[~valid] == 0).all()


@pytest.mark.gpu
def test_engine_matches_golden(hip):
    from lap_amd.model import LAP

    cfg, oc, P, obs, actions, noise, time = _setup()
    model = LAP(cfg, params=P, device="cuda")
    col = {}
    loss, _ = model.compute_loss(0, to_observation(obs, "cuda"), actions.cuda(), noise=noise.cuda(), time=time.cuda(), collect=col)
    assert abs(loss.item() - float(G["loss"])) / abs(float(G["loss"])) < 5e-3   # bf16 compute vs f32 golden
    assert rel(col["per_sample_lang"], torch.from_numpy(G["per_sample_lang"])) < 2e-2
    assert rel(col["per_sample_action"], torch.from_numpy(G["per_sample_action"])) < 2e-2
    assert np.array_equal(col["pos"].cpu().numpy().astype(np.int64), G["positions"])
    B = int(G["batch"])
    tok = col["img/out"][:B * 16].view(B, 16, -1)
    assert rel(tok, torch.from_numpy(G["img_tokens"])) < 1.5e-2
    valid = torch.cat([obs["image_masks"][k][:, None].expand(-1, 16) for k in cfg.image_keys] + [obs["tokenized_prompt_mask"]], 1)
    x0 = col["x0_out"].view(B, valid.shape[1], -1).float().cpu()
    assert rel(x0[valid], torch.from_numpy(G["x0_last"])[valid]) < 2e-2
    assert rel(col["x1_out"].view(B, cfg.action_horizon, -1), torch.from_numpy(G["x1_last"])) < 2e-2
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"}
    a = model.sample_actions(0, to_observation(so | {"tokenized_langact_mask": None}, "cuda"), num_steps=10, noise=noise.cuda())
    assert rel(a, torch.from_numpy(G["sampled_actions"])) < 2e-2


def test_oracle_loss_branches_decompose_the_joint_loss():
    """lap.py:426-462,557-596.  Prefix rows never attend to suffix rows, so the cross entropy of the prefix-only forward
    (enable_action_training=False) equals the joint forward's per sample, the flow-matching loss without the language loss
    (enable_langact_training=False) equals the joint one's, and the two branch losses add up to the joint loss."""
    import dataclasses

    from tests.common import debug_model_cfg, make_inputs, oracle_cfg

    cfg = debug_model_cfg()
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=3)
    obs, a, n, t = make_inputs(cfg, B=3, ragged=True)
    loss, m = O.compute_loss(P, oc, obs, a, n, t)
    loss_l, m_l = O.compute_loss(P, dataclasses.replace(oc, enable_action_training=False), obs, a, n, t)
    loss_a, m_a = O.compute_loss(P, dataclasses.replace(oc, enable_langact_training=False), obs, a, n, t)
    assert torch.equal(m["per_sample_lang"], m_l["per_sample_lang"]) and "per_sample_action" not in m_l
    assert torch.equal(m["per_sample_action"], m_a["per_sample_action"]) and float(m_a["per_sample_lang"].abs().max()) == 0.0
    assert abs(loss.item() - (loss_l.item() + loss_a.item())) < 1e-5
    # without a sample mask the action-off branch is the batch mean of the weighted per-sample losses (lap.py:594-596)
    obs2 = {k: v for k, v in obs.items() if k != "sample_mask"}
    loss_m, m_m = O.compute_loss(P, dataclasses.replace(oc, enable_action_training=False), obs2, a, n, t)
    assert abs(loss_m.item() - (oc.language_loss_weight * m_m["per_sample_lang"]).mean().item()) < 1e-6
