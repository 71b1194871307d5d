"""Client request -> actions through the assembled policy (create_trained_policy: checkpoint params + norm stats + the
request / response transform stacks of lap_amd/policy_io.py) on the GPU, against the same steps done by hand."""
import dataclasses
import json

import numpy as np
import pytest
import torch

from lap_amd import checkpoints, policy_io as pio
from lap_amd.config import get_config
from lap_amd.model import LAP
from lap_amd.observation import CoTObservation
from lap_amd.serve import create_trained_policy, create_trained_policy_ar
from tests.common import tiny_sentencepiece_proto

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _checkpoint(tmp_path, cfg, model):
    (tmp_path / "params").mkdir()
    checkpoints._save_tensors(tmp_path / "params" / "params.safetensors",
                              {"params/" + k: v for k, v in model.ps.to_reference_tree("master").items()})
    stats = {"state": {"mean": [0.0] * 7, "std": [1.0] * 7, "q01": [-2.0] * 7, "q99": [2.0] * 7},
             "actions": {"mean": [0.0] * 7, "std": [1.0] * 7, "q01": [-0.5, -0.4, -0.3, -0.2, -0.1, -1.0, 0.0], "q99": [0.5, 0.4, 0.3, 0.2, 0.1, 1.0, 1.0]}}
    (tmp_path / "assets" / "debug").mkdir(parents=True)
    (tmp_path / "assets" / "debug" / "norm_stats.json").write_text(json.dumps({"norm_stats": stats}))
    return stats


def test_policy_from_checkpoint_serves_raw_requests(hip, tmp_path):
    tc = get_config("debug")
    # (the policy serves with the data config's own CoTInputs, policy_config_adapter.py:137-150: a serving config switches the
    # training-time image randomness off)
    tc = dataclasses.replace(tc, data=dataclasses.replace(tc.data, asset_id="debug", wrist_image_dropout_prob=0.0, random_mask_prob=0.0))
    cfg = tc.model
    model = LAP(cfg, seed=5, device=DEV, with_grads=False)
    stats = _checkpoint(tmp_path, cfg, model)
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=cfg.max_token_len)
    policy = create_trained_policy(tc, tmp_path, tokenizer=tok, default_prompt="pick up the block", use_graph=False, device=DEV)
    rs = np.random.RandomState(3)
    req = {"observation": {"base_0_rgb": (rs.rand(48, 64, 3) * 255).astype(np.uint8),      # not the model's 56 x 56: resized + padded
                           "left_wrist_0_rgb": rs.rand(3, 48, 64).astype(np.float32), "state": rs.uniform(-1, 1, 7)}}
    noise = rs.randn(cfg.action_horizon, cfg.action_dim).astype(np.float32)
    out = policy.infer(req, noise=noise)
    assert set(out) == {"actions", "reasoning", "policy_timing"} and out["actions"].shape == (cfg.action_horizon, cfg.action_dim)
    # by hand: transforms -> observation -> sample_actions -> un-normalise
    inp = pio.compose([pio.InjectDefaultPrompt("pick up the block"), pio.CoTInputs(action_dim=cfg.action_dim), pio.Normalize(stats, "bounds_q99"),
                       pio.TokenizePromptAndReasoning(tok, discrete_state_input=True), pio.PadStatesAndActions(cfg.action_dim)])(dict(req))
    batched = {k: ({kk: np.asarray(vv)[None] for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)[None])
               for k, v in inp.items() if v is not None and not isinstance(v, str)}
    o = CoTObservation.from_dict(batched, device=DEV)
    assert o.images["base_0_rgb"].shape == (1, 48, 64, 3) and bool(o.image_masks["left_wrist_0_rgb"][0])
    a = model.sample_actions(0, o, num_steps=10, noise=torch.from_numpy(noise)[None].to(DEV))[0].cpu().numpy()
    ref = pio.Unnormalize(stats, "bounds_q99")({"actions": a})["actions"]
    np.testing.assert_array_equal(out["actions"], ref)
    q01, q99 = np.array(stats["actions"]["q01"]), np.array(stats["actions"]["q99"])
    np.testing.assert_allclose(out["actions"], (a + 1) / 2 * (q99 - q01 + 1e-6) + q01)
    # a prompt in the request wins over the default (with zero-initialised adaRMS gates the random-init expert ignores the
    # prefix, so this is checked on the model inputs, not on the actions)
    t1 = policy._input_transform(dict(req))["tokenized_prompt"]
    t2 = policy._input_transform(dict(req, prompt="move left 5 cm"))["tokenized_prompt"]
    assert not np.array_equal(t1, t2) and np.array_equal(t1, inp["tokenized_prompt"])
    # autoregressive mode: generated ids + their text
    ar = create_trained_policy_ar(tc, tmp_path, tokenizer=tok, default_prompt="pick up the block", device=DEV, sample_kwargs={"max_decoding_steps": 6})
    r = ar.infer(req)
    # output stack of the AR mode: ids -> text -> one end-effector delta (6 values, + gripper when the text names it)
    assert isinstance(r["reasoning"], str) and r["actions"].shape in ((6,), (7,)) and "policy_timing" in r
    # the reference's wiring carries the data config's training-time randomness into serving; `deterministic=True`
    # switches it off, the default keeps the reference's behaviour and says so loudly
    noisy = dataclasses.replace(tc, data=dataclasses.replace(tc.data, wrist_image_dropout_prob=0.5, random_mask_prob=0.5))
    with pytest.warns(UserWarning, match="training-time randomness"):
        create_trained_policy(noisy, tmp_path, tokenizer=tok, default_prompt="pick up the block", use_graph=False, device=DEV)
    import warnings
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*training-time randomness.*")
        det = create_trained_policy(noisy, tmp_path, tokenizer=tok, default_prompt="pick up the block", use_graph=False, device=DEV, deterministic=True)
    ref_in = det._input_transform(dict(req))
    for _ in range(20):
        again = det._input_transform(dict(req))
        assert bool(again["image_mask"]["left_wrist_0_rgb"]) and np.array_equal(again["tokenized_prompt"], ref_in["tokenized_prompt"])
        np.testing.assert_array_equal(again["image"]["left_wrist_0_rgb"], ref_in["image"]["left_wrist_0_rgb"])


def test_pi0_policy_serves_with_the_continuous_state(hip, tmp_path):
    """`pi05=False` end to end through the serving stack: the checkpoint holds the pi0 tree (state_proj, action_time_mlp_*), the
    prompt carries NO discretised state (`discrete_state_input=False`, lap_config.py:80) and the normalised state reaches the action
    expert as its state token — so two requests that differ in the state alone give different actions (with pi05 they would differ
    through the prompt's state bins instead).  Served eagerly (no captured graph for the generic layer loop)."""
    tc = get_config("debug")
    tc = dataclasses.replace(tc, model=dataclasses.replace(tc.model, pi05=False, discrete_state_input=False),
                             data=dataclasses.replace(tc.data, asset_id="debug", wrist_image_dropout_prob=0.0, random_mask_prob=0.0))
    cfg = tc.model
    model = LAP(cfg, seed=7, device=DEV, with_grads=False)
    stats = _checkpoint(tmp_path, cfg, model)
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=cfg.max_token_len)
    policy = create_trained_policy(tc, tmp_path, tokenizer=tok, default_prompt="pick up the block", use_graph=True, device=DEV)
    assert policy._sampler is None and "act/state_w" in policy.model.ps.names() and "ada/w" not in policy.model.ps.names()
    rs = np.random.RandomState(4)
    img = {"base_0_rgb": (rs.rand(56, 56, 3) * 255).astype(np.uint8), "left_wrist_0_rgb": (rs.rand(56, 56, 3) * 255).astype(np.uint8)}
    noise = rs.randn(cfg.action_horizon, cfg.action_dim).astype(np.float32)
    s1, s2 = rs.uniform(-1, 1, 7), rs.uniform(-1, 1, 7)
    out1 = policy.infer({"observation": dict(img, state=s1)}, noise=noise)
    out1b = policy.infer({"observation": dict(img, state=s1)}, noise=noise)
    out2 = policy.infer({"observation": dict(img, state=s2)}, noise=noise)
    assert out1["actions"].shape == (cfg.action_horizon, cfg.action_dim) and np.isfinite(out1["actions"]).all()
    np.testing.assert_array_equal(out1["actions"], out1b["actions"])
    assert np.abs(out1["actions"] - out2["actions"]).max() > 1e-4
    # by hand: the same transforms with the state kept out of the prompt, then the sampler
    inp = pio.compose([pio.InjectDefaultPrompt("pick up the block"), pio.CoTInputs(action_dim=cfg.action_dim), pio.Normalize(stats, "bounds_q99"),
                       pio.TokenizePromptAndReasoning(tok, discrete_state_input=False), pio.PadStatesAndActions(cfg.action_dim)])({"observation": dict(img, state=s1)})
    batched = {k: ({kk: np.asarray(vv)[None] for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)[None])
               for k, v in inp.items() if v is not None and not isinstance(v, str)}
    o = CoTObservation.from_dict(batched, device=DEV)
    a = model.sample_actions(0, o, num_steps=10, noise=torch.from_numpy(noise)[None].to(DEV))[0].cpu().numpy()
    np.testing.assert_array_equal(out1["actions"], pio.Unnormalize(stats, "bounds_q99")({"actions": a})["actions"])
