"""Train loop entry point (scripts/train.py:422-640 surface) on the debug model: interval logging, checkpointing,
resume from the latest checkpoint with the data position, and equivalence of "6 steps + resume + 2 steps" with an
uninterrupted 8-step run (up to the f32 atomics order of the small-unit gradients)."""
import dataclasses
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_resume_matches_uninterrupted_run(hip, tmp_path):
    from lap_amd import checkpoints as ck
    from lap_amd.config import get_config
    from lap_amd.train import main

    base = dataclasses.replace(get_config("debug"), checkpoint_base_dir=str(tmp_path), batch_size=4, log_interval=2,
                               save_interval=3, keep_period=None, seed=3)
    lines = []
    a = main(dataclasses.replace(base, exp_name="full", num_train_steps=8), log=lines.append)
    assert a.step == 8 and any(l.startswith("step 8:") for l in lines) and "samples/s" in lines[-2] and lines[-1] == "saved checkpoint 8"
    assert ck.CheckpointManager(base.checkpoint_base_dir and (tmp_path / base.name / "full")).all_steps() == (8,)
    b = main(dataclasses.replace(base, exp_name="split", num_train_steps=6), log=lambda s: None)
    assert b.step == 6
    lines2 = []
    c = main(dataclasses.replace(base, exp_name="split", num_train_steps=8), log=lines2.append)   # resume=True by default
    assert c.step == 8 and lines2[0].startswith("resumed from step 6")
    # yardstick: two uninterrupted runs differ by the f32 atomics order of some gradient reductions, amplified by bf16
    # rounding over the steps; a resume bug (moments, step count, data position) is orders of magnitude above that
    a2 = main(dataclasses.replace(base, exp_name="full2", num_train_steps=8), log=lambda s: None)

    def worst(p, q):
        """Relative distance per state kind over ALL units together: a per-unit maximum is dominated by units whose norm is
        ~0 (zero-initialised adaRMS bank, freshly started moments), where one reordered f32 atomic is a large fraction."""
        w = 0.0
        for buf in ("master", "m", "v"):
            num = sum(float((getattr(p, buf)[u.name] - getattr(q, buf)[u.name]).double().pow(2).sum()) for u in p.units)
            den = sum(float(getattr(p, buf)[u.name].double().pow(2).sum()) for u in p.units)
            w = max(w, (num / (den + 1e-30)) ** 0.5)
        return w

    # two uninterrupted runs are usually within ~1e-5 of each other and occasionally ~1e-3 (one flipped bf16 rounding early
    # on); lost moments, a wrong data position or a wrong step count move m / v by >= 1e-1
    noise = worst(a.model.ps, a2.model.ps)
    assert worst(a.model.ps, c.model.ps) <= max(5 * noise, 1e-2), (worst(a.model.ps, c.model.ps), noise)
    with pytest.raises(FileExistsError):
        main(dataclasses.replace(base, exp_name="split", num_train_steps=8, resume=False), log=lambda s: None)


def test_train_from_episode_store_then_serve(hip, tmp_path):
    """Data path end to end (SURVEY 8f rank 4 -> train -> rank 1): episodes -> transform stack -> train.main steps with a
    checkpoint carrying the loader's norm stats -> policy assembled from that checkpoint answers a raw request."""
    import dataclasses

    import numpy as np

    from lap_amd import data as D, policy_io as pio
    from lap_amd.config import get_config
    from lap_amd.serve import create_trained_policy
    from lap_amd.train import main
    from tests.common import tiny_sentencepiece_proto
    from tests.test_data_cpu import _episodes

    cfg = get_config("debug")
    cfg = dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, action_dim=16), num_train_steps=4, save_interval=4, log_interval=2,
                              checkpoint_base_dir=str(tmp_path), exp_name="episodes", overwrite=True, resume=False,
                              data=dataclasses.replace(cfg.data, asset_id="toy"))
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=cfg.model.max_token_len)
    ds = D.EpisodeDataset(_episodes(3, 12, hw=(56, 56)), action_horizon=cfg.model.action_horizon)
    loader = D.create_data_loader(cfg, ds, tok, seed=0, device="cuda")
    lines = []
    state = main(cfg, data_loader=loader, device="cuda", log=lines.append)
    assert state.step == 4 and loader.get_batches_seen() == 4 and any("saved checkpoint 4" in str(l) for l in lines)
    losses = [float(str(l).split(": loss=")[1].split(",")[0]) for l in lines if "loss=" in str(l)]
    assert losses and all(np.isfinite(losses))
    ckpt = next(p for p in tmp_path.rglob("4") if p.is_dir())
    stats = json.loads((ckpt / "assets" / "toy" / "norm_stats.json").read_text())["norm_stats"]
    assert len(stats["actions"]["q99"]) == 16
    policy = create_trained_policy(cfg, ckpt, tokenizer=tok, use_graph=False, device="cuda")
    e = ds.episodes[2]
    out = policy.infer({"observation": {"base_0_rgb": e["base_0_rgb"][0], "left_wrist_0_rgb": e["left_wrist_0_rgb"][0], "state": e["state"][0]},
                        "prompt": e["prompt"]})
    assert out["actions"].shape == (cfg.model.action_horizon, 16) and np.isfinite(out["actions"]).all()


def test_train_step_with_image_augmentation(hip):
    """The reference's main config trains with augmentation on (config.py:608-619): the step must run, be reproducible for
    a fixed rng and differ from the un-augmented loss."""
    import dataclasses

    from lap_amd.config import get_config
    from lap_amd.model import LAP
    from tests.common import make_inputs, to_observation

    cfg = dataclasses.replace(get_config("debug").model, enable_image_augmentation=True)
    obs, actions, noise, time = make_inputs(cfg, B=3, ragged=True)
    model = LAP(cfg, seed=2, device="cuda")
    o = to_observation(obs, "cuda")
    l1, _ = model.loss_and_grad(7, o, actions.cuda(), noise=noise.cuda(), time=time.cuda(), train=True)
    l2, _ = model.loss_and_grad(7, o, actions.cuda(), noise=noise.cuda(), time=time.cuda(), train=True)
    l3, _ = model.loss_and_grad(8, o, actions.cuda(), noise=noise.cuda(), time=time.cuda(), train=True)
    l0, _ = model.compute_loss(7, o, actions.cuda(), noise=noise.cuda(), time=time.cuda(), train=False)
    assert torch.isfinite(l1) and float(l1) == float(l2) and float(l1) != float(l3) and float(l1) != float(l0)


def test_backward_without_optimizer_pass_does_not_inflate_the_next_gradient_norm(hip):
    """the per-step sum of squares is cleared by `run_optimizer`; a backward that is NOT followed by an optimizer pass
    (`LAP.loss_and_grad` used directly: gradient checks, an exception mid-step) used to leave its partial sums behind, and the next
    train step's global norm — and clip factor — contained them."""
    from lap_amd.config import get_config
    from lap_amd.train import TrainingStepRunner, init_train_state
    from tests.common import make_inputs, to_observation

    tc = get_config("debug")
    obs, actions, noise, time = make_inputs(tc.model, B=2, ragged=True)
    batch = (to_observation(obs, "cuda"), actions.cuda())
    kw = dict(noise=noise.cuda(), time=time.cuda())
    runner = TrainingStepRunner(tc)
    _, ref = runner(0, init_train_state(tc, seed=3, device="cuda"), batch, 0, **kw)
    state = init_train_state(tc, seed=3, device="cuda")
    for _ in range(2):                                          # two backward passes nobody consumes
        state.model.loss_and_grad(0, *batch, **kw)
    _, info = runner(0, state, batch, 0, **kw)
    torch.cuda.synchronize()
    assert float(info["loss"]) == float(ref["loss"])
    assert abs(float(info["grad_norm"]) - float(ref["grad_norm"])) <= 1e-5 * float(ref["grad_norm"]), (float(info["grad_norm"]), float(ref["grad_norm"]))


def test_frozen_vlm_trains_only_the_action_expert(hip):
    """openpi freeze_filter semantics (scripts/train.py:225-240,358-363) with LAPConfig.get_vlm_freeze_filter: frozen
    parameters keep their (bf16-rounded) values and stay out of the gradient norm; trainable ones move exactly as in an
    unfrozen run fed the same gradients; the prefix stream's backward is skipped."""
    import dataclasses

    from lap_amd.config import get_config
    from lap_amd.train import TrainingStepRunner, init_train_state
    from oracle import lap_oracle as O
    from tests.common import make_inputs, oracle_cfg, rel, to_observation

    tc = get_config("debug")
    cfg = tc.model
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=5)
    frz = cfg.get_vlm_freeze_filter()
    P = {k: (v.to(torch.bfloat16).float() if frz(k) else v) for k, v in P.items()}     # what the frozen store holds
    obs, actions, noise, time = make_inputs(cfg, B=2, ragged=True)
    tcf = dataclasses.replace(tc, freeze_filter=frz)
    state = init_train_state(tcf, params=P, device="cuda")
    assert state.model._prefix_frozen()
    runner = TrainingStepRunner(tcf)
    state, info = runner(0, state, (to_observation(obs, "cuda"), actions.cuda()), 0, noise=noise.cuda(), time=time.cuda())
    torch.cuda.synchronize()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss32, _ = O.compute_loss(Pg, oc, obs, actions, noise, time)
    loss32.backward()
    gn = torch.sqrt(sum((v.grad ** 2).sum() for k, v in Pg.items() if not frz(k)))      # optax.global_norm over the trainable grads
    assert abs(info["grad_norm"].item() - gn.item()) / gn.item() < 3e-2, (info["grad_norm"].item(), gn.item())
    assert abs(info["loss"].item() - loss32.item()) / abs(loss32.item()) < 1e-2
    new = state.model.ps.to_reference_tree("master")
    moved = 0
    for k, v in P.items():
        if frz(k):
            assert torch.equal(new[k], v), k
        else:
            moved += int(not torch.equal(new[k], v))
    assert moved == sum(not frz(k) for k in P) == 19      # every action-expert / action-head array moved
    cs = O.clip_scale(gn.item(), tc.optimizer.clip_gradient_norm)
    for k in ("PaliGemma/llm/layers/mlp_1/linear", "action_out_proj/kernel", "PaliGemma/llm/layers/pre_ffw_norm_1/Dense_0/kernel"):
        p1, _, _ = O.adamw_step(P[k], Pg[k].grad, torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1, tc.lr_schedule(0), tc.optimizer.b1,
                                tc.optimizer.b2, tc.optimizer.eps, tc.optimizer.weight_decay, cs)
        assert rel(new[k] - P[k], p1 - P[k]) < 0.15, k


def test_validation_step_runner_and_loop(hip, tmp_path):
    """scripts/train.py:422-450 (ValidationStepRunner) and :539-571,619-660 (the loop's validation branch): the runner returns
    the loss metrics of `compute_loss(train=False)` plus `val_loss`, touches neither parameters nor optimizer state, scores the
    same fixed subset every time, and `use_validation` makes the loop report it every `val_interval` steps."""
    from lap_amd.config import get_config
    from lap_amd.train import SyntheticDataLoader, ValidationStepRunner, init_train_state, main, run_validation

    tc = dataclasses.replace(get_config("debug"), checkpoint_base_dir=str(tmp_path), batch_size=4, seed=5)
    state = init_train_state(tc, device="cuda")
    val = SyntheticDataLoader(tc.model, 4, "cuda", seed=11, num_batches=2)
    runner = ValidationStepRunner(tc)
    before = {u.name: state.model.ps.master[u.name].clone() for u in state.model.ps.units}
    batch = next(iter(val))
    m = runner(tc.seed, state, batch)
    seed = tc.seed * 1_000_003 + state.step
    loss, metrics = state.model.compute_loss(seed, batch[0], batch[1], train=False)
    assert torch.equal(m["val_loss"], loss) and set(m) == set(metrics) | {"val_loss"}
    assert all(torch.equal(m[k], metrics[k]) for k in metrics)
    a = run_validation(runner, tc.seed, state, val)
    b = run_validation(runner, tc.seed, state, val)          # a fresh iterator: the same two batches again
    assert a == b and set(a) >= {"val_loss", "val_lang_loss", "val_action_loss"} and val.num_val_batches() == 2
    torch.cuda.synchronize()
    assert all(torch.equal(before[u.name], state.model.ps.master[u.name]) for u in state.model.ps.units)
    lines = []
    main(dataclasses.replace(tc, exp_name="val", num_train_steps=4, use_validation=True, val_interval=2, log_interval=2, save_interval=100),
         log=lines.append)
    vl = [l for l in lines if "validation" in l]
    assert len(vl) == 2 and vl[0].startswith("step 2 validation: ") and "val_loss=" in vl[0]
