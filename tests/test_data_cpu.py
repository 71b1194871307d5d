"""Host-side data path: episode store -> transform stack -> (CoTObservation, actions) batches; resume,
rank sharding, norm stats, idle-sample masks."""
import numpy as np
import pytest
import torch

from lap_amd import data as D
from lap_amd import policy_io as pio
from lap_amd.config import get_config
from tests.common import tiny_sentencepiece_proto


def _episodes(n=3, T=12, seed=0, hw=(40, 48)):
    rs = np.random.RandomState(seed)
    eps = []
    for i in range(n):
        acts = np.concatenate([rs.uniform(-0.01, 0.01, (T, 3)), rs.uniform(-0.05, 0.05, (T, 3)), (rs.rand(T, 1) > 0.5).astype(float)], 1)
        if i == 0:
            acts[:, :6] = 0.0                                            # an episode that never moves: idle labels
        state = np.concatenate([rs.uniform(-0.5, 0.5, (T, 3)), rs.normal(size=(T, 6)), rs.rand(T, 1)], 1)
        e = {"base_0_rgb": rs.randint(1, 255, (T, *hw, 3)).astype(np.uint8), "state": state.astype(np.float32),
             "actions": acts.astype(np.float32), "prompt": f"task number {i}", "dataset_name": "droid"}
        if i != 1:
            e["left_wrist_0_rgb"] = rs.randint(1, 255, (T, *hw, 3)).astype(np.uint8)
        eps.append(e)
    return eps


@pytest.fixture(scope="module")
def setup():
    import dataclasses
    cfg = get_config("debug")
    cfg = dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, action_dim=16))   # room for the 10-value [xyz, rot6d, grip] state
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=cfg.model.max_token_len)
    ds = D.EpisodeDataset(_episodes(), action_horizon=cfg.model.action_horizon)
    return cfg, tok, ds


def test_samples_and_batches(setup):
    cfg, tok, ds = setup
    mc = cfg.model
    assert len(ds) == 36
    s = ds[12 + 9]                                                       # episode 1, step 9: the chunk runs past the end
    e = ds.episodes[1]
    assert s["actions"].shape == (mc.action_horizon, 7) and np.array_equal(s["actions"][:3], e["actions"][9:12])
    assert not s["actions"][3:, :6].any() and np.all(s["actions"][3:, 6] == e["actions"][11, 6])
    np.testing.assert_allclose(s["language_actions"][:6], e["actions"][9:12, :6].sum(0), rtol=1e-6)
    assert s["language_actions"][6] == e["actions"][11, 6] and "left_wrist_0_rgb" not in s["observation"] and not s["has_wrist_image"]
    dl = D.create_data_loader(cfg, ds, tok, shuffle=True, seed=3)
    obs, actions = next(iter(dl))
    B = cfg.batch_size
    assert actions.shape == (B, mc.action_horizon, mc.action_dim) and actions.dtype == torch.float32
    assert obs.images["base_0_rgb"].shape == (B, 40, 48, 3) and obs.images["base_0_rgb"].dtype == torch.float32
    assert obs.images["base_0_rgb"].min() >= -1 and obs.images["base_0_rgb"].max() <= 1
    assert obs.tokenized_prompt.shape == (B, mc.max_token_len) and obs.tokenized_langact_mask.dtype == torch.bool
    assert obs.state.shape == (B, mc.action_dim) and obs.sample_mask.shape == (B,)
    # bounds_q99 statistics of the store: normalised actions of a moving episode stay around [-1, 1]
    assert actions.abs().max() < 3.0
    # idle labels (episode 0 never moves) clear sample_mask; a missing wrist camera is masked out
    seq = D.create_data_loader(cfg, ds, tok, shuffle=False)
    it = iter(seq)
    o0, _ = next(it)
    assert not o0.sample_mask.any() and bool(o0.image_masks["left_wrist_0_rgb"].all())
    for _ in range(5):
        next(it)
    o6, _ = next(it)                                                    # samples 12, 13: episode 1 (no wrist camera, moving)
    assert bool(o6.sample_mask.all()) and not o6.image_masks["left_wrist_0_rgb"].any()
    st = D.compute_norm_stats(ds, action_pad_to=mc.action_dim)
    assert len(st["actions"]["q01"]) == mc.action_dim and len(st["state"]["mean"]) == 10
    with pytest.raises(ValueError, match="cannot fill"):
        D.create_data_loader(cfg, D.EpisodeDataset(_episodes(1, 1), action_horizon=10), tok)


def test_resume_and_rank_shards(setup):
    cfg, tok, ds = setup
    a = D.create_data_loader(cfg, ds, tok, seed=5)
    it = iter(a)
    first = [next(it) for _ in range(20)]                               # crosses an epoch boundary (18 batches per epoch)
    state = a.get_state()
    nxt = [next(it) for _ in range(3)]
    b = D.create_data_loader(cfg, ds, tok, seed=99)
    b.set_state(state)
    again = [next(iter(b)) for _ in range(1)] + [x for _, x in zip(range(2), iter(b))]
    for (o1, a1), (o2, a2) in zip(nxt, again):
        assert torch.equal(a1, a2) and torch.equal(o1.tokenized_prompt, o2.tokenized_prompt) and torch.equal(o1.images["base_0_rgb"], o2.images["base_0_rgb"])
    assert b.get_batches_seen() == 23
    # two ranks see disjoint halves of every global batch and together the single-rank global batch of twice the size
    import dataclasses
    cfg4 = dataclasses.replace(cfg, batch_size=4)
    whole = next(iter(D.create_data_loader(cfg4, ds, tok, seed=1)))
    r0 = next(iter(D.create_data_loader(cfg4, ds, tok, seed=1, rank=0, world_size=2)))
    r1 = next(iter(D.create_data_loader(cfg4, ds, tok, seed=1, rank=1, world_size=2)))
    assert torch.equal(torch.cat([r0[1], r1[1]]), whole[1])
    # validation split: exactly one pass
    val = D.create_data_loader(cfg, ds, tok, shuffle=False, split="val")
    assert sum(1 for _ in val) == 18
    with pytest.raises(ValueError, match="different world size"):
        D.create_data_loader(cfg, ds, tok, rank=0, world_size=2).set_state(state)


def test_episode_files_round_trip(tmp_path):
    eps = _episodes(2, 6)
    for i, e in enumerate(eps):
        np.savez(tmp_path / f"ep{i:03d}.npz", **e)
    ds = D.EpisodeDataset(tmp_path, action_horizon=4)
    assert len(ds) == 12 and ds[7]["prompt"] == "task number 1" and ds[7]["dataset_name"] == "droid"
    np.testing.assert_array_equal(ds[7]["observation"]["base_0_rgb"], eps[1]["base_0_rgb"][1])
    with pytest.raises(FileNotFoundError):
        D.EpisodeDataset(tmp_path / "none", action_horizon=4)
