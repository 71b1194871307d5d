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
    cfg = dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, action_dim=16),   # room for the 10-value [xyz, rot6d, grip] state
                              # (the reference's training-time image randomness — wrist dropout 0.1, random un-masking 0.2, drawn from np.random —
                              # is off here: these tests assert exact masks; test_training_time_image_randomness_follows_the_data_config turns it on)
                              data=dataclasses.replace(cfg.data, wrist_image_dropout_prob=0.0, random_mask_prob=0.0))
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
    np.testing.assert_allclose(s["language_actions"][:3], e["actions"][9:12, :3].sum(0), rtol=1e-6)
    from scipy.spatial.transform import Rotation
    rot = Rotation.identity()
    for rpy in e["actions"][9:12, 3:6]:                                  # rotations compose (base_dataset.py:756-766), they do not add
        rot = rot * Rotation.from_euler("xyz", rpy)
    np.testing.assert_allclose(s["language_actions"][3:6], rot.as_euler("xyz"), atol=1e-6)
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


def test_streamed_norm_stats_equal_the_two_pass_statistics():
    """compute_norm_stats streams episodes and merges per-episode (n, mean, M2) in float64: held to numpy's two-pass
    statistics over the concatenated rows, on a column with a large mean and a small spread (where E[x^2] - mean^2 in one pass loses the
    digits), and with rows that carry NaN (dropped from the state statistics)."""
    rng = np.random.default_rng(0)

    class _Store:
        def __init__(self):
            self.episodes = []
            for n in (7, 19, 4):
                st = rng.normal(size=(n, 3)).astype(np.float32)
                st[:, 0] = (1000.0 + 1e-2 * rng.normal(size=n)).astype(np.float32)
                self.episodes.append({"state": st, "actions": np.zeros((n, 2), np.float32)})
            self.episodes[1]["state"][5, 1] = np.nan
            self._chunks = [(5000.0 + 1e-2 * rng.normal(size=(len(e["actions"]), 4, 2))).astype(np.float32) for e in self.episodes]

        def chunks(self, e):
            return self._chunks[[id(x) for x in self.episodes].index(id(e))]

    ds = _Store()
    st = D.compute_norm_stats(ds)
    rows = np.concatenate([e["state"] for e in ds.episodes], 0)
    rows = rows[np.isfinite(rows).all(1)].astype(np.float64)
    acts = np.concatenate([c.reshape(-1, 2) for c in ds._chunks], 0).astype(np.float64)
    for got, ref in ((st["state"], rows), (st["actions"], acts)):
        np.testing.assert_allclose(got["mean"], ref.mean(0), rtol=1e-12)
        np.testing.assert_allclose(got["std"], ref.std(0), rtol=1e-9)
        np.testing.assert_allclose(got["q01"], np.quantile(ref, 0.01, axis=0), rtol=1e-12)
        np.testing.assert_allclose(got["q99"], np.quantile(ref, 0.99, axis=0), rtol=1e-12)
        np.testing.assert_array_equal(got["min"], ref.min(0)); np.testing.assert_array_equal(got["max"], ref.max(0))
    assert 5e-3 < st["actions"]["std"][0] < 2e-2          # (the one-pass form returns 0 or garbage here: 5000^2 eats 1e-4)


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


# ------------------------------------------------------------------------------------------- dataset mixture
def test_mixture_weights_follow_the_reference_formula():
    """dataset_mixer.py:146-156: balance -> weight x size, normalise; length = max(size / weight)."""
    w, n = D.mixture_weights([1000, 250, 50], [1.0, 2.0, 0.05], balance_weights=True)
    raw = np.array([1000 * 1.0, 250 * 2.0, 50 * 0.05])
    assert np.allclose(w, raw / raw.sum()) and abs(w.sum() - 1) < 1e-12
    assert n == int((np.array([1000, 250, 50]) / w).max())
    w2, n2 = D.mixture_weights([1000, 250, 50], [1.0, 2.0, 0.05], balance_weights=False)
    assert np.allclose(w2, np.array([1.0, 2.0, 0.05]) / 3.05) and n2 == int((np.array([1000, 250, 50]) / w2).max())
    with pytest.raises(ValueError):
        D.mixture_weights([10, 0], [1.0, 1.0])
    assert D.resolve_mixture("libero_finetune")[0] == ("libero_10_no_noops", 1.0) and D.resolve_mixture("droid") == [("droid", 1.0)]
    assert dict(D.resolve_mixture("oxe_magic_soup"))["droid"] == 2.0


def test_mixture_dataset_draws_by_weight_and_is_a_pure_function_of_seed_and_index(setup):
    cfg, tok, _ = setup
    H = cfg.model.action_horizon
    a = D.EpisodeDataset(_episodes(n=4, T=20, seed=1), action_horizon=H)       # 80 transitions
    b = D.EpisodeDataset(_episodes(n=2, T=10, seed=2), action_horizon=H)       # 20 transitions
    for e in b.episodes:
        e["dataset_name"] = "bridge_v2_oxe"
    mix = D.MixtureDataset({"droid": a, "bridge_v2_oxe": b}, [("droid", 1.0), ("bridge_v2_oxe", 3.0)], balance_weights=False, seed=5)
    # sizes in the weights / the effective length are the statistics' transition counts: frames x action horizon (dataset_mixer.py:134-156)
    assert a.num_transitions == 80 * H and np.allclose(mix.sample_weights, [0.25, 0.75]) and len(mix) == int(max(80 * H / 0.25, 20 * H / 0.75))
    picks = np.array([mix.locate(i) for i in range(4000)])
    assert abs((picks[:, 0] == 1).mean() - 0.75) < 0.03                         # dataset frequencies = mixture weights
    inner = picks[picks[:, 0] == 1, 1]
    assert set(inner) == set(range(20)) and np.bincount(inner).std() / np.bincount(inner).mean() < 0.2   # uniform inside a dataset
    again = D.MixtureDataset({"droid": a, "bridge_v2_oxe": b}, [("droid", 1.0), ("bridge_v2_oxe", 3.0)], balance_weights=False, seed=5)
    assert all(again.locate(i) == tuple(picks[i]) for i in range(0, 4000, 97))
    other = D.MixtureDataset({"droid": a, "bridge_v2_oxe": b}, [("droid", 1.0), ("bridge_v2_oxe", 3.0)], balance_weights=False, seed=6)
    assert any(other.locate(i) != tuple(picks[i]) for i in range(50))
    s = mix[3]
    d, i = mix.locate(3)
    assert s["dataset_name"] == ["droid", "bridge_v2_oxe"][d] and np.array_equal(s["actions"], [a, b][d][i]["actions"])
    with pytest.raises(KeyError):
        D.MixtureDataset({"droid": a}, "libero_finetune")
    # balanced: weight x size -> every transition of the pool is equally likely
    bal = D.MixtureDataset({"droid": a, "bridge_v2_oxe": b}, [("droid", 1.0), ("bridge_v2_oxe", 1.0)], balance_weights=True)
    assert np.allclose(bal.sample_weights, [0.8, 0.2]) and len(bal) == 100 * H
    # and it feeds the loader like a single dataset (global statistics, resume protocol)
    loader = D.create_data_loader(cfg, mix, tok, seed=0, num_batches=2)
    obs, actions = next(iter(loader))
    assert actions.shape == (cfg.batch_size, H, cfg.model.action_dim) and torch.isfinite(actions).all()


def test_global_norm_stats_are_the_pooled_moments_and_bracketing_quantiles():
    """datasets/utils/statistics.py:94-236: mean / std of a mixture = the exact moments of the pooled transitions; q01 / q99 / min /
    max bracket the pooled data (min of the lower, max of the upper bounds); VQA sets excluded; states pooled per state type."""
    rs = np.random.RandomState(0)
    data = {"droid": rs.normal(0.1, 1.0, (500, 7)), "bridge_v2_oxe": rs.normal(-0.3, 0.4, (120, 7)), "coco_captions": rs.normal(5, 1, (50, 7))}
    states = {"droid": rs.normal(0, 1, (500, 8)), "bridge_v2_oxe": rs.normal(1, 2, (120, 10)), "coco_captions": rs.normal(0, 1, (50, 8))}

    def stats(x):
        return {"mean": x.mean(0), "std": x.std(0), "q01": np.quantile(x, 0.01, 0), "q99": np.quantile(x, 0.99, 0), "min": x.min(0), "max": x.max(0),
                "num_transitions": len(x), "num_trajectories": len(x) // 10}
    per = {n: {"actions": stats(data[n]), "state": stats(states[n])} for n in data}
    g = D.global_norm_stats(per, action_dim=9, state_dim=10, state_types={"droid": "eef_pose", "bridge_v2_oxe": "joint_pos", "coco_captions": "none"},
                            exclude=["coco_captions"])
    pooled = np.concatenate([data["droid"], data["bridge_v2_oxe"]], 0)
    assert np.allclose(g["actions"]["mean"][:7], pooled.mean(0), atol=1e-5) and np.allclose(g["actions"]["std"][:7], pooled.std(0), atol=1e-5)
    assert np.all(g["actions"]["mean"][7:] == 0) and np.all(g["actions"]["std"][7:] == 0)          # padded to action_dim
    assert g["actions"]["num_transitions"] == 620 and g["actions"]["num_trajectories"] == 62
    assert np.allclose(g["actions"]["q01"][:7], np.minimum(per["droid"]["actions"]["q01"], per["bridge_v2_oxe"]["actions"]["q01"]))
    assert np.allclose(g["actions"]["q99"][:7], np.maximum(per["droid"]["actions"]["q99"], per["bridge_v2_oxe"]["actions"]["q99"]))
    assert np.all(g["actions"]["q01"][:7] <= np.quantile(pooled, 0.01, 0) + 1e-6) and np.all(g["actions"]["q99"][:7] >= np.quantile(pooled, 0.99, 0) - 1e-6)
    assert np.allclose(g["actions"]["min"][:7], pooled.min(0)) and np.allclose(g["actions"]["max"][:7], pooled.max(0))
    assert set(g) == {"actions", "state_eef_pose", "state_joint_pos"}
    assert np.allclose(g["state_eef_pose"]["mean"][:8], states["droid"].mean(0), atol=1e-5) and g["state_eef_pose"]["num_transitions"] == 500
    assert np.allclose(g["state_joint_pos"]["std"], states["bridge_v2_oxe"].std(0), atol=1e-5)
    empty = D.global_norm_stats({"coco_captions": per["coco_captions"]}, action_dim=4, state_dim=4, exclude=["coco_captions"])
    assert empty["actions"]["num_transitions"] == 0 and np.all(empty["actions"]["std"] == 1) and set(empty) == {"actions"}


# ------------------------------------------------------------------------------ offline RLDS exporter (lap_amd/rlds_export.py)
def test_rlds_export_rotation_helpers_match_scipy():
    """rotation_utils.py:84-160,453-471 and transforms.py:103-133 restated in numpy, against scipy's rotations."""
    from scipy.spatial.transform import Rotation

    from lap_amd import rlds_export as R

    g = np.random.default_rng(0)
    eul = np.stack([g.uniform(-3, 3, 64), g.uniform(-1.4, 1.4, 64), g.uniform(-3, 3, 64)], -1)
    want = Rotation.from_euler("xyz", eul).as_matrix()                      # extrinsic xyz = Rz Ry Rx
    assert np.allclose(R.euler_to_rotation_matrix(eul), want, atol=1e-12)
    assert np.allclose(R.rotation_matrix_to_euler(want), eul, atol=1e-9)
    e1, e2 = eul[:32], eul[32:]
    rel = R.euler_diff(e1, e2)
    assert np.allclose(Rotation.from_euler("xyz", e2).as_matrix() @ Rotation.from_euler("xyz", rel).as_matrix(),
                       Rotation.from_euler("xyz", e1).as_matrix(), atol=1e-9)
    rv = g.normal(size=(64, 3)) * g.uniform(0.0, 2.5, (64, 1))
    rv[0] = 0.0                                                             # zero rotation: the x-axis fallback
    assert np.allclose(Rotation.from_euler("xyz", R.axis_angle_to_extrinsic_xyz_euler(rv)).as_matrix(), Rotation.from_rotvec(rv).as_matrix(), atol=1e-9)


def test_rlds_export_libero_and_droid_record_to_episode():
    """transforms.py:1453-1481 (LIBERO) and :757-790 (DROID) on hand-built trajectories, through `episode_from_rlds` into the
    episode store and out of `EpisodeDataset` as the sample the train loader sees."""
    from lap_amd import data as D, rlds_export as R

    T = 6
    g = np.random.default_rng(1)
    img = g.integers(0, 255, (T, 8, 8, 3), dtype=np.uint8)
    # LIBERO: state [xyz, axis-angle, 2 finger joints]; action gripper -1 (open) .. +1 (close)
    rot = np.array([[0.0, 0.0, 0.1 * t] for t in range(T)])                 # pure yaw, 0.1 rad per step
    st = np.concatenate([np.arange(T)[:, None] * np.array([[0.01, 0.0, -0.02]]), rot, np.full((T, 1), 0.02), np.full((T, 1), -0.02)], 1)
    act = np.concatenate([g.normal(size=(T, 6)), np.array([[-1.0], [-1.0], [0.3], [1.0], [1.0], [-0.5]])], 1)
    traj = {"observation": {"image": img, "wrist_image": img[::-1].copy(), "state": st}, "action": act,
            "language_instruction": np.array([b"put the bowl on the plate"] * T)}
    ep = R.episode_from_rlds("libero_10_no_noops", traj)
    assert ep["prompt"] == "put the bowl on the plate" and ep["dataset_name"] == "libero_10_no_noops"
    assert ep["base_0_rgb"].dtype == np.uint8 and np.array_equal(ep["left_wrist_0_rgb"], img[::-1])
    assert np.allclose(ep["state"][:, :3], st[:, :3]) and np.allclose(ep["state"][:, 3:6], rot, atol=1e-7)      # yaw-only: euler = axis-angle
    assert np.allclose(ep["state"][:, 6], 0.5)                                                                 # 0.02 / 0.04
    want_grip = 1.0 - np.clip(act[:, -1], 0, 1)                                                                # 1 = open
    assert np.allclose(ep["actions"][:, 6], want_grip)
    assert np.allclose(ep["actions"][:-1, :3], [[0.01, 0.0, -0.02]] * (T - 1), atol=1e-7) and np.allclose(ep["actions"][:-1, 3:6], [[0, 0, 0.1]] * (T - 1), atol=1e-6)
    assert np.allclose(ep["actions"][-1, :6], 0.0)                                                             # zero-padded last step
    assert R.episode_from_rlds("libero_10_no_noops", dict(traj, language_instruction=np.array([b""] * T))) is None
    # what the exporter adds for the loader's trajectory transforms: the reference's `action` (LIBERO: the raw controller action with the
    # gripper flipped), how it is chunked, the dataset's clock, the state's encoding
    assert np.allclose(ep["target_actions"], np.concatenate([act[:, :6], want_grip[:, None]], 1), atol=1e-6)
    assert ep["chunk_mode"] == "window_zero" and int(ep["control_frequency"]) == 15 and ep["state_encoding"] == "pos_euler"
    ds = D.EpisodeDataset([ep], action_horizon=4)
    s = ds[1]
    # label window: 1 s x 15 Hz = 15 steps, cut at the episode's end (steps 1..5; the last step's movement is the zero padding)
    assert np.allclose(s["language_actions"][:3], 4 * np.array([0.01, 0.0, -0.02]), atol=1e-6) and np.isclose(s["language_actions"][5], 0.4, atol=1e-5)
    assert s["language_actions"][6] == want_grip[5] and s["has_wrist_image"] and np.isclose(s["time_horizon_seconds"], 5 / 15)
    # LIBERO chunk: rows 1..4 of the raw action; from step 4 on the window runs past the end and reads zeros (oxe_datasets.py:259-269)
    assert s["actions"].shape == (4, 7) and np.allclose(s["actions"], ep["target_actions"][1:5], atol=1e-6)
    assert np.allclose(ds[4]["actions"][:2], ep["target_actions"][4:6], atol=1e-6) and not ds[4]["actions"][2:].any()
    # the model-side state: [xyz, rot6d, gripper] (base_dataset.py:437-456); pure yaw psi: first two columns of Rz(psi)
    psi = 0.1
    assert s["observation"]["state"].shape == (10,) and np.allclose(s["observation"]["state"][3:9], [np.cos(psi), np.sin(psi), 0, -np.sin(psi), np.cos(psi), 0], atol=1e-6)
    assert np.array_equal(s["raw_state"], s["observation"]["state"]) and np.isclose(s["observation"]["state"][9], 0.5)
    # DROID: gripper_position 0 open .. 1 closed, rank-1; in-between values take the next decided step
    cart = np.concatenate([np.arange(T)[:, None] * np.array([[0.0, 0.02, 0.0]]), np.zeros((T, 3))], 1)
    gp = np.array([0.0, 0.1, 0.45, 0.55, 0.9, 1.0])
    dtraj = {"observation": {"exterior_image_1_left": img, "wrist_image_left": img, "cartesian_position": cart, "gripper_position": gp},
             "action_dict": {"gripper_position": gp[:, None]}, "language_instruction": b"open the drawer"}
    de = R.episode_from_rlds("droid", dtraj)
    assert np.allclose(de["state"][:, 6], [1, 1, 1, 0, 0, 0]) and np.allclose(de["actions"][:, 6], [1, 1, 1, 0, 0, 0])
    assert np.allclose(de["actions"][:-1, 1], 0.02) and np.allclose(de["actions"][-1, :6], 0)
    # DROID chunk (base_dataset.py:387-427): displacement from the CURRENT pose over a last-value-padded window of absolute poses, the
    # gripper command of the step itself
    assert de["chunk_mode"] == "relative" and int(de["control_frequency"]) == 15 and np.allclose(de["target_actions"][:, :6], cart)
    dd = D.EpisodeDataset([de], action_horizon=4)
    c3 = dd[3]["actions"]
    assert np.allclose(c3[:, 1], [0.02, 0.04, 0.04, 0.04], atol=1e-6) and not c3[:, [0, 2, 3, 4, 5]].any() and np.allclose(c3[:, 6], [0, 0, 0, 0])
    assert np.allclose(dd[0]["actions"][:, 1], [0.02, 0.04, 0.06, 0.08], atol=1e-6) and np.allclose(dd[0]["actions"][:, 6], [1, 1, 1, 0])
    st = D.compute_norm_stats(dd)                    # statistics over the CHUNKS of every frame (base_dataset.py:297-312), 6 x 4 rows
    assert np.isclose(st["actions"]["max"][1], 0.08) and np.isclose(st["actions"]["min"][1], 0.0) and len(st["state"]["mean"]) == 10
    b = R.binarize_gripper_actions(np.array([0.5, 0.5, 0.99, 0.5, 0.01, 0.5]))                                  # default threshold 0.95
    assert np.allclose(b, [1, 1, 1, 0, 0, 0.5])                                                                # the tail keeps the last raw value
    with pytest.raises(KeyError):
        R.episode_from_rlds("kuka", traj)


def test_rlds_export_helpers_of_the_oxe_transforms():
    """transform_helpers.py:57-82,165-189, rotation_utils.py:382-450,504-518 and tensorflow_graphics' quaternion -> euler, restated
    in numpy: against scipy and hand-computed cases."""
    from scipy.spatial.transform import Rotation

    from lap_amd import rlds_export as R

    g = np.random.default_rng(2)
    eul = np.stack([g.uniform(-3, 3, 32), g.uniform(-1.4, 1.4, 32), g.uniform(-3, 3, 32)], -1)
    rot = Rotation.from_euler("xyz", eul)
    assert np.allclose(R.quaternion_xyzw_to_euler(rot.as_quat()), eul, atol=1e-9)                      # scipy quaternions are xyzw
    # column-major 4 x 4 pose -> [xyz, rpy, gripper / 0.079 clipped]
    T = np.tile(np.eye(4), (32, 1, 1)); T[:, :3, :3] = rot.as_matrix(); T[:, :3, 3] = g.normal(size=(32, 3))
    flat = np.swapaxes(T, 1, 2).reshape(32, 16)                                                          # column-major flattening
    st = R.extract_state_from_matrix(flat, np.linspace(-0.01, 0.1, 32)[:, None])
    assert np.allclose(st[:, :3], T[:, :3, 3]) and np.allclose(st[:, 3:6], eul, atol=1e-9)
    assert st[0, 6] == 0.0 and st[-1, 6] == 1.0 and np.isclose(st[16, 6], np.linspace(-0.01, 0.1, 32)[16] / 0.079)
    # relative -> absolute gripper: holds between commands, starts as the opposite of the first command, open without any
    assert np.allclose(R.rel2abs_gripper_actions([0, 0, 0.5, 0, 0, -0.7, 0.05, 0]), [1, 1, 0, 0, 0, 1, 1, 1])
    assert np.allclose(R.rel2abs_gripper_actions([0, -1, 0, 1]), [0, 1, 1, 0])
    assert np.allclose(R.rel2abs_gripper_actions([0.05, -0.05, 0]), [1, 1, 1])
    # a frame change acts on the translation as C t and on the rotation as C R C^T
    for C in (R.TRANSFORM_BCZ, R.TRANSFORM_JACO):
        m = np.concatenate([g.normal(size=(8, 3)), eul[:8]], 1)
        out = R.apply_coordinate_transform(m, C)
        assert np.allclose(out[:, :3], m[:, :3] @ C.T)
        assert np.allclose(Rotation.from_euler("xyz", out[:, 3:]).as_matrix(), C @ rot[:8].as_matrix() @ C.T, atol=1e-9)
    assert np.allclose(R.apply_coordinate_transform(np.array([[1.0, 2.0, 3.0, 0, 0, 0]]), R.TRANSFORM_JACO)[0, :3], [-2.0, 1.0, 3.0])
    assert np.allclose(R.apply_coordinate_transform(np.array([[1.0, 2.0, 3.0, 0, 0, 0]]), R.TRANSFORM_BCZ)[0, :3], [-2.0, -1.0, -3.0])
    # fallback instruction: only for blank instructions; deterministic needs the injected hash, random draws from the 18 phrases
    assert R.fill_empty_language_instruction("pick up", 0.0) == "pick up" and len(R.FALLBACK_INSTRUCTIONS) == 18
    assert R.fill_empty_language_instruction("  ", 1.5, hash_bucket=lambda v, n: 3) == "Carry out the objective."
    with pytest.raises(ValueError):
        R.fill_empty_language_instruction("", 1.5)
    assert R.fill_empty_language_instruction("", 0.0, deterministic=False, rng=np.random.default_rng(0)) in R.FALLBACK_INSTRUCTIONS


def test_rlds_export_covers_the_training_mixture_of_the_lap_config():
    """Every dataset of `oxe_magic_soup` (mixtures.py:2-22, the `lap` config's data_mix) has its standardisation transform
    (transforms.py) restated: hand-built raw trajectories in each dataset's own schema come out with one convention — state =
    [xyz, extrinsic-XYZ euler, gripper (1 = open)], per-step language action = pose delta (zero at the last step) + gripper command."""
    from scipy.spatial.transform import Rotation

    from lap_amd import data as D, rlds_export as R

    assert set(D.NAMED_MIXTURES["oxe_magic_soup"] if isinstance(D.NAMED_MIXTURES["oxe_magic_soup"], dict) else dict(D.NAMED_MIXTURES["oxe_magic_soup"])) <= set(R.STANDARDIZE)
    T = 5
    g = np.random.default_rng(3)
    img = g.integers(0, 255, (T, 4, 4, 3), dtype=np.uint8)
    xyz = np.cumsum(g.normal(size=(T, 3)) * 0.01, 0)
    eul = np.stack([np.linspace(0.1, 0.3, T), np.linspace(-0.2, 0.1, T), np.linspace(1.0, 1.4, T)], -1)
    rot = Rotation.from_euler("xyz", eul)
    quat = rot.as_quat()                                                                                  # xyzw
    Tm = np.tile(np.eye(4), (T, 1, 1)); Tm[:, :3, :3] = rot.as_matrix(); Tm[:, :3, 3] = xyz
    flat = np.swapaxes(Tm, 1, 2).reshape(T, 16)
    lang = np.array([b"stack the cups"] * T)
    pose = np.concatenate([xyz, eul], 1)
    mov = R.compute_padded_movement_actions(pose)
    assert np.allclose(mov[-1], 0) and np.allclose(mov[:-1, :3], np.diff(xyz, axis=0))
    closed = np.array([0.0, 0.0, 1.0, 1.0, 0.0])[:, None]           # 1 = closed in most raw schemas

    def check(name, traj, want_state_grip, want_cmd, want_pose=pose, want_mov=mov, prompt="stack the cups", **kw):
        ep = R.episode_from_rlds(name, traj, **kw)
        assert ep is not None and ep["prompt"] == prompt and ep["dataset_name"] == name, name
        assert ep["state"].shape == (len(want_pose), 7) and ep["actions"].shape == (len(want_pose), 7), name
        assert np.allclose(ep["state"][:, :6], want_pose, atol=1e-6), name
        assert np.allclose(ep["state"][:, 6], np.ravel(want_state_grip), atol=1e-6), name
        assert np.allclose(ep["actions"][:, :6], want_mov, atol=1e-6), name
        assert np.allclose(ep["actions"][:, 6], np.ravel(want_cmd), atol=1e-6), name
        b, w = R.IMAGE_KEYS[name]
        assert ep["base_0_rgb"].shape[0] == len(want_pose) and (("left_wrist_0_rgb" in ep) == (w is not None)), name
        return ep

    nli = {"natural_language_instruction": lang}
    # RT-1: quaternion pose, gripper_closed state, relative gripper command (+ close / - open)
    check("fractal20220817_data", {"observation": {"image": img, "base_pose_tool_reached": np.concatenate([xyz, quat], 1), "gripper_closed": closed, **nli},
                                   "action": {"gripper_closedness_action": np.array([0, 1.0, 0, -1.0, 0])[:, None], "world_vector": xyz, "rotation_delta": eul}},
          1 - closed, [1, 0, 0, 1, 1])
    # Bridge V2 (website version): first step dropped, 7-dim state, gripper command binarised
    st7 = np.concatenate([pose, np.array([0.2, 1.3, -0.1, 0.5, 0.9])[:, None]], 1)
    act7 = np.concatenate([g.normal(size=(T, 6)), np.array([1.0, 1.0, 0.5, 0.0, 0.0])[:, None]], 1)
    check("bridge_v2_oxe", {"observation": {"image_0": img, "state": st7}, "action": act7, "language_instruction": lang, "traj_metadata": {"episode_id": 7}},
          np.clip(st7[1:, 6], 0, 1), [1, 0, 0, 0], want_pose=pose[1:], want_mov=R.compute_padded_movement_actions(pose[1:]))
    # taco_play: robot_obs = [pose, gripper width ...]; command in -1 .. 1
    ro = np.concatenate([pose, np.array([0.0, 0.04, 0.0807, 0.1, -0.01])[:, None], np.zeros((T, 8))], 1)
    check("taco_play", {"observation": {"rgb_static": img, "rgb_gripper": img, "robot_obs": ro, **nli}, "action": {"rel_actions_world": np.concatenate([np.zeros((T, 6)), np.array([-1, -1, 1, 0.5, 3.0])[:, None]], 1)}},
          np.clip(12.3903 * ro[:, 6], 0, 1), [0, 0, 1, 0.75, 1])
    # jaco_play: pose rotated into x' = -y, y' = x
    jp = R.apply_coordinate_transform(pose, R.TRANSFORM_JACO)
    check("jaco_play", {"observation": {"image": img, "image_wrist": img, "end_effector_cartesian_pos": np.concatenate([pose, np.array([0.0, 0.1, 0.2, 0.3, 0.15])[:, None]], 1), **nli},
                        "action": {"gripper_closedness_action": np.array([0, 0, 1.0, 0, -1.0])[:, None], "world_vector": xyz}},
          np.clip(np.array([0.0, 0.1, 0.2, 0.3, 0.15]) * 4.33, 0, 1), [1, 1, 0, 0, 1], want_pose=jp, want_mov=R.compute_padded_movement_actions(jp))
    # viola / mutex / austin_*: column-major pose matrices
    width = np.array([0.0, 0.0395, 0.079, 0.1, 0.02])[:, None]
    check("viola", {"observation": {"agentview_rgb": img, "eye_in_hand_rgb": img, "ee_states": flat, "gripper_states": width, **nli},
                    "action": {"gripper_closedness_action": np.array([-1.0, 0.0, 1.0, 0.5, 2.0])}}, np.clip(width / 0.079, 0, 1), [1, 1, 0, 0.5, 0])
    st24 = np.concatenate([np.zeros((T, 7)), width, flat], 1)                                            # [7 joints, gripper, 16 pose]
    a7 = np.concatenate([np.zeros((T, 6)), np.array([-1.0, 0.0, 1.0, 0.3, 1.0])[:, None]], 1)
    check("utaustin_mutex", {"observation": {"image": img, "wrist_image": img, "state": st24}, "action": a7, "language_instruction": lang},
          np.clip(width / 0.079, 0, 1), [1, 1, 0, 0.7, 0])
    blank = np.array([b""] * T)
    ep = check("austin_buds_dataset_converted_externally_to_rlds", {"observation": {"image": img, "wrist_image": img, "state": st24}, "action": a7, "language_instruction": blank},
               np.clip(width / 0.079, 0, 1), [1, 1, 0, 0.7, 0], prompt=R.FALLBACK_INSTRUCTIONS[5], hash_bucket=lambda v, n: 5)
    st8 = np.concatenate([np.zeros((T, 7)), width], 1)
    for name, kw in (("austin_sailor_dataset_converted_externally_to_rlds", dict(hash_bucket=lambda v, n: 0)), ("austin_sirius_dataset_converted_externally_to_rlds", dict(rng=np.random.default_rng(1)))):
        e2 = R.episode_from_rlds(name, {"observation": {"image": img, "wrist_image": img, "state": st8, "state_ee": flat}, "action": a7, "language_instruction": blank}, **kw)
        assert e2["prompt"] in R.FALLBACK_INSTRUCTIONS and np.allclose(e2["state"][:, :6], pose, atol=1e-6) and np.allclose(e2["actions"][:, 6], [1, 1, 0, 0.7, 0])
    # quaternion-pose datasets
    check("furniture_bench_dataset_converted_externally_to_rlds", {"observation": {"image": img, "wrist_image": img, "state": np.concatenate([xyz, quat, width], 1)}, "action": a7, "language_instruction": lang},
          np.clip(width / 0.079, 0, 1), [1, 1, 0, 0.7, 0])
    check("fmb", {"observation": {"image_side_1": img, "image_wrist_2": img, "eef_pose": np.concatenate([xyz, quat], 1), "state_gripper_pose": closed[:, 0]}, "action": np.concatenate([np.zeros((T, 6)), closed], 1), "language_instruction": lang},
          1 - closed, 1 - closed)
    rs = np.concatenate([np.zeros((T, 6)), xyz, quat, closed, np.zeros((T, 1))], 1)
    check("berkeley_autolab_ur5", {"observation": {"image": img, "hand_image": img, "image_with_depth": img[..., :1], "robot_state": rs, **nli},
                                   "action": {"gripper_closedness_action": np.array([0, 0, 1.0, 0, -1.0]), "world_vector": xyz, "rotation_delta": eul}}, 1 - closed, [1, 1, 0, 0, 1])
    # fanuc: the stored action is the movement part of the language action, the gripper STATE stands in for the command
    fa = g.normal(size=(T, 6))
    check("berkeley_fanuc_manipulation", {"observation": {"image": img, "wrist_image": img, "state": np.concatenate([np.zeros((T, 6)), closed], 1), "end_effector_state": np.concatenate([xyz, quat], 1)},
                                          "action": fa, "language_instruction": lang}, 1 - closed, 1 - closed, want_mov=fa)
    # molmoact: the stored action IS the language action; gripper conventions inverted
    ma = np.concatenate([g.normal(size=(T, 6)), closed], 1)
    check("molmoact_dataset", {"observation": {"first_view_image": img, "wrist_image": img, "state": np.concatenate([pose, closed], 1)}, "action": ma, "language_instruction": lang},
          1 - closed, 1 - closed, want_mov=ma[:, :6])
    # bc_z: axis-angle pose in the frame x' = -y, y' = -x, z' = -z; sensed_close scaled by 0.8
    bp = R.apply_coordinate_transform(pose, R.TRANSFORM_BCZ)
    sensed = np.array([0.0, 0.2, 0.6, 1.0, 0.5])[:, None]
    check("bc_z", {"observation": {"image": img, "present/xyz": xyz, "present/axis_angle": rot.as_rotvec(), "present/sensed_close": sensed, **nli},
                   "action": {"future/xyz_residual": np.zeros((T, 30)), "future/axis_angle_residual": np.zeros((T, 30)), "future/target_close": np.tile(closed, (1, 10)).astype(np.int64)}},
          np.clip((1 - sensed) / 0.8, 0, 1), 1 - closed, want_pose=bp, want_mov=R.compute_padded_movement_actions(bp))
    # an instruction-free trajectory of a dataset without a fallback is dropped
    assert R.episode_from_rlds("fmb", {"observation": {"image_side_1": img, "image_wrist_2": img, "eef_pose": np.concatenate([xyz, quat], 1), "state_gripper_pose": closed[:, 0]},
                                       "action": np.concatenate([np.zeros((T, 6)), closed], 1), "language_instruction": blank}) is None


# ------------------------------------------------------------------------------ VQA readers (lap_amd/vqa_export.py)
def test_vqa_tables_and_geometry_match_the_reference_fixture():
    """Every prompt table of datasets/vqa/*.py and bbox/prompts.py, the loc-token text of 186 boxes and the direction words of 486
    (box, slope) cases, against tests/golden/vqa_v1.json — produced from the reference's own sources by make_vqa_golden.py."""
    import json
    import pathlib

    from lap_amd import vqa_export as V

    d = json.loads((pathlib.Path(__file__).parent / "golden" / "vqa_v1.json").read_text())
    for k in ("COCO_CAPTION_PROMPTS", "PIXMO_CAP_PROMPTS"):
        assert list(getattr(V, k)) == d[k], k
    for k in ("PIXMO_POINT_PROMPT_PARTS", "GENERAL_BBOX_PROMPT_PARTS", "ROBOT_BBOX_PROMPT_PARTS", "ROBOT_BBOX_PROMPT_PARTS_OXE", "ROBOT_BBOX_PROMPT_PARTS_EE",
              "DIRECTION_PROMPT_PARTS", "ROBOT_DIRECTION_PROMPT_PARTS_OXE", "ROBOT_DIRECTION_PROMPT_PARTS_EE"):
        assert [list(p) for p in getattr(V, k)] == d[k], k
    assert V.MAX_POINTS == d["MAX_POINTS"]
    for c in d["loc_cases"]:
        assert V.bbox_to_text(*c["box"], num_bins=c.get("bins", 1024)) == c["expected"], c
    for c in d["direction_cases"]:
        assert V.direction_from_bbox(*c["box"], slope=c["slope"]) == c["expected"], c
        assert V.direction_from_bbox(*c["box"], slope=c["slope"], add_move_prefix=True) == "move " + c["expected"]


def test_vqa_mixture_sizes_are_the_reference_constants():
    import json
    import pathlib

    from lap_amd import vqa_export as V
    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "vqa_v1.json").read_text())
    assert V.VQA_NUM_TRANSITIONS == fx["NUM_TRANSITIONS"] and set(V.VQA_NUM_TRANSITIONS) == set(V.VQA_DATASET_IDS)


def test_vqa_records_become_samples_and_join_a_mixture(setup):
    """Hand-built records in each dataset's TFDS schema -> (trajectory id, prompt, caption) by the rules of the six reader classes, with the
    random draws injected; then a VqaDataset next to an episode store in one mixture, through the loader's transform stack."""
    from lap_amd import vqa_export as V

    first = lambda seed_pair, n: 0
    last = lambda seed_pair, n: n - 1
    hb = lambda text, n: len(text) % n
    img = np.full((6, 8, 3), 7, dtype=np.uint8)
    # VQAv2: the record's own question / answer
    r = {"image": img, "question_id": 42, "image/id": 9, "question_text": b"what color is the cup?", "top_answer": b"red"}
    assert V.prompt_and_caption("vqa", r) == ("what color is the cup?", "red") and V.trajectory_id("vqa", r) == "vqa_42_9"
    assert V.sample_from_record("vqa", dict(r, top_answer=b"")) is None                       # empty answers are filtered
    # COCO: one of the captions, one of the 20 prompts, seeded by the image id
    r = {"image": img, "image/filename": b"a.jpg", "image/id": 3, "captions": {"text": np.array([b"two dogs", b"dogs playing"])}}
    assert V.prompt_and_caption("coco_captions", r, choose=first) == (V.COCO_CAPTION_PROMPTS[0], "two dogs")
    assert V.prompt_and_caption("coco_captions", r, choose=last) == (V.COCO_CAPTION_PROMPTS[-1], "dogs playing")
    assert V.trajectory_id("coco_captions", r) == "coco_a.jpg_3"
    seeds = []
    V.prompt_and_caption("coco_captions", r, seed=11, choose=lambda sp, n: seeds.append(tuple(sp)) or 0)
    assert seeds == [(11, 3), (12, 3)]                                                        # [seed, id] for the caption, [seed + 1, id] for the prompt
    # PixMo-Cap: the caption as it is; id and seed from the FarmHash of the file name
    r = {"image": img, "image_filename": b"xy.png", "caption": b"a long caption"}
    assert V.prompt_and_caption("pixmo_cap", r, choose=last, hash_bucket=hb) == (V.PIXMO_CAP_PROMPTS[-1], "a long caption")
    assert V.trajectory_id("pixmo_cap", r, hb) == f"pixmo_cap_xy.png_{len('xy.png_a long caption') % 2147483647}"
    # PixMo-Points: points on a 0-100 scale, sorted by x then y, as <locY><locX> pairs; at most 20
    r = {"image": img, "image_sha256": b"ab12", "label": b"mugs", "count": 2, "points": {"x": np.array([50.0, 10.04]), "y": np.array([25.0, 99.96])}}
    p, c = V.prompt_and_caption("pixmo_point", r, scale=1.0 * 1, choose=first, hash_bucket=hb)
    assert p == "How many mugs are in the image? Point them out."
    want = lambda x, y: f"<loc{int(round(y / 100 * 1023)):04d}><loc{int(round(x / 100 * 1023)):04d}>"
    assert V.points_to_text(np.array([[50.0, 25.0], [10.0, 100.0]])) == want(10.0, 100.0) + want(50.0, 25.0)
    many = {"x": np.linspace(1, 99, 30), "y": np.linspace(99, 1, 30)}
    assert V.prompt_and_caption("pixmo_point", dict(r, points=many), scale=100.0 / 100.0, choose=first, hash_bucket=hb)[1].count("<loc") == 2 * V.MAX_POINTS
    # LVIS / PACO: a category and its box -> an OXE bbox prompt + loc tokens, or (directional / with direction_prob) a direction
    r = {"image": img, "image_id": b"000123", "annotations": {"category_name": np.array([b"spoon"]), "bbox": np.array([[[0.1, 0.6], [0.3, 0.9]]], dtype=np.float32)}}
    p, c = V.prompt_and_caption("lvis", r, choose=first, hash_bucket=hb, uniform=lambda sp: 0.9)
    assert p == "Pick up the spoon, predict where it is in the image." and c == V.bbox_to_text(np.float32(0.1), np.float32(0.6), np.float32(0.3), np.float32(0.9))
    assert c == "<loc0614><loc0102><loc0921><loc0307>"
    p, c = V.prompt_and_caption("paco_lvis", r, direction_prob=0.5, choose=last, hash_bucket=hb, uniform=lambda sp: 0.2)
    assert p == "".join([V.GENERAL_BBOX_PROMPT_PARTS[-1][0], "spoon", V.GENERAL_BBOX_PROMPT_PARTS[-1][1]]) and c == "move left"
    p, c = V.prompt_and_caption("lvis", r, directional=True, choose=first, hash_bucket=hb)
    assert p.startswith("From the image center") and "spoon" in p and c == "move left"
    assert V.trajectory_id("lvis", r, hb).startswith("lvis_000123_") and V.trajectory_id("paco_ego4d", r, hb).startswith("paco_000123_")
    assert V.is_validation("vqa", {"question_id": 1, "image/id": 2}, 0, 1.0, hb) and not V.is_validation("vqa", {"question_id": 1, "image/id": 2}, 0, 0.0, hb)
    assert V.VQA_DATASET_IDS == {"coco_captions": 1, "lvis": 2, "paco_lvis": 3, "paco_ego4d": 4, "pixmo_cap": 5, "pixmo_point": 6, "vqa": 7}
    # the stand-in draws are pure functions of the seed pair
    assert V.numpy_choices((3, 4), 20) == V.numpy_choices((3, 4), 20) and 0 <= V.numpy_uniform((1, 2)) < 1
    # ---- into a mixture with robot episodes, through the loader
    cfg, tok, eps_ds = setup
    recs = [{"image": np.full((30, 44, 3), 10 * i + 5, dtype=np.uint8), "question_id": i, "image/id": i, "question_text": f"what is item {i}?".encode(), "top_answer": f"thing {i}".encode()}
            for i in range(6)]
    vqa = D.VqaDataset([V.sample_from_record("vqa", r) for r in recs], action_horizon=cfg.model.action_horizon, action_dim=7, state_dim=10)
    s = vqa[2]
    assert s["is_vqa_sample"] and s["vqa_dataset_id"] == 7 and s["caption"] == "thing 2" and s["observation"]["base_0_rgb"].shape == (30, 44, 3)
    mix = D.MixtureDataset({"droid": eps_ds, "vqa": vqa}, [("droid", 1.0), ("vqa", 1.0)], balance_weights=False, seed=0)
    loader = D.create_data_loader(cfg, mix, tok, shuffle=True, seed=1, num_batches=6, split="val")
    n_vqa = n = 0
    for obs, actions in loader:
        isv = np.asarray(obs.is_vqa_sample.cpu()) if hasattr(obs, "is_vqa_sample") and obs.is_vqa_sample is not None else None
        assert isv is not None, "the observation carries the VQA flag compute_loss mixes by"
        n_vqa += int(isv.sum()); n += len(isv)
        assert torch.isfinite(actions).all()        # (a VQA sample's dummy actions are masked out of the action loss, lap.py:560-569)
        # the mixer normalises robot datasets only (dataset_mixer.py:338-341): a VQA sample's all-zero state and actions stay zero
        sel = torch.as_tensor(isv, dtype=torch.bool)
        assert float(actions.cpu()[sel].abs().max() if sel.any() else 0.0) == 0.0 and float(obs.state.cpu()[sel].abs().max() if sel.any() else 0.0) == 0.0
    assert 0 < n_vqa < n
    # frames of different native sizes were brought to the model's resolution the way the reference's decode step does it
    small = np.zeros((30, 60, 3), dtype=np.uint8); small[:, :30] = 200
    out = pio.dataset_resize_with_pad(small, 56, 56)                       # ratio 60 / 56: 28 x 56 rows, 14 rows of padding above and below
    assert out.shape == (56, 56, 3) and (out[:14] == 0).all() and (out[42:] == 0).all() and out[14:42, :20].min() == 200 and out[14:42, 36:].max() == 0
    assert pio.dataset_resize_with_pad(out, 56, 56) is out


def test_global_norm_stats_match_reference_generated_fixture():
    """tests/golden/global_stats_v1.json (make_global_stats_golden.py): the reference's `GlobalStatisticsBuilder` on five datasets of
    different widths and sizes — pooled mean / variance weighted by transitions, bracketing quantiles, states pooled per state type,
    the stateless and the VQA set left out."""
    import json
    import pathlib

    from lap_amd.data import global_norm_stats

    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "global_stats_v1.json").read_text())
    g = global_norm_stats(fx["per_dataset"], action_dim=fx["action_dim"], state_dim=fx["state_dim"], state_types=fx["state_types"], exclude=fx["vqa"])
    assert sorted(g) == sorted(fx["global"])
    for key, ref in fx["global"].items():
        for f in ("mean", "std", "q01", "q99", "min", "max"):
            np.testing.assert_allclose(np.asarray(g[key][f], dtype=np.float64), np.asarray(ref[f]), rtol=2e-6, atol=2e-6, err_msg=f"{key}.{f}")
        assert int(g[key]["num_transitions"]) == ref["num_transitions"] and int(g[key]["num_trajectories"]) == ref["num_trajectories"]


def test_vla0_strategy_labels_come_from_the_normalised_chunk():
    """`transform_strategy="vla0"` (training/config.py:701-751): the reference's mixer normalises the trajectory BEFORE `CoTInputs`, and the
    VLA-0 handler writes the normalised, padded chunk as integers (sample_handlers.py:434-457).  The loader therefore normalises first for
    this strategy: the label text of a batch row must be the format's own summary of that row's normalised actions."""
    import dataclasses

    from lap_amd import lang_actions as la

    base = get_config("debug")
    cfg = dataclasses.replace(base, model=dataclasses.replace(base.model, action_dim=7, max_token_len=400, prompt_format="vla0_chunked"),
                              data=dataclasses.replace(base.data, transform_strategy="vla0", language_action_format_name="vla0_chunked",
                                                       wrist_image_dropout_prob=0.0, random_mask_prob=0.0, enable_diverse_questions=False))
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=400, prompt_format="vla0_chunked")
    ds = D.EpisodeDataset(_episodes(), action_horizon=cfg.model.action_horizon)
    loader = D.create_data_loader(cfg, ds, tok, shuffle=False, num_batches=2)
    obs, actions = next(iter(loader))
    assert float(actions.abs().max()) <= 1.0                      # the training side clips
    for b in range(actions.shape[0]):
        want = la.VLA0_CHUNKED_FORMAT.summarize_actions(actions[b].numpy())
        ids = obs.tokenized_prompt[b][obs.tokenized_langact_mask[b]].tolist()
        assert ids == tok._tokenizer.encode(want, add_eos=True), (b, want[:60])     # (ids: the tiny vocabulary has no piece for every digit)
        assert bool(obs.sample_mask[b])


def test_training_time_image_randomness_follows_the_data_config(setup):
    """training/config.py:336-352: the data config's `wrist_image_dropout_prob` / `random_mask_prob` reach `CoTInputs`.  With both at 1 every
    wrist image is dropped (zeros) and every all-zero image is un-masked; with the defaults (0.1 / 0.2) the rates over many samples are the
    configured ones."""
    import dataclasses
    cfg, tok, _ = setup
    eps = _episodes(2, 12, seed=4)
    for e in eps:
        e["left_wrist_0_rgb"] = np.full_like(e["base_0_rgb"], 7)
    ds = D.EpisodeDataset(eps, action_horizon=cfg.model.action_horizon)
    on = dataclasses.replace(cfg, data=dataclasses.replace(cfg.data, wrist_image_dropout_prob=1.0, random_mask_prob=1.0))
    obs, _ = next(iter(D.create_data_loader(on, ds, tok, shuffle=False, num_batches=1)))
    assert float(obs.images["left_wrist_0_rgb"].abs().max()) == 1.0 and bool((obs.images["left_wrist_0_rgb"] == -1.0).all())     # zeros, scaled to [-1, 1]
    assert bool(obs.image_masks["left_wrist_0_rgb"].all())
    dflt = dataclasses.replace(cfg, batch_size=8, data=dataclasses.replace(cfg.data, wrist_image_dropout_prob=0.1, random_mask_prob=0.2))
    np.random.seed(0)
    dropped = masked_on = total = 0
    for obs, _ in D.create_data_loader(dflt, ds, tok, shuffle=True, seed=2, num_batches=60):
        gone = (obs.images["left_wrist_0_rgb"] == -1.0).flatten(1).all(1)
        dropped += int(gone.sum()); total += gone.numel()
        masked_on += int((obs.image_masks["left_wrist_0_rgb"] & gone).sum())
    assert 0.05 < dropped / total < 0.16 and 0.08 < masked_on / max(dropped, 1) < 0.4, (dropped, masked_on, total)


def test_prediction_samples_and_label_windows_follow_the_reference_rules():
    """base_dataset.py:493-590,603-697 restated (data.EpisodeDataset): the label window is horizon_seconds x control_frequency steps cut at
    the episode's end, with `time_horizon_seconds` = steps used / frequency; with prediction co-training a frame becomes a (now, future)
    pair of ONE camera with probability `pred_prob`, the future m = clamp(int(2.5 f), 1, T - 1) steps ahead, labelled with the movement over
    those m steps (zero-padded: past the end the gripper reads 0), horizon m / f; without a wrist camera the base camera is always used."""
    import dataclasses
    T, f = 40, 5
    eps = _episodes(2, T, seed=3)
    for e in eps:
        e["left_wrist_0_rgb"] = (255 - e["base_0_rgb"]).astype(np.uint8)
        e["control_frequency"] = np.int32(f)
        e["dataset_name"] = "bridge_v2_oxe"            # (no upside-down wrist camera: frames compare as stored)
    plain = D.EpisodeDataset(eps, action_horizon=10)
    s = plain[7]
    assert np.isclose(s["time_horizon_seconds"], 1.0) and not s["is_prediction_sample"]
    np.testing.assert_allclose(s["language_actions"][:3], eps[0]["actions"][7:12, :3].sum(0), rtol=1e-5)      # 1 s x 5 Hz = 5 steps
    assert np.isclose(plain[T - 2]["time_horizon_seconds"], 2 / f) and plain[T - 2]["language_actions"][6] == eps[0]["actions"][T - 1, 6]
    two = D.EpisodeDataset(eps, action_horizon=10, horizon_seconds=(1.0, 2.0))
    assert sorted({round(two[i]["time_horizon_seconds"], 3) for i in range(25)}) == [1.0, 2.0]
    cfg = get_config("lap_cotrain")
    cfg = dataclasses.replace(cfg, data=dataclasses.replace(cfg.data, resize_resolution=None))
    ds = D.episode_dataset_from_config(dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, action_horizon=10)), eps, seed=1)
    assert ds.enable_prediction_training and ds.pred_prob == 0.3 and ds.primary_pred_prob == 0.8
    m = int(2.5 * f)
    pred = [ds[i] for i in range(len(ds)) if ds[i]["is_prediction_sample"]]
    assert 0.2 < len(pred) / len(ds) < 0.42 and 0.6 < np.mean([p["pred_use_primary"] for p in pred]) < 0.95
    for i in range(len(ds)):
        p = ds[i]
        if not p["is_prediction_sample"]:
            continue
        ep, t = divmod(i, T)
        e = eps[ep]
        cam = e["base_0_rgb"] if p["pred_use_primary"] else e["left_wrist_0_rgb"]
        assert np.array_equal(p["observation"]["base_0_rgb"], cam[t]) and np.array_equal(p["observation"]["left_wrist_0_rgb"], cam[min(t + m, T - 1)])
        assert np.isclose(p["time_horizon_seconds"], m / f)
        np.testing.assert_allclose(p["language_actions"][:3], e["actions"][t:t + m, :3].sum(0), rtol=1e-4, atol=1e-6)
        assert p["language_actions"][6] == (e["actions"][t + m - 1, 6] if t + m <= T else 0.0)
        assert np.array_equal(p["actions"], plain[i]["actions"])                      # the action chunk is the frame's own either way
    again = D.episode_dataset_from_config(dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, action_horizon=10)), eps, seed=1)
    assert [ds[i]["is_prediction_sample"] for i in range(20)] == [again[i]["is_prediction_sample"] for i in range(20)]      # a function of (seed, episode, step)
    no_wrist = D.EpisodeDataset([{k: v for k, v in eps[0].items() if k != "left_wrist_0_rgb"}], action_horizon=10, enable_prediction_training=True, pred_prob=1.0)
    q = no_wrist[3]
    assert q["pred_use_primary"] and np.array_equal(q["observation"]["left_wrist_0_rgb"], eps[0]["base_0_rgb"][3 + m])
    # through the loader: a prediction sample keeps both frames and gets the prediction prompt / question
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=200)
    small = dataclasses.replace(get_config("debug"), batch_size=4, model=dataclasses.replace(get_config("debug").model, action_dim=16, max_token_len=200, enable_prediction_training=True))
    pds = D.episode_dataset_from_config(dataclasses.replace(small, data=dataclasses.replace(small.data, pred_prob=1.0, wrist_image_dropout_prob=0.0, random_mask_prob=0.0)), eps)
    obs, _ = next(iter(D.create_data_loader(small, pds, tok, shuffle=False, num_batches=1)))
    assert bool(obs.is_prediction_sample.all()) and bool(obs.image_masks["left_wrist_0_rgb"].all())


def test_control_frequencies_match_the_reference_table():
    """tests/golden/dataset_configs_v1.json (make_dataset_configs_golden.py: `OXE_DATASET_METADATA` of datasets/utils/configs.py read with
    ast.literal_eval): the control frequency of every dataset the exporter standardises."""
    import json
    import pathlib

    from lap_amd import rlds_export as R
    ref = json.loads((pathlib.Path(__file__).parent / "golden" / "dataset_configs_v1.json").read_text())["control_frequency"]
    for name, f in R.CONTROL_FREQUENCY.items():
        assert ref[name] == f, (name, f, ref[name])
    assert set(R.STANDARDIZE) - set(R.CONTROL_FREQUENCY) == {"libero_combined"} and "libero_combined" not in ref


def test_train_val_split_is_per_trajectory_and_salted():
    """base_dataset.py:375-385: a trajectory is a validation trajectory when the bucket of (seed, trajectory id) falls below
    val_fraction x 1000; train and val are complementary, a function of the seed, and whole episodes."""
    eps = _episodes(60, 3, seed=9)
    tr = D.EpisodeDataset(eps, action_horizon=4, split="train", val_fraction=0.25, seed=3)
    va = D.EpisodeDataset(eps, action_horizon=4, split="val", val_fraction=0.25, seed=3)
    assert len(tr.episodes) + len(va.episodes) == 60 and 5 <= len(va.episodes) <= 25
    ids = lambda d: {e["prompt"] for e in d.episodes}
    assert not ids(tr) & ids(va)
    assert ids(va) != ids(D.EpisodeDataset(eps, action_horizon=4, split="val", val_fraction=0.25, seed=4))
    assert len(D.EpisodeDataset(eps, action_horizon=4, split="train", val_fraction=0.0).episodes) == 60
    with pytest.raises(ValueError, match="empty"):
        D.EpisodeDataset(eps, action_horizon=4, split="val", val_fraction=0.0)
    cfg = get_config("lap_libero")          # val_fraction 0.0: everything trains
    assert len(D.episode_dataset_from_config(cfg, eps, split="train").episodes) == 60


def test_wrist_rotation_and_resize_follow_the_dataset_rules():
    """datasets/registry.py:155-163 + image_utils.py:289-377: DROID's wrist camera is upside down — the wrist frame is rotated by 180 degrees
    (after the resize with padding), `rotation_applied` reaches the label code; a prediction pair of the wrist camera rotates both frames, one
    of the base camera none; other datasets are left alone."""
    eps = _episodes(1, 8, seed=2, hw=(20, 32))
    eps[0]["left_wrist_0_rgb"] = np.random.default_rng(0).integers(0, 255, eps[0]["base_0_rgb"].shape, dtype=np.uint8)
    eps[0]["dataset_name"] = "droid"
    ds = D.EpisodeDataset(eps, action_horizon=4, resize_to=(24, 24))
    s = ds[2]
    want = pio.dataset_resize_with_pad(eps[0]["left_wrist_0_rgb"][2], 24, 24)[::-1, ::-1]
    assert s["rotation_applied"] and s["observation"]["left_wrist_0_rgb"].shape == (24, 24, 3) and np.array_equal(s["observation"]["left_wrist_0_rgb"], want)
    assert np.array_equal(s["observation"]["base_0_rgb"], pio.dataset_resize_with_pad(eps[0]["base_0_rgb"][2], 24, 24))
    assert not D.EpisodeDataset(eps, action_horizon=4, not_rotate_wrist_prob=1.0)[2]["rotation_applied"]
    other = D.EpisodeDataset([dict(eps[0], dataset_name="bridge_v2_oxe")], action_horizon=4)[2]
    assert not other["rotation_applied"] and np.array_equal(other["observation"]["left_wrist_0_rgb"], eps[0]["left_wrist_0_rgb"][2])
    pw = D.EpisodeDataset(eps, action_horizon=4, enable_prediction_training=True, pred_prob=1.0, primary_pred_prob=0.0)[1]
    assert pw["is_prediction_sample"] and not pw["pred_use_primary"] and pw["rotation_applied"]
    assert np.array_equal(pw["observation"]["base_0_rgb"], eps[0]["left_wrist_0_rgb"][1][::-1, ::-1])
    pp = D.EpisodeDataset(eps, action_horizon=4, enable_prediction_training=True, pred_prob=1.0, primary_pred_prob=1.0)[1]
    assert pp["pred_use_primary"] and not pp["rotation_applied"] and np.array_equal(pp["observation"]["base_0_rgb"], eps[0]["base_0_rgb"][1])
    assert D.needs_wrist_rotation("berkeley_fanuc_manipulation") and not D.needs_wrist_rotation("taco_play")


def test_droid_episode_options():
    """droid_dataset.py:104-232: three instructions / two exterior cameras drawn per episode, unsuccessful or instruction-less recordings
    dropped, frames outside the keep ranges are not samples."""
    from lap_amd import rlds_export as R
    T = 6
    g = np.random.default_rng(2)
    im = lambda: g.integers(0, 255, (T, 8, 8, 3), dtype=np.uint8)
    cart = np.concatenate([np.arange(T)[:, None] * np.array([[0.0, 0.02, 0.0]]), np.zeros((T, 3))], 1)
    gp = np.linspace(0, 1, T)
    base = {"observation": {"exterior_image_1_left": im(), "exterior_image_2_left": im(), "wrist_image_left": im(), "cartesian_position": cart,
                            "gripper_position": gp},
            "action_dict": {"gripper_position": gp[:, None]}, "language_instruction": b"open the top drawer", "language_instruction_2": b"pull the drawer open",
            "language_instruction_3": b"", "traj_metadata": {"episode_metadata": {"file_path": np.array([b"/x/success/2023/traj.h5"] * T)}}}
    ep = R.episode_from_rlds("droid", base, keep_mask=[1, 1, 0, 0, 1, 1])
    assert list(ep["prompt_alternatives"]) == ["open the top drawer", "pull the drawer open"] and ep["base_0_rgb_alt"].shape == (T, 8, 8, 3)
    assert R.episode_from_rlds("droid", dict(base, traj_metadata={"episode_metadata": {"file_path": np.array([b"/x/failure/traj.h5"] * T)}})) is None
    assert R.episode_from_rlds("droid", dict(base, language_instruction=b"open it")) is None          # <= 10 characters
    ds = D.EpisodeDataset([ep], action_horizon=3)
    assert len(ds) == 4 and np.allclose(ds[2]["actions"][:, 1], [0.02, 0.02, 0.02])                  # frame 4: last-value padding after frame 5
    seen_p, seen_c = set(), set()
    for seed in range(12):
        s = D.EpisodeDataset([ep], action_horizon=3, seed=seed)[0]
        seen_p.add(s["prompt"]); seen_c.add(bool(np.array_equal(s["observation"]["base_0_rgb"], ep["base_0_rgb"][0])))
    assert seen_p == {"open the top drawer", "pull the drawer open"} and seen_c == {True, False}
    a = D.EpisodeDataset([ep], action_horizon=3, seed=5)
    assert a[0]["prompt"] == a[3]["prompt"]                                                             # one draw per episode


def test_droid_keep_mask_from_ranges():
    from lap_amd import rlds_export as R
    traj = {"observation": {"cartesian_position": np.zeros((8, 6))},
            "traj_metadata": {"episode_metadata": {"recording_folderpath": b"gs://x/rec", "file_path": np.array([b"gs://x/success/t.h5"] * 8)}}}
    m = R.droid_keep_mask({"gs://x/rec--gs://x/success/t.h5": [[1, 3], [5, 20]]}, traj)
    assert m.tolist() == [False, True, True, False, False, True, True, True] and not R.droid_keep_mask({}, traj).any()


def test_vqa_store_split_follows_the_same_rule():
    """vqa_base.py:190-202: VQA samples split train / val by the salted hash of their id, like robot trajectories."""
    samples = [{"image": np.zeros((4, 4, 3), np.uint8), "prompt": f"q{i}", "caption": "a", "dataset_name": "coco_captions", "vqa_dataset_id": 1} for i in range(80)]
    tr = D.VqaDataset(samples, action_horizon=4, split="train", val_fraction=0.25, seed=1)
    va = D.VqaDataset(samples, action_horizon=4, split="val", val_fraction=0.25, seed=1)
    assert len(tr) + len(va) == 80 and 8 <= len(va) <= 32 and not {s["prompt"] for s in tr.samples} & {s["prompt"] for s in va.samples}
    assert va.num_transitions == 80340          # the mixture weight stays the reference's constant


def test_exported_droid_episode_runs_through_the_train_loader_with_eef_labels():
    """Exporter -> episode store -> config-driven dataset -> train loader, end to end on a DROID-shaped trajectory: the rot6d state the
    loader derives feeds the end-effector-frame label text (with `rotation_applied` from the rotated wrist camera), the chunk is the
    displacement from the current pose, statistics are over chunks, tokens carry a language-action span."""
    import dataclasses

    from lap_amd import rlds_export as R
    T = 24
    g = np.random.default_rng(5)
    im = lambda: g.integers(1, 255, (T, 30, 40, 3), dtype=np.uint8)
    cart = np.concatenate([np.cumsum(g.normal(scale=0.004, size=(T, 3)), 0) + 0.3, np.cumsum(g.normal(scale=0.01, size=(T, 3)), 0)], 1)
    gp = (np.arange(T) > 12).astype(np.float64)
    traj = {"observation": {"exterior_image_1_left": im(), "wrist_image_left": im(), "cartesian_position": cart, "gripper_position": gp},
            "action_dict": {"gripper_position": gp[:, None]}, "language_instruction": b"put the marker in the cup"}
    ep = R.episode_from_rlds("droid", traj)
    base = get_config("debug")
    cfg = dataclasses.replace(base, batch_size=4, model=dataclasses.replace(base.model, action_dim=16, max_token_len=160),
                              data=dataclasses.replace(base.data, resize_resolution=(56, 56), wrist_image_dropout_prob=0.0, random_mask_prob=0.0))
    ds = D.episode_dataset_from_config(cfg, [ep])
    s = ds[3]
    assert s["rotation_applied"] and s["raw_state"].shape == (10,) and s["observation"]["base_0_rgb"].shape == (56, 56, 3)
    np.testing.assert_allclose(s["actions"][:, :3], cart[4:4 + cfg.model.action_horizon, :3] - cart[3, :3], atol=1e-6)
    out = pio.CoTInputs(action_dim=16)(dict(s))
    assert out["frame_description"] == "end-effector frame" and isinstance(out["language_actions"], str) and out["language_actions"]
    tok = pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=160)
    loader = D.create_data_loader(cfg, ds, tok, shuffle=False, num_batches=2)
    obs, actions = next(iter(loader))
    assert actions.shape == (4, cfg.model.action_horizon, 16) and float(actions.abs().max()) <= 1.0 and bool(obs.tokenized_langact_mask.any())
    assert obs.state.shape == (4, 16) and bool(obs.image_masks["left_wrist_0_rgb"].all())
    stats, kind = loader.get_norm_stats_for_checkpoint()
    assert len(stats["actions"]["q99"]) == 16 and len(stats["state"]["q99"]) == 10


def test_validation_pass_over_a_mixture_is_bounded_by_its_distinct_samples(setup):
    """a mixture's len() is its weighted effective length (transitions x horizon / weight); one validation pass draws at most
    as many samples as the member datasets hold."""
    cfg, tok, eps_ds = setup
    mix = D.MixtureDataset({"a": eps_ds, "b": eps_ds}, [("a", 1.0), ("b", 3.0)], seed=0)
    assert len(mix) > mix.num_distinct_samples == 2 * len(eps_ds)
    loader = D.create_data_loader(cfg, mix, tok, shuffle=False, seed=1, split="val")
    n = sum(1 for _ in loader)
    assert n == max(1, min(len(mix), mix.num_distinct_samples) // loader.batch_size)
