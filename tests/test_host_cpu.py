"""Host-side logic on CPU (no kernel launches): reference <-> engine key map, parameter store geometry,
observation parsing, and the per-token info words / positions against the oracle's boolean mask."""
import numpy as np
import pytest
import torch

from lap_amd import params as PR
from lap_amd.config import get_config
from lap_amd.observation import CoTObservation
from oracle import lap_oracle as O
from tests.common import debug_model_cfg, make_inputs, oracle_cfg, to_observation


def test_reference_engine_key_map_round_trip():
    cfg = debug_model_cfg()
    P = O.init_params(oracle_cfg(cfg), 0)
    E = PR.reference_to_engine(cfg, P)
    P2 = PR.engine_to_reference(cfg, E)
    assert set(P) == set(P2)
    for k in P:
        assert P[k].shape == P2[k].shape and torch.equal(P[k], P2[k]), k
    ps = PR.ParamStore(cfg, device="cpu")
    ps.load_reference_tree(P)
    assert set(ps.names()) == set(E)
    for name in ("llm/0/wqkv0", "img/0/w1", "act/out_w", "ada/w"):
        assert torch.equal(ps._view(ps.master[ps.tensor_unit[name].name], name), E[name]), name
    P3 = ps.to_reference_tree("master")
    assert all(torch.equal(P3[k], P[k]) for k in P)
    with pytest.raises(KeyError):
        ps.load_reference_tree({k: v for k, v in P.items() if "time_mlp_in" not in k})


def test_siglip_mlp_padding_is_invisible_in_the_reference_tree(monkeypatch):
    """An MLP width that is no multiple of 128 (So400m: 4304) is zero-padded inside the engine only: the reference tree goes in
    and comes out with its own shapes, the padding is exactly zero after a load and after the random init, and the random init
    draws the same numbers as the unpadded layout would."""
    import dataclasses
    from lap_amd import config as C
    monkeypatch.setitem(C._SIGLIP, "pad-test/14", C.SiglipConfig(32, 2, 1100, 2))
    cfg = dataclasses.replace(debug_model_cfg(), siglip_variant="pad-test/14")
    assert PR.siglip_mlp_pad(1100) == 1280 and PR.siglip_mlp_pad(128) == 128 and PR.siglip_mlp_pad(4352) == 4352
    shapes = PR.reference_shapes(cfg)
    blk = "PaliGemma/img/Transformer/encoderblock/MlpBlock_0"
    assert shapes[f"{blk}/Dense_0/kernel"] == (2, 32, 1100) and shapes[f"{blk}/Dense_1/kernel"] == (2, 1100, 32) and shapes[f"{blk}/Dense_0/bias"] == (2, 1100)
    g = torch.Generator().manual_seed(0)
    P = {k: torch.randn(sh, generator=g) for k, sh in shapes.items()}
    E = PR.reference_to_engine(cfg, P)
    assert E["img/1/w1"].shape == (1280, 32) and E["img/1/b1"].shape == (1280,) and E["img/1/w2"].shape == (32, 1280)
    assert not E["img/1/w1"][1100:].any() and not E["img/1/b1"][1100:].any() and not E["img/1/w2"][:, 1100:].any()
    P2 = PR.engine_to_reference(cfg, E)
    assert all(torch.equal(P[k], P2[k]) for k in P)
    ps = PR.ParamStore(cfg, device="cpu")
    ps.init_random(3)
    w1, w2, b1 = (ps._view(ps.master[ps.tensor_unit[n].name], n) for n in ("img/0/w1", "img/0/w2", "img/0/b1"))
    assert not w1[1100:].any() and not w2[:, 1100:].any() and not b1[1100:].any() and w1[:1100].abs().min() > 0
    monkeypatch.setenv("LAP_SIGLIP_PAD", "0")
    ps0 = PR.ParamStore(cfg, device="cpu")
    ps0.init_random(3)
    T0, T1 = ps0.to_reference_tree("master"), ps.to_reference_tree("master")
    assert all(torch.equal(T0[k], T1[k]) for k in T0)


def test_lap3b_parameter_count_and_units():
    units = PR.build_specs(get_config("lap_bench").model)
    import math
    n = sum(math.prod(t.valid or t.shape) for u in units for t in u.tensors)     # the reference's extents (the engine pads SigLIP's MLP width)
    assert abs(n - 3.353e9) < 2e6   # 3.353 B parameters
    n_engine = sum(t.numel for u in units for t in u.tensors)
    assert n_engine - n == 27 * (PR.siglip_mlp_pad(4304) - 4304) * (2 * 1152 + 1) and PR.siglip_mlp_pad(4304) == 4352
    names = [u.name for u in units]
    assert names[0] == "small" and names.count("embed") == 1 and sum(x.startswith("llm") for x in names) == 18
    assert sum(x.startswith("img") and x[3:].isdigit() for x in names) == 27
    for ws in (1, 2, 4, 8):   # shard geometry: equal slices, embedding shards on whole vocabulary rows
        ps_geom = PR.ParamStore.__new__(PR.ParamStore)
        ps_geom.world_size, ps_geom.rank = ws, ws - 1
        for u in units:
            pad = PR.ParamStore.padded(ps_geom, u)
            assert pad % ws == 0 and pad >= u.numel
            if u.name == "embed":
                assert (pad // ws) % u.tensors[0].shape[1] == 0


def test_observation_from_dict_accepts_reference_batch():
    B, L = 2, 24
    d = {"image": {"base_0_rgb": np.zeros((B, 56, 56, 3), np.uint8), "left_wrist_0_rgb": np.full((B, 56, 56, 3), 255, np.uint8)},
         "image_mask": {"base_0_rgb": np.ones(B, bool), "left_wrist_0_rgb": np.ones(B, bool)},
         "state": np.zeros((B, 7), np.float32), "tokenized_prompt": np.zeros((B, L), np.int64),
         "tokenized_prompt_mask": np.ones((B, L), bool), "extras": {"cot": {"tokenized_langact_mask": np.zeros((B, L), bool)}},
         "sample_mask": np.ones(B, bool)}
    o = CoTObservation.from_dict(d)
    assert o.images["base_0_rgb"].dtype == torch.float32 and o.images["base_0_rgb"].min() == -1.0
    assert o.images["left_wrist_0_rgb"].max() == 1.0 and o.tokenized_prompt.dtype == torch.int32
    assert o.tokenized_langact_mask is not None and o.sample_mask.dtype == torch.bool
    with pytest.raises(ValueError):
        CoTObservation.from_dict({"image": {}, "tokenized_prompt": np.zeros((1, 2))})


def _mask_from_info(qinfo, kinfo):
    qc, qx, kc, kx = qinfo >> 24, qinfo & 0xFFFFFF, kinfo >> 24, kinfo & 0xFFFFFF
    return ((qc[:, :, None] & kc[:, None, :]) != 0) & (kx[:, None, :] <= qx[:, :, None])


@pytest.mark.parametrize("ragged", [False, True])
def test_info_words_reproduce_reference_masks_and_positions(ragged):
    from lap_amd.model import LAP

    cfg = debug_model_cfg()
    oc = oracle_cfg(cfg)
    obs, actions, noise, time = make_inputs(cfg, B=3, ragged=ragged)
    model = LAP(cfg, seed=0, device="cpu", with_grads=False)
    S = cfg.action_horizon
    qinfo, kinfo, pos = model._train_infos(to_observation(obs, "cpu"), S)
    assert qinfo.dtype == torch.int32 and kinfo.dtype == torch.int32 and pos.dtype == torch.int32
    # oracle: boolean mask + positions of lap.py:303-377
    T = model.n_img_tok
    pm = torch.cat([obs["image_masks"][k][:, None].expand(3, T) for k in cfg.image_keys] + [obs["tokenized_prompt_mask"]], 1)
    ar = torch.cat([torch.zeros(3, 2 * T, dtype=torch.bool), obs["tokenized_langact_mask"]], 1)
    sm = torch.ones(3, S, dtype=torch.bool); sar = torch.zeros(3, S, dtype=torch.bool); sar[:, 0] = True
    _, mask, positions = O.build_masks_positions(oc, obs, pm, ar, sm, sar)
    assert torch.equal(_mask_from_info(qinfo, kinfo), mask)
    assert torch.equal(pos.long(), positions)
    # serving: prefix attends per make_attn_mask(prefix_mask, 0); suffix sees every valid prefix token and all suffix tokens
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"} | {"tokenized_langact_mask": None}
    qp, kp, ppos, qs, kall, pall = model._serve_infos(to_observation(so, "cpu"), S)
    assert torch.equal(_mask_from_info(qp, kp), O.make_attn_mask(pm, torch.zeros_like(pm)))
    full = torch.cat([pm[:, None, :].expand(3, S, -1), torch.ones(3, S, S, dtype=torch.bool)], -1)
    assert torch.equal(_mask_from_info(qs, kall), full)
    assert torch.equal(pall[:, -S:].long(), pm.long().sum(-1)[:, None] + torch.arange(S)[None])


@pytest.mark.parametrize("hole", [False, True])
def test_sample_tokens_right_alignment_is_a_relabelling(hole):
    """The engine's sample_tokens keeps the prefix in place; the reference rolls it to the right edge (lap.py:696-704).
    On the oracle: prefill logits and the first decode step computed WITHOUT the roll, with the decode range mask
    `[prefix_start, prefill_size + step]` rewritten as `seqlen - prefill_len <= j < seqlen`, equal the literal
    restatement - also with a hole in the prefix (masked image), where both formulations attend the same tokens."""
    import math

    cfg = debug_model_cfg()
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=13)
    obs, _, _, _ = make_inputs(cfg, B=3, ragged=True)
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"}
    if not hole:
        so["image_masks"] = {k: torch.ones_like(m) for k, m in so["image_masks"].items()}
    c = {}
    O.sample_tokens(P, oc, so, max_decoding_steps=2, collect=c)
    table = P["PaliGemma/llm/embedder/input_embedding"]
    pt, pm, par = O.embed_prefix(P, oc, so)
    B, size = pm.shape
    ar = torch.arange(size)
    seqlen = (pm.long() * ar).max(-1).values + 1
    plen = pm.long().sum(-1)
    assert hole == bool((seqlen != plen).any())
    (pre, _), cache = O.gemma_forward(P, oc, [pt, None], torch.cumsum(pm.long(), -1) - 1, O.make_attn_mask(pm, par), [None, None])
    lg0 = pre[torch.arange(B), seqlen - 1] @ table.t()
    assert (lg0 - c["logit/0"]).abs().max() < 1e-4
    emb = table[lg0.argmax(-1)][:, None] * math.sqrt(oc.vlm.width)
    in_range = (ar[None] >= (seqlen - plen)[:, None]) & (ar[None] < seqlen[:, None])
    mask = torch.cat([in_range, torch.ones(B, 1, dtype=torch.bool)], 1)[:, None, :]
    (pre1, _), _ = O.gemma_forward(P, oc, [emb, None], plen[:, None], mask, [None, None], kv_cache=cache)
    assert ((pre1 @ table.t())[:, 0] - c["logit/1"]).abs().max() < 1e-4


def test_config_registry_and_defaults_match_reference_source_fixture():
    """tests/golden/train_configs_v1.json (make_train_configs_golden.py: keyword literals of the reference's `_CONFIGS` and `LAPConfig`
    field defaults, read from the parsed source).  Every named config built here must carry exactly the reference's overrides; the
    names not built are the Gemma-3 / FAST-tokenizer / VLA-0 variants (out of scope) and nothing else."""
    import dataclasses
    import json
    import pathlib

    from lap_amd import config as C

    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "train_configs_v1.json").read_text())
    mine = {f.name: (f.default if f.default is not dataclasses.MISSING else None) for f in dataclasses.fields(C.LAPConfig)}
    for k, v in fx["lap_config_defaults"].items():
        assert k in mine and mine[k] == v, (k, v, mine.get(k))

    def check(name, ref, obj, path=""):
        for k, v in ref.items():
            if k in ("__call__", "name"):
                continue
            assert hasattr(obj, k), f"{name}{path}.{k} missing here"
            m = getattr(obj, k)
            if isinstance(v, dict) and "__call__" in v:
                check(name, v, m, f"{path}.{k}")
            else:
                assert getattr(m, "value", m) == v, (name, path, k, v, m)

    built, absent = [], []
    for name, ref in fx["registry"].items():
        try:
            obj = C.get_config(name)
        except ValueError:
            absent.append(name)
            continue
        built.append(name)
        check(name, ref, obj)
    assert sorted(built) == ["lap", "lap_cotrain", "lap_libero", "pi0_replicated", "vla0_replicated", "vla0_replicated_libero"]
    for name in absent:
        m = fx["registry"][name].get("model", {})
        assert "gemma3" in str(m.get("paligemma_variant", "")) or m.get("use_fast"), name
    # field defaults of the data config and of TrainConfig: every field carried here has the reference's default
    for cls, ref in ((C.RLDSDataConfig, fx["data_config_defaults"]), (C.TrainConfig, fx["train_config_defaults"])):
        for f in dataclasses.fields(cls):
            if f.name in ref and not isinstance(ref[f.name], dict):
                want = tuple(ref[f.name]) if isinstance(f.default, tuple) else ref[f.name]
                assert getattr(f.default, "value", f.default) == want, (cls.__name__, f.name, f.default, ref[f.name])
    carried = {f.name for f in dataclasses.fields(C.RLDSDataConfig)}
    for must in ("wrist_image_dropout_prob", "random_mask_prob", "random_base_prob", "use_rough_scale", "language_action_format_name",
                 "transform_strategy", "enable_diverse_questions", "val_fraction", "data_mix", "action_proprio_normalization_type"):
        assert must in carried and must in fx["data_config_defaults"]


def test_reference_tree_without_action_training_has_no_action_expert():
    """lap.py:40-74: with `enable_action_training=False` (the vla0_* configs) the reference builds `gemma.Module(configs=[paligemma])` and none of
    the action / time projections; the exported `params` item must be that tree, and such a tree must load (the engine's unused expert
    tensors become zeros)."""
    import dataclasses

    from lap_amd.params import _is_action_expert_key, engine_to_reference, reference_shapes, reference_to_engine

    cfg = get_config("debug").model
    full = reference_shapes(cfg)
    expert = {k for k in full if _is_action_expert_key(k)}
    assert len(expert) == 19 and all(k.startswith(("PaliGemma/llm/", "action_", "time_mlp_")) for k in expert)
    assert not any("img" in k for k in expert)
    off = dataclasses.replace(cfg, enable_action_training=False)
    sh = reference_shapes(off)
    assert set(sh) == set(full) - expert
    g = torch.Generator().manual_seed(0)
    P = {k: torch.randn(v, generator=g) for k, v in sh.items()}
    E = reference_to_engine(off, P)
    back = engine_to_reference(off, E)
    assert set(back) == set(P) and all(torch.equal(back[k], P[k]) for k in P)
    assert float(E["ada/w"].abs().max()) == 0.0 and float(E["llm/0/wqkv1"].abs().max()) == 0.0 and float(E["act/in_w"].abs().max()) == 0.0
    assert get_config("vla0_replicated").model.enable_action_training is False


def test_pi0_parameter_tree_and_oracle_suffix():
    """`pi05=False` (lap.py:46-61): the reference's tree has `state_proj`, `action_time_mlp_in` (2 w -> w), `action_time_mlp_out`, plain
    `..._norm_1/scale` arrays for the action expert and NO adaRMS Dense layers / `time_mlp_*`; the key map round-trips it, the engine
    has no adaRMS unit, and the oracle's suffix is [state | actions] with two autoregressive blocks (the action tokens see the state
    token, the state token does not see them; the prefix sees neither)."""
    import dataclasses

    from lap_amd.config import LAPConfig
    from lap_amd.params import build_specs, engine_sources, engine_to_reference, reference_shapes, reference_to_engine
    from oracle import lap_oracle as O
    from tests.common import make_inputs, oracle_cfg

    assert LAPConfig(pi05=False, max_token_len=None).max_token_len == 48 and LAPConfig(max_token_len=None).max_token_len == 200   # lap_config.py:78
    assert LAPConfig(pi05=False, discrete_state_input=None).discrete_state_input is False                                         # lap_config.py:80
    cfg = dataclasses.replace(get_config("debug").model, pi05=False, enable_action_training=True)
    sh = reference_shapes(cfg)
    w = O.GEMMA[cfg.action_expert_variant].width
    assert sh["state_proj/kernel"] == (cfg.action_dim, w) and sh["action_time_mlp_in/kernel"] == (2 * w, w) and sh["action_time_mlp_out/kernel"] == (w, w)
    assert "time_mlp_in/kernel" not in sh and not any("Dense_0" in k for k in sh if k.startswith("PaliGemma/llm/"))
    assert sh["PaliGemma/llm/final_norm_1/scale"] == (w,) and sh["PaliGemma/llm/layers/pre_ffw_norm_1/scale"][1] == w
    units = build_specs(cfg)
    assert "ada" not in [u.name for u in units]
    assert set(engine_sources(cfg)) == {t.name for u in units for t in u.tensors}
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=3)
    assert {k: tuple(v.shape) for k, v in P.items()} == sh
    back = engine_to_reference(cfg, reference_to_engine(cfg, P))
    assert set(back) == set(P) and all(torch.equal(back[k], P[k]) for k in P)
    # the suffix and its mask
    obs, actions, noise, time = make_inputs(cfg, B=2, ragged=True)
    S = cfg.action_horizon
    tok, smask, ar, cond = O.embed_suffix(P, oc, noise, time, state=obs["state"])
    assert tok.shape == (2, S + 1, w) and cond is None and ar.tolist() == [True, True] + [False] * (S - 1) and bool(smask.all())
    col = {}
    loss, m = O.compute_loss(P, oc, obs, actions, noise, time, collect=col)
    mask, pos = col["mask"], col["positions"]
    Pn = mask.shape[1] - (S + 1)
    assert not mask[:, :Pn, Pn:].any()                       # prefix rows never see the suffix
    assert not mask[:, Pn, Pn + 1:].any() and mask[:, Pn, Pn].all()          # the state token sees itself, not the actions
    assert mask[:, Pn + 1:, Pn:].all()                       # the action tokens see the state token and each other
    assert (pos[:, Pn + 1:] - pos[:, Pn:-1] == 1).all()      # suffix positions count on from the state token
    assert torch.isfinite(loss) and m["v_t"].shape == (2, S, cfg.action_dim)
    # a different state moves the loss (it reaches the action tokens through attention only)
    obs2 = dict(obs, state=obs["state"] + 0.5)
    assert abs(float(O.compute_loss(P, oc, obs2, actions, noise, time)[0]) - float(loss)) > 1e-7
