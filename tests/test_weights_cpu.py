"""Weight loading into a train state (scripts/train.py:157-187,225-240,248-310; weight_loaders.py:55-105,184-189,691-719),
the trainable / frozen partition, the Orbax converter's tree handling, the sharded parameter norm and the rank-0
checkpoint-directory decision — all on CPU (ParamStore works on any torch device; kernels are not called)."""
import dataclasses
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lap_amd import checkpoints as ck
from lap_amd.config import PathFilter, WeightLoaderChoice, get_config
from lap_amd.params import ParamStore, engine_sources, engine_to_reference, reference_shapes
from lap_amd.train import load_weights, validate_loaded_params
from oracle import lap_oracle as O
from tests.common import oracle_cfg


def _cfg():
    return get_config("debug")


def test_reference_shapes_match_the_oracle_tree():
    cfg = _cfg().model
    P = O.init_params(oracle_cfg(cfg), seed=0)
    sh = reference_shapes(cfg)
    assert set(sh) == set(P) and all(tuple(P[k].shape) == sh[k] for k in P)
    src = engine_sources(cfg)
    assert {k for v in src.values() for k in v} == set(P)           # every reference array lands in exactly the engine tensors listed


def test_validate_loaded_params_semantics():
    cfg = _cfg().model
    exp = reference_shapes(cfg)
    P = O.init_params(oracle_cfg(cfg), seed=1)
    assert validate_loaded_params(exp, dict(P), allow_partial=False) is not None
    with pytest.raises(ValueError, match="unexpected keys"):
        validate_loaded_params(exp, dict(P, **{"PaliGemma/llm/lora_a": torch.zeros(1)}), allow_partial=True)
    bad = dict(P); bad["action_in_proj/bias"] = torch.zeros(3)
    with pytest.raises(ValueError, match="shape"):
        validate_loaded_params(exp, bad, allow_partial=True)
    bad = dict(P); bad["action_in_proj/bias"] = torch.zeros_like(P["action_in_proj/bias"]).to(torch.int32)
    with pytest.raises(ValueError, match="dtype"):
        validate_loaded_params(exp, bad, allow_partial=True)
    part = {k: v for k, v in P.items() if "time_mlp" not in k}
    with pytest.raises(ValueError, match="missing required keys"):
        validate_loaded_params(exp, part, allow_partial=False)
    assert len(validate_loaded_params(exp, part, allow_partial=True)) == len(P) - 4


def _write_params(path, tree, value_suffix=False):
    path.mkdir(parents=True)
    ck._save_tensors(path / "params.safetensors", {"params/" + k + ("/value" if value_suffix else ""): v for k, v in tree.items()})


@pytest.mark.parametrize("value_suffix", [False, True])
def test_weight_loader_merges_checkpoint_over_init(tmp_path, value_suffix):
    """lap_libero-style fine-tuning: `weight_loader=checkpoint(<params dir>)` replaces the random init; a partial
    checkpoint (here: without the action head + with an extra key the model does not know) keeps the init for what it
    lacks, and `allow_partial_weights=False` turns that into an error."""
    tc = _cfg()
    cfg = tc.model
    P = O.init_params(oracle_cfg(cfg), seed=4)
    _write_params(tmp_path / "full" / "params", P, value_suffix)
    tc_full = dataclasses.replace(tc, weight_loader=WeightLoaderChoice("checkpoint", str(tmp_path / "full" / "params")))
    ps = ParamStore(cfg, "cpu")
    ps.init_random(0)
    assert load_weights(tc_full, ps)
    back = ps.to_reference_tree("master")
    assert all(torch.equal(back[k], P[k]) for k in P)
    # partial + foreign key
    part = {k: v for k, v in P.items() if not k.startswith("action_")}
    part["PaliGemma/llm/layers/attn/q_einsum/lora_a"] = torch.zeros(2, 2)        # dropped like CheckpointWeightLoader._merge_params does
    _write_params(tmp_path / "part" / "params", part, value_suffix)
    tc_part = dataclasses.replace(tc, weight_loader=WeightLoaderChoice("checkpoint", str(tmp_path / "part" / "params")))
    ps2 = ParamStore(cfg, "cpu")
    ps2.init_random(0)
    init = ps2.to_reference_tree("master")
    load_weights(tc_part, ps2)
    back = ps2.to_reference_tree("master")
    for k in P:
        assert torch.equal(back[k], init[k] if k.startswith("action_") else P[k]), k
    with pytest.raises(ValueError, match="missing required keys"):
        load_weights(dataclasses.replace(tc_part, allow_partial_weights=False), ParamStore(cfg, "cpu"))
    with pytest.raises(NotImplementedError):
        load_weights(dataclasses.replace(tc, weight_loader=WeightLoaderChoice("gemma3", "x")), ps)
    assert not load_weights(tc, ps)                                              # kind "none" (the debug config says so explicitly)
    assert get_config("lap_libero").weight_loader.kind == "checkpoint"          # training/config.py:776-779


def test_paligemma_weight_loader(tmp_path, monkeypatch):
    """PaliGemmaWeightLoader (weight_loaders.py:109-124) is the reference's DEFAULT loader (:648-654) and what its `lap` config
    trains from: a flat `.npz` whose `params/...` subtree lands under `PaliGemma/`, everything the file lacks (action expert,
    heads) keeps its init whatever `allow_partial_weights` says (missing_regex ".*"), foreign keys are dropped, and a missing
    file is an error — never a silent random init."""
    assert WeightLoaderChoice().kind == "paligemma" and get_config("lap").weight_loader.kind == "paligemma"
    tc = _cfg()
    cfg = tc.model
    P = O.init_params(oracle_cfg(cfg), seed=9)
    # what the official checkpoint holds: the VLM (expert 0) and the image tower, bf16 / f32 mixed, plus keys we do not have
    pg = {k: v for k, v in P.items() if k.startswith("PaliGemma/") and "_1/" not in k and not k.endswith("_1")}
    assert 0 < len(pg) < len(P)
    flat = {"params/" + k[len("PaliGemma/"):]: v.numpy() for k, v in pg.items()}
    flat["params/img/head_extra/kernel"] = np.zeros((2, 2), np.float32)          # not in the model: dropped by _merge_params
    flat["opt_state/count"] = np.zeros((), np.int32)                              # outside `params`: never looked at
    some = next(k for k in flat if k.endswith("q_einsum/w"))
    flat[some] = flat[some].astype(np.float16)                                    # dtype follows the model's (cast on merge)
    path = tmp_path / "pt_224.npz"
    np.savez(path, **flat)
    for choice, env in ((WeightLoaderChoice("paligemma", str(path)), None), (WeightLoaderChoice("paligemma"), str(path))):
        if env:
            monkeypatch.setenv("LAP_PALIGEMMA_NPZ", env)
        ps = ParamStore(cfg, "cpu")
        ps.init_random(0)
        init = ps.to_reference_tree("master")
        assert load_weights(dataclasses.replace(tc, weight_loader=choice, allow_partial_weights=False), ps)
        back = ps.to_reference_tree("master")
        for k in P:
            want = (pg[k].to(torch.float16).float() if "params/" + k[len("PaliGemma/"):] == some else pg[k]) if k in pg else init[k]
            assert torch.equal(back[k], want), k
    monkeypatch.delenv("LAP_PALIGEMMA_NPZ")
    monkeypatch.setenv("OPENPI_DATA_HOME", str(tmp_path / "nothing_here"))
    with pytest.raises(FileNotFoundError, match="pt_224.npz"):
        load_weights(dataclasses.replace(tc, weight_loader=WeightLoaderChoice("paligemma")), ParamStore(cfg, "cpu"))
    bad = dict(flat); bad[some] = np.zeros((3, 3), np.float32)
    np.savez(tmp_path / "bad.npz", **bad)
    with pytest.raises(ValueError, match="shape"):
        load_weights(dataclasses.replace(tc, weight_loader=WeightLoaderChoice("paligemma", str(tmp_path / "bad.npz"))), ParamStore(cfg, "cpu"))


def test_freeze_filter_partitions_the_store():
    """get_vlm_freeze_filter (lap_config.py:171-191): VLM + image encoder frozen, action expert + heads trainable."""
    tc = _cfg()
    cfg = tc.model
    ps = ParamStore(cfg, "cpu")
    ps.init_random(0)
    before = {u.name: ps.master[u.name].clone() for u in ps.units}
    tc = dataclasses.replace(tc, freeze_filter=cfg.get_vlm_freeze_filter())
    ps.set_frozen(tc.is_frozen)
    assert not ps.is_trainable("llm/0/wqkv0") and ps.is_trainable("llm/0/wqkv1") and not ps.is_trainable("img/0/w1")
    assert ps.is_trainable("ada/w") and ps.is_trainable("act/in_w") and not ps.is_trainable("llm/embed") and not ps.is_trainable("llm/1/n_ffw")
    assert tc.trainable_filter("action_out_proj/kernel") and not tc.trainable_filter("PaliGemma/img/head/kernel")
    covered = 0
    for u in ps.units:
        ranges = ps.local_train_ranges(u)
        assert all(a % 64 == 0 and b % 64 == 0 and a < b for a, b in ranges)
        inside = torch.zeros(ps.padded(u), dtype=torch.bool)
        for a, b in ranges:
            inside[a:b] = True
        for t in u.tensors:
            sl = inside[t.offset:t.offset + t.numel]
            assert bool(sl.all()) == ps.is_trainable(t.name) and bool(sl.any()) == ps.is_trainable(t.name), t.name
            v, w = ps.master[u.name][t.offset:t.offset + t.numel], before[u.name][t.offset:t.offset + t.numel]
            if ps.is_trainable(t.name):
                assert torch.equal(v, w)
            else:                                       # frozen parameters are stored at bf16 precision (train.py:225-231)
                assert torch.equal(v, w.to(torch.bfloat16).float())
        covered += sum(b - a for a, b in ranges)
        assert ps.unit_trainable(u) == bool(ranges)
    assert 0 < covered < sum(ps.padded(u) for u in ps.units)
    with pytest.raises(ValueError, match="splits engine tensor"):
        ps.set_frozen(PathFilter(all_of=(".*kv_einsum/w",)))     # q and kv projections share one packed engine tensor
    ps.set_frozen(None)
    assert all(ps.local_train_ranges(u) == [(0, ps.padded(u))] for u in ps.units)
    # a plain regex string works like nnx_utils.PathRegex (full match)
    assert dataclasses.replace(tc, freeze_filter=".*img.*").is_frozen("PaliGemma/img/head/bias")
    assert not dataclasses.replace(tc, freeze_filter="img").is_frozen("PaliGemma/img/head/bias")


def test_orbax_converter_tree_handling(tmp_path):
    """tools/convert_orbax_checkpoint.py without Orbax: nested nnx.State-style tree with trailing 'value' keys (and the
    {'params': ...} wrapper), bf16 leaves, validation against the model, and the written file loads as a weight source."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import convert_orbax_checkpoint as conv

    tc = _cfg()
    cfg = tc.model
    P = O.init_params(oracle_cfg(cfg), seed=6)
    nested = {}
    for k, v in P.items():
        node = nested.setdefault("params", {})
        parts = k.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = {"value": v.numpy()}
    flat = conv.flatten_tree(nested)
    assert set(flat) == set(P)
    assert set(conv.flatten_tree({"a": {"b": np.zeros(1)}, "c": np.ones(2)})) == {"a/b", "c"}       # no 'value' keys: untouched
    f = conv.write_params(flat, tmp_path / "conv", config="debug")
    tree = ck.restore_params(tmp_path / "conv")
    assert all(torch.equal(tree[k], P[k]) for k in P) and f.name == "params.safetensors"
    with pytest.raises(ValueError, match="missing required keys"):
        conv.write_params({k: v for k, v in flat.items() if "embedder" not in k}, tmp_path / "conv2", config="debug")
    ps = ParamStore(cfg, "cpu")
    ps.init_random(0)
    load_weights(dataclasses.replace(tc, weight_loader=WeightLoaderChoice("checkpoint", str(tmp_path / "conv"))), ps)
    assert torch.equal(ps.to_reference_tree("master")["time_mlp_in/kernel"], P["time_mlp_in/kernel"])


# ------------------------------------------------------------------------------------------ world-2 (gloo) pieces
def _w2(rank, world, port, ret, tmp, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lap_amd.fsdp import FsdpComm
        from lap_amd.train import _is_kernel_param

        cfg = get_config("debug").model
        P = O.init_params(oracle_cfg(cfg), seed=2)
        if mode == "param_norm":
            ps = ParamStore(cfg, "cpu", world_size=world, rank=rank)
            ps.load_reference_tree(P)
            got = FsdpComm(ps).param_sumsq(_is_kernel_param).sqrt().item()
            ref = ParamStore(cfg, "cpu")
            ref.load_reference_tree(P)
            eng = {t.name: ref.f32(t.name) for u in ref.units for t in u.tensors}
            want = torch.sqrt(sum((v.double() ** 2).sum() for k, v in eng.items() if _is_kernel_param(k, v.shape))).item()
            assert abs(got - want) / want < 1e-6, (got, want)
            ret[rank] = "ok"
        else:   # the checkpoint-directory decision of train.main, exercised through its own code path
            import lap_amd.train as T

            seen = {}

            def fake_init(config, **kw):
                seen["resume"] = kw["resume"]
                raise KeyboardInterrupt          # stop main() right after the decision

            T_init, T.init_train_state = T.init_train_state, fake_init
            tc = dataclasses.replace(get_config("debug"), exp_name="x", checkpoint_base_dir=tmp, **mode)
            try:
                T.main(tc, device="cpu", log=lambda *_: None)
            except KeyboardInterrupt:
                pass
            finally:
                T.init_train_state = T_init
            ret[rank] = ("ok", seen["resume"])
    except BaseException:  # noqa: BLE001
        import traceback

        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run2(tmp, mode):
    world = 2
    port = 30700 + os.getpid() % 400
    mgr = mp.get_context("spawn").Manager()   # (a forked manager next to an initialised HIP runtime is not safe)
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_w2, args=(r, world, port, ret, str(tmp), mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    return dict(ret)


def test_param_norm_is_correct_under_sharding():
    assert _run2("", "param_norm") == {0: "ok", 1: "ok"}


def test_checkpoint_dir_is_decided_on_rank0_only(tmp_path):
    """every rank used to call initialize_checkpoint_dir itself.  (a) resume=False on a fresh directory must
    not raise FileExistsError on the late rank; (b) overwrite=True over a committed step must give resuming=False on
    BOTH ranks; (c) a plain resume sees the committed step on both."""
    base = tmp_path / "ckpts"
    r = _run2(base, dict(resume=False, overwrite=False))
    assert r == {0: ("ok", False), 1: ("ok", False)}, r
    step = base / "debug" / "x" / "7"
    step.mkdir(parents=True)
    (step / "_COMMITTED").write_text("7")
    r = _run2(base, dict(resume=True, overwrite=False))
    assert r == {0: ("ok", True), 1: ("ok", True)}, r
    r = _run2(base, dict(resume=True, overwrite=True))
    assert r == {0: ("ok", False), 1: ("ok", False)}, r
    assert not step.exists()
    (base / "debug" / "x").mkdir(parents=True, exist_ok=True)
    r = _run2(base, dict(resume=False, overwrite=False))       # exists, neither resume nor overwrite: the error reaches every rank
    assert all("already exists" in str(v) for v in r.values()), r
