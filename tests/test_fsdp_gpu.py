"""FSDP engine equivalence on ONE GPU: two ranks (gloo backend, both on cuda:0) running the sharded train step on
half-batches must reproduce the single-rank step on the concatenated batch (loss, gradient norm, updated params)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _batch(cfg, B):
    from tests.common import make_inputs

    return make_inputs(cfg, B=B, ragged=True, seed=3)


def _slice_obs(obs, sl):
    out = {}
    for k, v in obs.items():
        out[k] = {kk: vv[sl] for kk, vv in v.items()} if isinstance(v, dict) else (v[sl] if v is not None else None)
    return out


def _debug_tc(pi05):
    import dataclasses

    from lap_amd.config import get_config

    tc = get_config("debug")
    return tc if pi05 else dataclasses.replace(tc, model=dataclasses.replace(tc.model, pi05=False))


def _worker(rank, world, port, ret, pi05=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lap_amd.config import get_config
        from lap_amd.train import TrainingStepRunner, init_train_state
        from oracle import lap_oracle as O
        from tests.common import oracle_cfg, to_observation

        dev = "cuda:0"
        tc = _debug_tc(pi05)
        cfg = tc.model
        P = O.init_params(oracle_cfg(cfg), seed=9)
        obs, actions, noise, time = _batch(cfg, 4)
        sl = slice(2 * rank, 2 * rank + 2)
        state = init_train_state(tc, params=P, device=dev, world_size=world, rank=rank, use_fsdp=True)
        runner = TrainingStepRunner(tc)
        infos = []
        for step in range(2):  # second step exercises the all-gathered bf16 mirrors
            state, info = runner(0, state, (to_observation(_slice_obs(obs, sl), dev), actions[sl].to(dev)), step,
                                 noise=noise[sl].to(dev), time=time[sl].to(dev))
            infos.append((info["loss"].item(), info["grad_norm"].item()))
        torch.cuda.synchronize()
        tree = state.model.ps.to_reference_tree("master")
        ret[rank] = ("ok", infos, {k: v for k, v in tree.items() if k in ("PaliGemma/llm/layers/mlp/linear", "PaliGemma/img/head/kernel",
                                                                           "PaliGemma/llm/embedder/input_embedding", "action_out_proj/kernel")})
    except Exception:  # noqa: BLE001
        import traceback

        ret[rank] = ("err", traceback.format_exc(), None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pi05", [True, False])
def test_two_rank_step_equals_single_rank(hip, pi05):
    """(pi05=False: the pi0 parameter tree — no adaRMS unit in the sharded schedule, its extra tensors in the replicated unit)"""
    from lap_amd.config import get_config
    from lap_amd.train import TrainingStepRunner, init_train_state
    from oracle import lap_oracle as O
    from tests.common import oracle_cfg, rel, to_observation

    world = 2
    port = 29600 + os.getpid() % 300
    mgr = mp.get_context("spawn").Manager()   # (a forked manager next to an initialised HIP runtime is not safe)
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, world, port, ret, pi05)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(ret.get(r, ("missing",))[0] == "ok" for r in range(world)), {r: ret.get(r, ("missing",))[:2] for r in range(world)}
    # single rank on the full batch
    tc = _debug_tc(pi05)
    cfg = tc.model
    P = O.init_params(oracle_cfg(cfg), seed=9)
    obs, actions, noise, time = _batch(cfg, 4)
    state = init_train_state(tc, params=P, device="cuda:0")
    runner = TrainingStepRunner(tc)
    ref_infos = []
    for step in range(2):
        state, info = runner(0, state, (to_observation(obs, "cuda:0"), actions.cuda()), step, noise=noise.cuda(), time=time.cuda())
        ref_infos.append((info["loss"].item(), info["grad_norm"].item()))
    ref_tree = state.model.ps.to_reference_tree("master")
    for r in range(world):
        _, infos, tree = ret[r]
        for (l, g), (lr, gr) in zip(infos, ref_infos):
            assert abs(l - lr) / abs(lr) < 2e-3 and abs(g - gr) / gr < 2e-2, (infos, ref_infos)
        for k, v in tree.items():
            upd, upd_ref = v - P[k], ref_tree[k] - P[k]
            assert rel(upd, upd_ref) < 0.1, (k, rel(upd, upd_ref))  # Adam's first steps are sign-like: compare updates loosely
            assert rel(v, ref_tree[k]) < 1e-3, k
    assert torch.equal(ret[0][2]["action_out_proj/kernel"], ret[1][2]["action_out_proj/kernel"])  # replicated unit stays in sync


def _full_size_tc():
    import dataclasses

    from lap_amd.config import CosineDecaySchedule, get_config

    # (a visible learning rate: lap_bench's warm-up starts at 5e-8, below f32 resolution of most weights)
    return dataclasses.replace(get_config("lap_bench"), lr_schedule=CosineDecaySchedule(warmup_steps=0, peak_lr=1e-4, decay_steps=10, decay_lr=1e-4))


_STRIDE = 997


def _full_size_worker(rank, world, port, ret, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lap_amd.train import TrainingStepRunner, init_train_state
        from tests.common import to_observation

        dev = "cuda:0"
        tc = _full_size_tc()
        cfg = tc.model
        obs, actions, noise, time = _batch(cfg, 4)
        sl = slice(2 * rank, 2 * rank + 2)
        state = init_train_state(tc, seed=5, device=dev, world_size=world, rank=rank, use_fsdp=True)
        ps = state.model.ps
        init = {u.name: ps.master[u.name][::_STRIDE].cpu() for u in ps.units}
        runner = TrainingStepRunner(tc)
        infos = []
        for step in range(2):  # second step: the all-gathered bf16 mirrors (+ the embedding table's lo plane) of step 1's update
            state, info = runner(0, state, (to_observation(_slice_obs(obs, sl), dev), actions[sl].to(dev)), step,
                                 noise=noise[sl].to(dev), time=time[sl].to(dev))
            infos.append((info["loss"].item(), info["grad_norm"].item()))
        torch.cuda.synchronize()
        state.model.comm.flush()
        torch.cuda.synchronize()
        fin = {u.name: ps.master[u.name][::_STRIDE].cpu() for u in ps.units}
        mirror = {u.name: (ps.full16[u.name].double().sum().item(), ps.lo16[u.name].double().abs().sum().item() if u.name in ps.lo16 else 0.0)
                  for u in ps.units if u.big}
        torch.save({"init": init, "fin": fin, "ranges": {u.name: ps.shard_range(u) for u in ps.units}}, os.path.join(out_dir, f"rank{rank}.pt"))
        ret[rank] = ("ok", infos, mirror)
    except Exception:  # noqa: BLE001
        import traceback

        ret[rank] = ("err", traceback.format_exc(), None)
    finally:
        dist.destroy_process_group()


def test_full_size_two_rank_step_equals_single_rank(hip, tmp_path):
    """VERDICT r5 next #4c: the LAP-3B shard geometry — embedding table with its hi / lo planes and f32 gradients, bf16 reduce-scatter
    buffers of 45 big units, the adaRMS bank — through a REAL two-process step (gloo, both ranks on this box's one GPU, B = 2 per rank,
    two steps so that the second forward runs on all-gathered mirrors) against the single-rank step on the concatenated batch:
    losses and gradient norms, every unit's master slice (sampled with stride 997: the initial values exactly — same init, right slice —,
    the two-step update loosely, Adam's first steps being sign-like), the gathered mirrors identical on both ranks."""
    from lap_amd.train import TrainingStepRunner, init_train_state
    from tests.common import rel, to_observation

    world = 2
    port = 29700 + os.getpid() % 200
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_full_size_worker, args=(r, world, port, ret, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1200)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(ret.get(r, ("missing",))[0] == "ok" for r in range(world)), {r: ret.get(r, ("missing",))[:2] for r in range(world)}
    tc = _full_size_tc()
    cfg = tc.model
    obs, actions, noise, time = _batch(cfg, 4)
    state = init_train_state(tc, seed=5, device="cuda:0")
    ps = state.model.ps
    init = {u.name: ps.master[u.name].clone() for u in ps.units}          # (13 GB of HBM: cheaper than a host copy)
    runner = TrainingStepRunner(tc)
    ref_infos = []
    for step in range(2):
        state, info = runner(0, state, (to_observation(obs, "cuda:0"), actions.cuda()), step, noise=noise.cuda(), time=time.cuda())
        ref_infos.append((info["loss"].item(), info["grad_norm"].item()))
    torch.cuda.synchronize()
    state.model.comm.flush()
    torch.cuda.synchronize()
    moved = 0
    for r in range(world):
        _, infos, _ = ret[r]
        for (l, g), (lr, gr) in zip(infos, ref_infos):
            assert abs(l - lr) / abs(lr) < 2e-3 and abs(g - gr) / gr < 2e-2, (infos, ref_infos)
        d = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        for u in ps.units:
            a, b = d["ranges"][u.name]
            n = ps.master[u.name].numel()
            a, b = (0, n) if not u.big else (a, min(b, n))        # (a sharded unit is padded to a multiple of the world size)
            i0 = init[u.name][a:b][::_STRIDE].cpu()
            f0 = ps.master[u.name][a:b][::_STRIDE].cpu()
            ri, rf = d["init"][u.name][:i0.numel()], d["fin"][u.name][:f0.numel()]
            assert torch.equal(ri, i0), (u.name, r)               # same random init, and this rank holds the right slice of it
            du, dr = rf - ri, f0 - i0
            if dr.abs().max() > 0:
                moved += 1
                assert rel(du, dr) < 0.15, (u.name, r, rel(du, dr))
                if ri.norm() > 100 * dr.norm():          # (the zero-initialised adaRMS bank IS its update)
                    assert rel(rf, f0) < 1e-3, (u.name, r)
    assert moved >= len(ps.units)
    for name, (fsum, losum) in ret[0][2].items():       # both ranks hold the same gathered mirrors; the table's lo plane travelled too
        assert ret[1][2][name] == (fsum, losum), name
        ref = ps.full16[name].double().sum().item()
        assert abs(fsum - ref) <= 1e-3 * ps.full16[name].double().abs().sum().item(), name
    assert ret[0][2]["embed"][1] > 0


def _rccl_worker(port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from lap_amd.config import get_config
        from lap_amd.fsdp import FsdpComm
        from lap_amd.params import ParamStore
        from lap_amd.train import TrainingStepRunner, init_train_state
        from oracle import lap_oracle as O
        from tests.common import make_inputs, oracle_cfg, to_observation

        tc = get_config("debug")
        cfg = tc.model
        P = O.init_params(oracle_cfg(cfg), seed=9)
        ps = ParamStore(cfg, "cuda:0", world_size=1, rank=0)
        ps.load_reference_tree(P)
        comm = FsdpComm(ps)
        assert comm.fused and comm.world_size == 1            # backend "nccl" = RCCL: the fused tensor collectives
        u = ps.unit_by_name["llm0"]
        g = torch.randn(ps.padded(u), device="cuda:0")
        out = torch.empty_like(g)
        comm._reduce_scatter(out, g)                            # reduce_scatter_tensor on RCCL
        full = torch.randn(ps.padded(u), device="cuda:0").bfloat16()
        want = full.clone()
        comm._all_gather(full, full[0:full.numel()])            # in place (aliasing input / output), as the train step does
        comm.inplace_gather = False
        full2 = want.clone()
        comm._all_gather(full2, full2[0:full2.numel()])         # two-buffer fallback
        t = comm.all_reduce_sum(torch.tensor([3.0], device="cuda:0"))
        torch.cuda.synchronize()
        assert torch.equal(out, g) and torch.equal(full, want) and torch.equal(full2, want) and t.item() == 3.0
        # a full train step through FsdpComm on the RCCL process group equals the plain single-rank pipeline
        obs, actions, noise, time = make_inputs(cfg, B=2, ragged=True, seed=3)
        res = []
        for use_fsdp in (True, False):
            state = init_train_state(tc, params=P, device="cuda:0", world_size=1, rank=0, use_fsdp=use_fsdp)
            runner = TrainingStepRunner(tc)
            for step in range(2):
                state, info = runner(0, state, (to_observation(obs, "cuda:0"), actions.cuda()), step, noise=noise.cuda(), time=time.cuda())
            torch.cuda.synchronize()
            res.append((info["loss"].item(), info["grad_norm"].item(), runner.param_norm(state).item()))
        assert abs(res[0][0] - res[1][0]) < 1e-4 * abs(res[1][0]) and abs(res[0][1] - res[1][1]) < 1e-3 * res[1][1], res
        assert abs(res[0][2] - res[1][2]) < 1e-5 * res[1][2], res
        ret["r"] = "ok"
    except Exception:  # noqa: BLE001
        import traceback

        ret["r"] = traceback.format_exc()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_rccl_backend_executes_with_world_size_one(hip):
    """No multi-GPU box is available to the build, so the RCCL branch of FsdpComm (`fused=True`: reduce_scatter_tensor,
    all_gather_into_tensor in place and through the two-buffer fallback, all_reduce) runs here at least in its
    degenerate world_size-1 form on the real "nccl" backend, plus a train step over that process group."""
    port = 29950 + os.getpid() % 40
    mgr = mp.get_context("spawn").Manager()   # (a forked manager next to an initialised HIP runtime is not safe)
    ret = mgr.dict()
    p = mp.get_context("spawn").Process(target=_rccl_worker, args=(port, ret))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
    assert ret.get("r") == "ok", ret.get("r", "worker timed out")


def test_bench_self_launch_two_ranks_gloo_flow():
    """`bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (rendezvous on 127.0.0.1) exactly as the
    driver's multi-GPU bench does; with `--backend gloo` both ranks share this box's one GPU, so the whole N > 1 flow — rank
    environment, FSDP train state, barrier + max-over-ranks timing, rank 0's single JSON line — runs here (VERDICT r3 next #7a)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "debug", "--steps", "2",
                        "--warmup", "1", "--batch", "2", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parallelism"] == "fsdp2" and d["config"]["global_batch"] == 4
    assert d["scaling"] == "weak" and d["value"] > 0 and d["ms_per_step"] > 0 and "serve" not in d and "NON-HEADLINE" in d["config"]["workload"]
    assert d["final_loss"] == d["final_loss"]           # finite, not NaN
