"""Train state and train step of the MI355X engine.

Mirrors scripts/train.py:329-419 (TrainingStepRunner.__call__) and src/lap/training/state.py:
    new_state, info = train_step(rng, state, (observation, actions), step)
with info keys loss / grad_norm / grad_norm_f32 / param_norm + the loss metrics.  The optimizer is
optax.chain(clip_by_global_norm(1.0), adamw(lr_t, b1, b2, eps, weight_decay)) [UPSTREAM-RECALL openpi
optimizer.create_optimizer] followed by the EMA update `where(enabled, d*ema + (1-d)*p, ema)`
(train.py:376-396) — fused into one HBM pass per parameter unit (csrc/loss_optim.hip).
"""
from __future__ import annotations

import dataclasses
import os

import torch

from lap_amd import hip
from lap_amd.config import TrainConfig
from lap_amd.model import LAP
from lap_amd.params import ParamStore


@dataclasses.dataclass
class TrainState:
    """training/state.py:10-18.  `params` / `opt_state` / `ema_params` live in `model.ps` (flat HBM buffers)."""
    step: int
    model: LAP
    ema_decay: float | None

    @property
    def params(self) -> ParamStore:
        return self.model.ps


def _is_kernel_param(name: str, shape) -> bool:
    """train.py:401-408: ndim > 1 and not bias / scale / pos_embedding / input_embedding."""
    if len(shape) < 2:
        return False
    return not any(name.endswith(s) or s in name for s in ("img/pos", "llm/embed"))


def validate_loaded_params(expected: dict, got: dict, *, allow_partial: bool) -> dict:
    """scripts/train.py:157-187 (_validate_loaded_params) on '/'-flattened trees.  `expected`: path -> shape of the
    model's parameter tree; `got`: path -> array.  A key the model does not have, or a shape / dtype that differs, is an
    error; missing keys are an error unless `allow_partial` (they keep their initial values).  Returns `got` (f32)."""
    import logging

    unexpected = [k for k in got if k not in expected]
    if unexpected:
        raise ValueError(f"Loaded params contain unexpected keys (sample): {', '.join(unexpected[:8])}")
    bad = []
    for k, v in got.items():
        if tuple(v.shape) != tuple(expected[k]):
            bad.append(f"{k} (shape {tuple(v.shape)} != {tuple(expected[k])})")
        elif not torch.as_tensor(v).dtype.is_floating_point:
            bad.append(f"{k} (dtype {torch.as_tensor(v).dtype} is not a float type)")
    if bad:
        raise ValueError(f"Loaded params do not match expected shapes/dtypes (sample): {', '.join(bad[:8])}")
    missing = sorted(set(expected) - set(got))
    if missing:
        if not allow_partial:
            raise ValueError(f"Loaded params missing required keys: {', '.join(missing)}")
        logging.info("Weight loader missing %d params; using random init for them: %s", len(missing), ", ".join(missing))
    return got


def load_paligemma_npz(path, expected: dict) -> dict:
    """PaliGemmaWeightLoader.load (weight_loaders.py:109-124): the official PaliGemma `.npz` is a flat dict of '/'-joined
    keys; the subtree under `params` becomes the model's `PaliGemma` subtree, keys the model does not have are dropped and
    dtypes follow the model (`_merge_params`, :691-719).  Returns {reference path: f32 tensor} for the keys it could take."""
    import numpy as np

    path = __import__("pathlib").Path(path)
    if not path.is_file():
        raise FileNotFoundError(
            f"PaliGemma checkpoint {path} not found.  The reference downloads gs://vertex-model-garden-paligemma-us/paligemma/"
            "pt_224.npz (weight_loaders.py:117-119); this engine has no network access: place the file there, or point "
            "weight_loader.params_path / $LAP_PALIGEMMA_NPZ at it, or choose weight_loader.kind='none' for a random init.")
    out = {}
    with np.load(path, allow_pickle=False) as z:
        for key in z.files:
            if not key.startswith("params/"):
                continue
            ref = "PaliGemma/" + key[len("params/"):]
            if ref in expected:
                a = z[key]
                out[ref] = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32) if a.dtype.kind == "f" or a.dtype.kind == "V" else a))
    if not out:
        raise ValueError(f"{path}: no key under 'params/' matches the model's PaliGemma subtree")
    return out


def load_weights(config: TrainConfig, store: ParamStore):
    """The `weight_loader` of the config executed against a freshly initialised store (scripts/train.py:191-199,248-310;
    weight_loaders.py:55-124,691-719): load a parameter tree in the reference's layout, keep the keys the model knows
    (`_merge_params` drops the rest and casts dtypes), validate, and merge over the init.
      checkpoint  a checkpoint's `params` item; missing keys allowed iff `allow_partial_weights` (CheckpointWeightLoader merges
                  only `.*lora.*` back, so anything else missing reaches `_validate_loaded_params`);
      paligemma   the PaliGemma `.npz` (the reference's default kind); `missing_regex=".*"` — every key it lacks (the action
                  expert, the action / time heads) keeps its init regardless of `allow_partial_weights`."""
    from lap_amd import checkpoints as ck
    from lap_amd.params import reference_shapes

    wl = config.weight_loader
    if wl.kind == "none":
        return False
    expected = reference_shapes(config.model)
    if wl.kind == "checkpoint":
        if not wl.params_path:
            raise ValueError("--weight-loader.params-path must be set when kind=checkpoint")     # weight_loaders.py:659-661
        loaded = ck.restore_params(wl.params_path)           # '/value' suffixes and the 'params/' prefix are stripped there
        subset = {k: torch.as_tensor(v) for k, v in loaded.items() if k in expected}     # _merge_params: subset of the model's keys
        allow_partial = config.allow_partial_weights
    elif wl.kind == "paligemma":
        subset = load_paligemma_npz(wl.resolve_paligemma_path(), expected)
        allow_partial = True
    else:
        raise NotImplementedError(f"weight loader kind {wl.kind!r} (paligemma2 / gemma3 loaders belong to model variants that are "
                                  "out of scope: SURVEY.md §2)")
    subset = validate_loaded_params(expected, subset, allow_partial=allow_partial)
    if len(subset) < len(expected):                      # partial: the model's own (random-init) arrays fill the gaps
        base = store.to_reference_tree("master")
        base.update(subset)
        subset = base
    store.load_reference_tree(subset)
    return True


def init_train_state(config: TrainConfig, seed: int | None = None, *, device="cuda", params: dict | None = None, comm=None,
                     world_size: int = 1, rank: int = 0, use_fsdp: bool = False, resume: bool = False) -> TrainState:
    """scripts/train.py:202-326 (init_train_state): model init, the config's weight loader merged over it (skipped when
    resuming: the checkpoint overwrites everything), frozen parameters per `freeze_filter`, optimizer and EMA state.
    With use_fsdp the store is sharded over the default process group (every rank draws the same full random
    init from the same seed and keeps its slice)."""
    ema_decay, ema_enabled = config.get_ema_init()
    store = ParamStore(config.model, device, world_size=world_size, rank=rank, with_optimizer=True, with_ema=ema_enabled)
    if params is not None:
        store.load_reference_tree(params)
    else:
        store.init_random(config.seed if seed is None else seed)
        if not resume:
            load_weights(config, store)
    store.set_frozen(config.is_frozen if config.freeze_filter is not None else None)
    if comm is None:
        from lap_amd.fsdp import FsdpComm, UnitPipeline

        comm = FsdpComm(store) if use_fsdp else UnitPipeline(store)
    model = LAP(config.model, device=device, store=store, comm=comm, gemm_dtype=config.gemm_dtype)
    return TrainState(step=0, model=model, ema_decay=ema_decay)


class TrainingStepRunner:
    def __init__(self, config: TrainConfig):
        self.config = config
        self._scalars = None

    def __call__(self, rng, state: TrainState, batch, step: int | None = None, *, noise=None, time=None):
        cfg = self.config
        model, ps = state.model, state.model.ps
        observation, actions = batch
        step = state.step if step is None else int(step)
        pipe = model.comm
        if not hasattr(pipe, "run_optimizer"):
            raise RuntimeError("train step needs a UnitPipeline / FsdpComm (use init_train_state)")
        pipe.begin_step()
        # weight gradients are written (beta = 0) by the wgrad GEMMs; only the replicated f32 unit is accumulated
        # into with atomics (norm scales, biases, f32 stem / action head) and must start from zero
        for u in ps.units:
            if not u.big:
                ps.grad[u.name].zero_()
        seed = (int(rng) * 1_000_003 + state.step) if not isinstance(rng, torch.Generator) else rng  # fold_in(rng, step)
        loss, metrics = model.loss_and_grad(seed, observation, actions, train=True, noise=noise, time=time)
        # gradient reductions / norm partials were enqueued per unit during the backward; the fused clip+AdamW+EMA
        # (+ parameter all-gather) now runs unit by unit on the side stream and overlaps with the next forward
        opt = cfg.optimizer
        t = state.step + 1
        ema_decay, ema_on = cfg.get_ema_decay_for_step(step)
        grad_norm = pipe.run_optimizer(cfg.lr_schedule(step), 1.0 - opt.b1 ** t, 1.0 - opt.b2 ** t, ema_decay, ema_on, opt)
        info = {"loss": loss, "grad_norm": grad_norm, "grad_norm_f32": grad_norm, **metrics}
        return dataclasses.replace(state, step=state.step + 1), info

    def param_norm(self, state: TrainState) -> torch.Tensor:
        """optax.global_norm over kernel parameters (train.py:401-415); computed on demand (logging interval).
        Collective under FSDP: per-shard partial sums, all-reduced."""
        return state.model.comm.param_sumsq(_is_kernel_param).sqrt().view(())


class ValidationStepRunner:
    """scripts/train.py:422-450: `metrics = runner(rng, state, (observation, actions))` — `compute_loss(..., train=False)` on the
    live parameters with `fold_in(rng, state.step)` as the key, the loss reported as `val_loss` next to the loss metrics.  No
    gradients, no optimizer; under FSDP the loss is the all-reduced global value like the train loss."""

    def __init__(self, config: TrainConfig):
        self.config = config

    @torch.no_grad()
    def __call__(self, rng, state: TrainState, batch, *, noise=None, time=None) -> dict:
        observation, actions = batch
        seed = (int(rng) * 1_000_003 + state.step) if not isinstance(rng, torch.Generator) else rng
        val_loss, val_metrics = state.model.compute_loss(seed, observation, actions, train=False, noise=noise, time=time)
        val_metrics = dict(val_metrics)
        val_metrics["val_loss"] = val_loss
        return val_metrics


def run_validation(runner: ValidationStepRunner, rng, state: TrainState, val_loader, num_batches: int | None = None) -> dict:
    """The periodic validation pass of the train loop (scripts/train.py:619-660): a FRESH iterator every time, so the same
    fixed validation subset is scored at every interval; up to `num_batches` batches (default: the loader's
    `num_val_batches()`, else until it is exhausted); returns the mean of each scalar metric with a `val_` prefix
    (`val_loss` keeps its name)."""
    if num_batches is None and hasattr(val_loader, "num_val_batches"):
        num_batches = int(val_loader.num_val_batches())
    infos = []
    for i, batch in enumerate(iter(val_loader)):
        if num_batches is not None and i >= num_batches:
            break
        infos.append(runner(rng, state, batch))
    if not infos:
        return {}
    keys = [k for k, v in infos[0].items() if torch.is_tensor(v) and v.numel() == 1]
    return {(k if k.startswith("val_") else "val_" + k): float(torch.stack([i[k].float().reshape(()) for i in infos]).mean()) for k in keys}


# ====================================================================================== training entry point
class SyntheticDataLoader:
    """Stand-in for datasets/data_loader.py: seeded batches of the benchmark shape (§8d) with
    the protocol the train loop and the checkpoint code use (`__iter__`, `get_state`, `set_state`).  Rank r of N draws
    batch `N * i + r`, so a resumed run continues with the batches the interrupted run would have seen."""

    def __init__(self, cfg, per_rank_batch: int, device, *, seed: int = 0, rank: int = 0, world_size: int = 1,
                 num_batches: int | None = None):
        self.cfg, self.B, self.device = cfg, per_rank_batch, device
        self.seed, self.rank, self.world = seed, rank, world_size
        self.index = 0
        self.num_batches = num_batches          # None: endless (train split); n: one pass of n batches per iterator (val split)

    def num_val_batches(self) -> int:
        if self.num_batches is None:
            raise ValueError("an endless loader has no validation batch count")
        return self.num_batches

    def get_state(self) -> dict:
        return {"index": self.index, "seed": self.seed}

    def set_state(self, s: dict):
        self.index, self.seed = int(s["index"]), int(s["seed"])

    def __iter__(self):
        if self.num_batches is not None:
            self.index = 0                        # a fresh iterator scores the same fixed subset again
        return self

    def __next__(self):
        from lap_amd.observation import CoTObservation

        if self.num_batches is not None and self.index >= self.num_batches:
            raise StopIteration

        cfg, B, dev = self.cfg, self.B, self.device
        g = torch.Generator(device="cpu").manual_seed(self.seed * 1_000_003 + self.index * self.world + self.rank)
        self.index += 1
        L, H = cfg.max_token_len, cfg.image_size
        la = torch.zeros(B, L, dtype=torch.bool)
        la[:, L - min(16, L // 2):] = True
        obs = CoTObservation(
            images={k: (torch.rand(B, H, H, 3, generator=g) * 2 - 1).to(dev) for k in cfg.image_keys},
            image_masks={k: torch.ones(B, dtype=torch.bool, device=dev) for k in cfg.image_keys},
            state=(torch.rand(B, cfg.action_dim, generator=g) * 2 - 1).to(dev),
            tokenized_prompt=torch.randint(0, cfg.vocab_size, (B, L), generator=g, dtype=torch.int32).to(dev),
            tokenized_prompt_mask=torch.ones(B, L, dtype=torch.bool, device=dev),
            tokenized_langact_mask=la.to(dev), token_loss_mask=torch.ones(B, L, dtype=torch.bool, device=dev),
            sample_mask=torch.ones(B, dtype=torch.bool, device=dev), loss_rows_max=int(la[:, 1:].sum(-1).max()))
        return obs, torch.randn(B, cfg.action_horizon, cfg.action_dim, generator=g).to(dev)


def main(config: TrainConfig, *, data_loader=None, val_data_loader=None, device: str | None = None, log=print) -> TrainState:
    """scripts/train.py:422-640 (main): distributed init, train state (+ resume), the step loop with interval logging
    and checkpointing.  One process per GPU; under `torch.distributed.run` the parameters / optimizer / EMA are ZeRO-3
    sharded over all ranks and `batch_size` is the GLOBAL batch (config.py:783)."""
    import pathlib
    import time as _time

    import torch.distributed as dist

    from lap_amd import checkpoints as ck

    # kernel arguments in device memory: ~1 ms per step less launch latency (see bench.py); only effective if the HIP runtime
    # has not started yet in this process, harmless otherwise; an explicit value in the environment wins
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        device = f"cuda:{local % max(torch.cuda.device_count(), 1)}"
    if torch.device(device).type == "cuda" and torch.device(device).index is not None:
        torch.cuda.set_device(torch.device(device))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.device(device).type == "cuda" else "gloo", rank=rank, world_size=world)
    if config.batch_size % world:
        raise ValueError(f"batch_size {config.batch_size} must be divisible by the number of ranks {world}")
    # rank 0 alone looks at / wipes / creates the directory and decides whether this is a resume; everyone else waits
    # and takes its verdict (each rank deciding for itself races: a late rank would see the directory rank 0 just made)
    if rank == 0:
        try:
            mngr, resuming = ck.initialize_checkpoint_dir(config.checkpoint_dir, keep_period=config.keep_period,
                                                          overwrite=config.overwrite, resume=config.resume)
            verdict = [resuming, None, None]
        except Exception as e:   # noqa: BLE001 - re-raised on every rank below
            verdict = [False, type(e).__name__, str(e)]
    else:
        verdict = [False, None, None]
    if world > 1:
        dist.broadcast_object_list(verdict, src=0)
    if verdict[1] is not None:   # the same exception on every rank (FileExistsError keeps its type: callers catch it)
        raise (FileExistsError if verdict[1] == "FileExistsError" else RuntimeError)(verdict[2])
    resuming = bool(verdict[0])
    if rank != 0:
        mngr = ck.CheckpointManager(pathlib.Path(config.checkpoint_dir), keep_period=config.keep_period)
    state = init_train_state(config, device=device, world_size=world, rank=rank, use_fsdp=world > 1, resume=resuming)
    if data_loader is None:
        data_loader = SyntheticDataLoader(config.model, config.batch_size // world, device, seed=config.seed, rank=rank, world_size=world)
    if resuming:
        state = ck.restore_state(mngr, state, data_loader)
        log(f"resumed from step {state.step} ({mngr.directory})")
    runner = TrainingStepRunner(config)
    val_runner = None
    if config.use_validation:                            # scripts/train.py:539-571
        if val_data_loader is None:
            if not isinstance(data_loader, SyntheticDataLoader):
                raise ValueError("use_validation=True needs a `val_data_loader` (split='val' of the training data)")
            val_data_loader = SyntheticDataLoader(config.model, config.batch_size // world, device, seed=config.seed + 7919, rank=rank,
                                                  world_size=world, num_batches=2)
        val_runner = ValidationStepRunner(config)
    it = iter(data_loader)
    infos, t_last, start = [], _time.perf_counter(), state.step
    for step in range(start, config.num_train_steps):
        state, info = runner(config.seed, state, next(it), step)
        infos.append(info)
        last = step == config.num_train_steps - 1
        if (step + 1) % config.log_interval == 0 or last:   # mean of the interval's step infos (metrics_logging.py:181-237)
            keys = [k for k, v in infos[0].items() if torch.is_tensor(v) and v.numel() == 1]
            mean = {k: float(torch.stack([i[k].float().reshape(()) for i in infos]).mean()) for k in keys}
            if world > 1:
                t = torch.tensor([mean[k] for k in keys], device=device)
                dist.all_reduce(t)
                mean = {k: float(v) / world for k, v in zip(keys, t)}
            mean["param_norm"] = float(runner.param_norm(state))
            dt = _time.perf_counter() - t_last
            if rank == 0:
                log(f"step {step + 1}: " + ", ".join(f"{k}={v:.4f}" for k, v in mean.items()) +
                    f" | {len(infos) * config.batch_size / dt:.1f} samples/s")
            infos, t_last = [], _time.perf_counter()
        if val_runner is not None and (step + 1) % config.val_interval == 0:          # scripts/train.py:619-660
            vm = run_validation(val_runner, config.seed, state, val_data_loader)
            if world > 1 and vm:
                t = torch.tensor(list(vm.values()), device=device)
                dist.all_reduce(t)
                vm = {k: float(v) / world for k, v in zip(vm, t)}
            if rank == 0 and vm:
                log(f"step {step + 1} validation: " + ", ".join(f"{k}={v:.4f}" for k, v in vm.items()))
        if ((step + 1) % config.save_interval == 0 and step + 1 > start) or last:
            # save_assets callback (training/checkpoints.py:216-285): the loader's normalisation statistics travel with the
            # checkpoint, so the policy built from it un-normalises with the training-time numbers
            stats = data_loader.get_norm_stats_for_checkpoint()[0] if hasattr(data_loader, "get_norm_stats_for_checkpoint") else None
            ck.save_state(mngr, state, data_loader, step + 1, norm_stats=stats, asset_id=getattr(config.data, "asset_id", None) or "combined")
            if rank == 0:
                log(f"saved checkpoint {step + 1}")
    return state


if __name__ == "__main__":
    from lap_amd.config import cli

    main(cli())
