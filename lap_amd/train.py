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


_KERNEL_PARAM_EXCLUDE = ("_b", "ln1_g", "ln2_g", "norm_g", "n_attn", "n_ffw", "final_norm", "img/pos", "llm/embed", "/b1", "/b2", "/bo",
                         "bqkv", "ada/b")


def _is_kernel_param(name: str, shape) -> bool:
    """train.py:401-408: ndim > 1 and not bias / scale / pos_embedding / input_embedding."""
    if len(shape) < 2:
        return False
    return not any(name.endswith(s) or s in name for s in ("img/pos", "llm/embed"))


def init_train_state(config: TrainConfig, seed: int | None = None, *, device="cuda", params: dict | None = None, comm=None,
                     world_size: int = 1, rank: int = 0, use_fsdp: bool = False) -> TrainState:
    """scripts/train.py:202-326 (init_train_state): model init (+ optional weight merge), optimizer and EMA state.
    With use_fsdp the store is sharded over the default process group (every rank draws the same full random
    init from the same seed and keeps its slice)."""
    ema_decay, ema_enabled = config.get_ema_init()
    store = ParamStore(config.model, device, world_size=world_size, rank=rank, with_optimizer=True, with_ema=ema_enabled)
    if params is not None:
        store.load_reference_tree(params)
    else:
        store.init_random(config.seed if seed is None else seed)
    if use_fsdp and comm is None:
        from lap_amd.fsdp import FsdpComm

        comm = FsdpComm(store)
    model = LAP(config.model, device=device, store=store, comm=comm)
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
        dev = model.device
        # weight gradients are written (beta = 0) by the wgrad GEMMs; only the replicated f32 unit is accumulated
        # into with atomics (norm scales, biases, f32 stem / action head) and must start from zero
        for u in ps.units:
            if not u.big:
                ps.grad[u.name].zero_()
        seed = (int(rng) * 1_000_003 + state.step) if not isinstance(rng, torch.Generator) else rng  # fold_in(rng, step)
        loss, metrics = model.loss_and_grad(seed, observation, actions, train=True, noise=noise, time=time)
        comm = model.comm
        comm.finish_grads() if hasattr(comm, "finish_grads") else None
        # global gradient norm over this rank's shards (+ replicated unit once), then across ranks
        sumsq = torch.zeros(2, dtype=torch.float32, device=dev)
        for u in ps.units:
            hip.sumsq_f32(ps.gshard[u.name], sumsq[0:1] if ps.sharded(u) or comm.world_size == 1 else sumsq[1:2])
        if comm.world_size > 1:
            comm.all_reduce_sum(sumsq[0:1])
        total = (sumsq[0] + sumsq[1]).view(1)
        lr = cfg.lr_schedule(step)
        opt = cfg.optimizer
        t = state.step + 1
        ema_decay, ema_on = cfg.get_ema_decay_for_step(step)
        scal = torch.tensor([0.0, lr, 1.0 - opt.b1 ** t, 1.0 - opt.b2 ** t, ema_decay, 1.0 if ema_on else 0.0, 0.0, 0.0],
                            dtype=torch.float32, device=dev)
        scal[0:1] = total
        for u in ps.units:
            a, b = ps.shard_range(u)
            p16 = ps.full16[u.name][a:b] if u.big else None
            hip.adamw_ema(ps.master[u.name], ps.m[u.name], ps.v[u.name], ps.ema.get(u.name), ps.gshard[u.name], p16, scal,
                          opt.b1, opt.b2, opt.eps, opt.weight_decay, opt.clip_gradient_norm)
        if hasattr(comm, "start_param_gather"):
            comm.start_param_gather()
        grad_norm = total.sqrt().view(())
        info = {"loss": loss, "grad_norm": grad_norm, "grad_norm_f32": grad_norm, **metrics}
        return dataclasses.replace(state, step=state.step + 1), info

    def param_norm(self, state: TrainState) -> torch.Tensor:
        """optax.global_norm over kernel parameters (train.py:401-415); computed on demand (logging interval)."""
        ps = state.model.ps
        acc = torch.zeros(1, dtype=torch.float32, device=state.model.device)
        for name, spec in ps.tensor_spec.items():
            if _is_kernel_param(name, spec.shape) and not ps.sharded(ps.tensor_unit[name]):
                hip.sumsq_f32(ps.f32(name).reshape(-1), acc)
        return acc.sqrt().view(())
