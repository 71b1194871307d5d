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
    if comm is None:
        from lap_amd.fsdp import FsdpComm, UnitPipeline

        comm = FsdpComm(store) if use_fsdp else UnitPipeline(store)
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
        """optax.global_norm over kernel parameters (train.py:401-415); computed on demand (logging interval)."""
        ps = state.model.ps
        state.model.comm.synchronize() if hasattr(state.model.comm, "synchronize") else None
        acc = torch.zeros(1, dtype=torch.float32, device=state.model.device)
        for name, spec in ps.tensor_spec.items():
            if _is_kernel_param(name, spec.shape) and not ps.sharded(ps.tensor_unit[name]):
                hip.sumsq_f32(ps.f32(name).reshape(-1), acc)
        return acc.sqrt().view(())
