"""Checkpoint save / resume for the MI355X engine.

Mirrors the surface of src/lap/training/checkpoints.py:
    mngr, resuming = initialize_checkpoint_dir(dir, keep_period=..., overwrite=..., resume=...)   (:67-127)
    save_state(mngr, state, data_loader, step, preserve_checkpoint=False, max_retries=...)         (:163-339)
    state = restore_state(mngr, state, data_loader, step=None)                                      (:342-437)
    params = restore_params(mngr_or_dir, step=None)                                                 (:440-474)
    load_norm_stats(assets_dir)                                                                     (:477-497)
and its three checkpoint items per step (:42-64, 529-547):
    params/       the parameters used for inference — the EMA parameters when EMA is on, else the live ones
                  (`_split_params`), as a tree in the REFERENCE's names and layouts (`{"params": {...}}` flattened with "/");
    train_state/  everything else: step, live f32 master parameters (when `params/` holds the EMA), Adam m / v;
    assets/       written by a callback: `<asset_id>/norm_stats.json`, `dataloader_process_<i>/dataloader_state.json`.
The reference stores these with Orbax (absent here, F5); this module writes safetensors + json.  `params/` is therefore
loadable by anything that knows the reference tree (`lap_amd.params.reference_to_engine` documents the key map); Orbax
directories must be converted offline to that file once.  `train_state/` is engine-layout and sharded: every rank writes
the slices it owns (`rank<r>_of<N>.safetensors`), so a resume needs the same world size.

Retention: max_to_keep = 1 plus every step that is a multiple of `keep_period` (:42-64); `preserve_checkpoint=True`
writes under `additional/` which is never pruned (:187-203).  A step directory becomes visible by an atomic rename after
all ranks have written, so a crash never leaves a half-written latest step.
"""
from __future__ import annotations

import dataclasses
import json
import logging
import pathlib
import shutil
import time

import torch

_COMMIT = "_COMMITTED"


@dataclasses.dataclass
class CheckpointManager:
    directory: pathlib.Path
    keep_period: int | None = None
    max_to_keep: int = 1

    def all_steps(self) -> tuple[int, ...]:
        if not self.directory.exists():
            return ()
        steps = [int(p.name) for p in self.directory.iterdir() if p.is_dir() and p.name.isdigit() and (p / _COMMIT).exists()]
        return tuple(sorted(steps))

    def latest_step(self) -> int | None:
        s = self.all_steps()
        return s[-1] if s else None

    def step_dir(self, step: int) -> pathlib.Path:
        return self.directory / str(int(step))

    def prune(self):
        steps = list(self.all_steps())
        keep = set(steps[-self.max_to_keep:])
        if self.keep_period:
            keep |= {s for s in steps if s % self.keep_period == 0}
        for s in steps:
            if s not in keep:
                shutil.rmtree(self.step_dir(s), ignore_errors=True)


def initialize_checkpoint_dir(checkpoint_dir, *, keep_period: int | None, overwrite: bool, resume: bool,
                              async_timeout_secs: int | None = 7200, async_enable: bool = True) -> tuple[CheckpointManager, bool]:
    """checkpoints.py:67-127 incl. the special case "directory exists but holds no checkpoint -> do not resume"."""
    checkpoint_dir = pathlib.Path(checkpoint_dir)
    resuming = False
    if checkpoint_dir.exists():
        if overwrite:
            shutil.rmtree(checkpoint_dir, ignore_errors=True)
            logging.info("Wiped checkpoint directory %s", checkpoint_dir)
        elif resume:
            resuming = True
        else:
            raise FileExistsError(f"Checkpoint directory {checkpoint_dir} already exists. Use --overwrite or --resume "
                                  "to indicate how to handle it.")
    checkpoint_dir.mkdir(parents=True, exist_ok=True)
    mngr = CheckpointManager(checkpoint_dir, keep_period=keep_period)
    if resuming and mngr.all_steps() in [(), (0,)]:
        logging.info("Checkpoint directory exists, but does not contain any checkpoints. Aborting resume.")
        resuming = False
    return mngr, resuming


# ----------------------------------------------------------------------------------------------------------- helpers
def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def _barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def _engine_layout(cfg) -> dict:
    """What shapes the engine-layout `train_state/` item besides the model config: the SigLIP MLP width after `siglip_mlp_pad`."""
    from lap_amd.config import get_siglip_config
    from lap_amd.params import siglip_mlp_pad

    md = get_siglip_config(cfg.siglip_variant).mlp_dim
    return {"siglip_mlp_dim": md, "siglip_mlp_width": siglip_mlp_pad(md)}


def _save_tensors(path: pathlib.Path, tensors: dict, metadata: dict | None = None):
    from safetensors.torch import save_file

    save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, str(path), metadata={k: str(v) for k, v in (metadata or {}).items()})


def _load_tensors(path: pathlib.Path) -> dict:
    from safetensors.torch import load_file

    return load_file(str(path), device="cpu")


def _strip_value(tree: dict) -> dict:
    """weight_loaders.py:184-189: leaves of released checkpoints may carry a trailing "value" key."""
    return {(k[:-len("/value")] if k.endswith("/value") else k): v for k, v in tree.items()}


# -------------------------------------------------------------------------------------------------------------- save
def save_state(checkpoint_manager: CheckpointManager, state, data_loader, step: int, *, max_retries: int = 0,
               retry_delay_secs: float = 0.0, retry_backoff: float = 1.0, fallback_to_sync: bool = False,
               async_timeout_secs: int | None = None, keep_period: int | None = None, preserve_checkpoint: bool = False,
               norm_stats: dict | None = None, asset_id: str = "combined") -> CheckpointManager:
    """checkpoints.py:163-339.  `state`: lap_amd.train.TrainState.  `data_loader`: anything with `get_state() -> dict`
    (or None).  `norm_stats`: {key: {mean,std,q01,q99,...}} written to assets/<asset_id>/norm_stats.json."""
    t0 = time.perf_counter()
    mngr = checkpoint_manager
    if preserve_checkpoint:   # never pruned (:187-203)
        mngr = CheckpointManager(checkpoint_manager.directory / "additional", keep_period=None, max_to_keep=10 ** 9)
        mngr.directory.mkdir(parents=True, exist_ok=True)
    elif keep_period is not None:
        mngr = dataclasses.replace(checkpoint_manager, keep_period=keep_period)
    attempt, delay = 0, retry_delay_secs
    while True:
        attempt += 1
        try:
            _save_once(mngr, state, data_loader, int(step), norm_stats, asset_id)
            break
        except Exception:   # noqa: BLE001 - the reference retries any failure with back-off (:206-339)
            if attempt > max_retries:
                raise
            logging.exception("Checkpoint save failed (attempt %d/%d); retrying in %.1fs", attempt, max_retries + 1, delay)
            time.sleep(delay)
            delay *= retry_backoff
    logging.info("Checkpoint save done | step=%d | dir=%s | %.2fs", step, mngr.directory, time.perf_counter() - t0)
    return checkpoint_manager


def _all_ok(ok: bool, device) -> bool:
    """True iff every rank reports success (MIN all-reduce): ranks agree on retrying or committing together."""
    d = _dist()
    if d is None:
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if d.get_backend() == "nccl" else "cpu")
    d.all_reduce(t, op=d.ReduceOp.MIN)
    return bool(t.item())


def _save_once(mngr: CheckpointManager, state, data_loader, step: int, norm_stats, asset_id):
    """One save attempt.  Every rank runs the same sequence of collectives whatever happens to its own FILE I/O: such local
    failures (directory setup, tensor / json writes, the commit rename) are caught, agreed on with `_all_ok`, and re-raised on
    ALL ranks, so the caller's retry loop iterates in lock-step (a rank retrying alone would leave barrier / all-gather counts
    mismatched and hang the job).  NOT recoverable, by construction: a failure INSIDE the collective phase in between (the
    stream synchronisation and the all-gathers of `to_reference_tree`: a HIP error surfacing at the sync, host memory exhausted
    while gathering the f32 tree) — the other ranks are already inside the collective, so that exception propagates and the job
    has to be restarted from the previous checkpoint."""
    ps = state.model.ps
    rank, world = ps.rank, ps.world_size
    final = mngr.step_dir(step)
    tmp = mngr.directory / f".tmp_{step}"
    err = None
    try:
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
            (tmp / "params").mkdir(parents=True)
            (tmp / "train_state").mkdir()
            (tmp / "assets").mkdir()
    except Exception as e:   # noqa: BLE001
        err = e
    if not _all_ok(err is None, ps.device):
        raise err if err is not None else RuntimeError("checkpoint save failed on another rank (directory setup)")
    if hasattr(state.model.comm, "synchronize"):
        state.model.comm.synchronize()   # the optimizer of the last step runs on the side stream
    if torch.cuda.is_available() and ps.device.type == "cuda":
        torch.cuda.synchronize(ps.device)
    has_ema = bool(ps.ema)
    # ---- params item: EMA-or-live in the reference's tree (collective when sharded: every rank takes part)
    tree = ps.to_reference_tree("ema" if has_ema else "master")
    try:
        if rank == 0:
            _save_tensors(tmp / "params" / "params.safetensors", {"params/" + k: v for k, v in tree.items()},
                          {"format": "lap reference tree, flattened with '/'", "ema": has_ema, "step": step})
        del tree
        # ---- train_state item: this rank's slices
        shard = {}
        for u in ps.units:
            if has_ema:
                shard[f"master/{u.name}"] = ps.master[u.name]
            if u.name in ps.m:
                shard[f"m/{u.name}"], shard[f"v/{u.name}"] = ps.m[u.name], ps.v[u.name]
        _save_tensors(tmp / "train_state" / f"rank{rank}_of{world}.safetensors", shard, {"step": step, "world_size": world})
        if rank == 0:
            (tmp / "train_state" / "meta.json").write_text(json.dumps(
                {"step": step, "world_size": world, "ema_decay": state.ema_decay, "has_ema": has_ema,
                 "units": {u.name: ps.padded(u) for u in ps.units}, "layout": _engine_layout(ps.cfg)}))
            # ---- assets (save_assets callback, :216-285)
            if norm_stats is not None:
                d = tmp / "assets" / asset_id
                d.mkdir(parents=True, exist_ok=True)
                (d / "norm_stats.json").write_text(json.dumps({"norm_stats": norm_stats}))
        if data_loader is not None and hasattr(data_loader, "get_state"):
            d = tmp / "assets" / f"dataloader_process_{rank}"
            d.mkdir(parents=True, exist_ok=True)
            (d / "dataloader_state.json").write_text(json.dumps(data_loader.get_state()))
    except Exception as e:   # noqa: BLE001
        err = e
    if not _all_ok(err is None, ps.device):
        raise err if err is not None else RuntimeError("checkpoint save failed on another rank (write phase)")
    try:
        if rank == 0:
            (tmp / _COMMIT).write_text(str(step))
            shutil.rmtree(final, ignore_errors=True)
            tmp.rename(final)
            mngr.prune()
    except Exception as e:   # noqa: BLE001
        err = e
    if not _all_ok(err is None, ps.device):
        raise err if err is not None else RuntimeError("checkpoint save failed on rank 0 (commit)")


# ----------------------------------------------------------------------------------------------------------- restore
def restore_params(checkpoint_manager, step: int | None = None) -> dict:
    """checkpoints.py:440-474: the `params` item (reference tree, f32) of `step` (default: latest).  Accepts a manager,
    a checkpoint directory, a step directory, a `params` item directory or the safetensors file itself (e.g. the output
    of tools/convert_orbax_checkpoint.py)."""
    p = checkpoint_manager.directory if isinstance(checkpoint_manager, CheckpointManager) else pathlib.Path(checkpoint_manager)
    if p.is_dir() and (p / "params.safetensors").exists():     # the `params` item itself (weight_loader.params_path, :440-474)
        p = p / "params.safetensors"
    if p.is_dir() and not (p / "params").exists():
        mngr = CheckpointManager(p)
        step = mngr.latest_step() if step is None else step
        if step is None:
            raise FileNotFoundError(f"no committed checkpoint under {p}")
        p = mngr.step_dir(step)
    if p.is_dir():
        p = p / "params" / "params.safetensors"
    flat = _strip_value(_load_tensors(p))
    return {(k[len("params/"):] if k.startswith("params/") else k): v for k, v in flat.items()}


def restore_state(checkpoint_manager: CheckpointManager, state, data_loader=None, step: int | None = None,
                  train_state_sharding=None):
    """checkpoints.py:342-437: parameters, optimizer moments, EMA, step counter and dataloader position."""
    mngr = checkpoint_manager
    step = mngr.latest_step() if step is None else int(step)
    if step is None:
        raise FileNotFoundError(f"no committed checkpoint under {mngr.directory}")
    state.model.ps.quiesce()     # nothing of an earlier optimizer pass may land on top of the restored values
    d = mngr.step_dir(step)
    meta = json.loads((d / "train_state" / "meta.json").read_text())
    ps = state.model.ps
    if meta["world_size"] != ps.world_size:
        raise ValueError(f"checkpoint was written by {meta['world_size']} ranks, this job has {ps.world_size} "
                         "(train_state is sharded; restore_params() gives the full parameter tree for any world size)")
    # `train_state/` is engine-layout: the switches that shape it are recorded in the meta (round 5) so that a mismatch
    # is reported by name.  Older checkpoints carry no "layout": the per-unit sizes below still catch a mismatch.
    lay, now = meta.get("layout"), _engine_layout(ps.cfg)
    if lay is not None and lay != now:
        raise ValueError(
            f"train_state/ of step {step} was written with engine layout {lay}, this process runs {now}.  The SigLIP MLP padding is chosen "
            f"by LAP_SIGLIP_PAD (lap_amd/params.py siglip_mlp_pad): resume with LAP_SIGLIP_PAD={'1' if lay['siglip_mlp_width'] != lay['siglip_mlp_dim'] else '0'}, "
            "or start a new optimizer state from the layout-independent `params/` item (restore_params / weight_loader.kind='checkpoint').")
    for u in ps.units:
        if meta["units"].get(u.name) != ps.padded(u):
            raise ValueError(f"unit {u.name}: checkpoint geometry {meta['units'].get(u.name)} != {ps.padded(u)}")
    shard = _load_tensors(d / "train_state" / f"rank{ps.rank}_of{ps.world_size}.safetensors")
    # inference params item -> EMA (or live when the run has no EMA)
    inf = restore_params(d)
    from lap_amd.params import reference_to_engine

    eng = reference_to_engine(ps.cfg, inf)
    target = ps.ema if meta["has_ema"] else ps.master
    if meta["has_ema"] and not ps.ema:
        raise ValueError("checkpoint carries EMA parameters but the train state was created without EMA")
    for u in ps.units:
        full = torch.zeros(ps.padded(u), dtype=torch.float32)
        for t in u.tensors:
            full[t.offset:t.offset + t.numel].copy_(eng[t.name].reshape(-1).to(torch.float32))
        a, b = ps.shard_range(u)
        target[u.name].copy_(full[a:b])
        if meta["has_ema"]:
            ps.master[u.name].copy_(shard[f"master/{u.name}"])
        if u.name in ps.m:
            ps.m[u.name].copy_(shard[f"m/{u.name}"])
            ps.v[u.name].copy_(shard[f"v/{u.name}"])
    if not meta["has_ema"] and ps.ema:
        # the run tracks an EMA the checkpoint never had: start it from the restored parameters (scripts/train.py:376-396
        # re-initialises a structurally missing EMA from the live parameters the same way)
        logging.warning("checkpoint %s has no EMA parameters; initialising the EMA from the restored parameters", d)
        ps.sync_ema_from_master()
    _refresh_mirrors(state)
    if data_loader is not None and hasattr(data_loader, "set_state"):
        f = d / "assets" / f"dataloader_process_{ps.rank}" / "dataloader_state.json"
        if f.exists():
            data_loader.set_state(json.loads(f.read_text()))
    return dataclasses.replace(state, step=int(meta["step"]), ema_decay=meta.get("ema_decay", state.ema_decay))


def _refresh_mirrors(state):
    """bf16 compute mirrors <- restored f32 masters (all-gathered under FSDP)."""
    ps, comm = state.model.ps, state.model.comm
    ps.version += 1      # masters / mirrors were written directly: derived copies (the fp8 weight mirrors of LAP._w8_of) are stale
    if ps.world_size == 1:
        ps.refresh_mirror_local()
        return
    for u in ps.units:
        if ps.sharded(u):
            a, b = ps.shard_range(u)
            ps.full16[u.name][a:b].copy_(ps.master[u.name])
    ps.refresh_lo_shard()
    comm.start_param_gather()
    comm.synchronize()


def load_norm_stats(assets_dir) -> dict | None:
    """checkpoints.py:477-497: exactly one sub-directory with a norm_stats.json is expected."""
    assets_dir = pathlib.Path(assets_dir)
    norm_dirs = [p for p in assets_dir.iterdir() if p.is_dir() and (p / "norm_stats.json").exists()]
    assert len(norm_dirs) == 1, (f"Expected exactly one norm stats directory in {assets_dir}, but found {len(norm_dirs)}: "
                                 f"{[p.name for p in norm_dirs]}")
    data = json.loads((norm_dirs[0] / "norm_stats.json").read_text())
    return data.get("norm_stats", data)
