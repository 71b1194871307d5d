"""FSDP over RCCL/xGMI for the MI355X engine: one process per GPU, torch.distributed (backend "nccl" = RCCL).

Reference: parameters / optimizer state / EMA sharded by `fsdp_sharding` over the `fsdp` mesh axis and XLA GSPMD
inserting a per-layer parameter all-gather in the forward, a second one in the rematerialised backward, and a
gradient reduce-scatter (src/lap/training/mh_sharding.py:14-100, scripts/train.py:532-537).

MI355X-first partitioning (ZeRO-3 state, resident bf16 replicas):
  * each rank owns a contiguous 1/N slice of every big unit's f32 master / Adam m,v / EMA (lap_amd/params.py);
  * the fused optimizer kernel writes the updated bf16 values of the owned slice straight into the unit's bf16
    mirror; `start_param_gather` then all-gathers the mirrors IN PLACE, unit by unit in forward order, on a side
    HIP stream, overlapping with the next step's forward (a unit's first GEMM waits on that unit's event only);
  * 288 GB of HBM per GPU keeps all gathered bf16 weights (6.7 GB) resident, so the backward needs NO second
    all-gather: 2 x 5.9 GB of xGMI traffic per step instead of the reference's 3 x 5.9 GB;
  * gradients are produced in full f32 unit buffers; `grads_ready(unit)` enqueues a reduce-scatter (sum) of that
    unit on the side stream as soon as its backward is done, overlapping with the remaining backward;
  * the small replicated unit (norm scales, biases, f32 stem / action head) is all-reduced;
  * the embedding gather needs f32 rows (gemma.py:148-151): every rank looks up the rows it owns for ALL ranks'
    tokens, and one bf16 reduce-scatter (sum of one non-zero and N-1 zero rows: exact) hands each rank its rows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from lap_amd import hip
from lap_amd.params import ParamStore


class FsdpComm:
    def __init__(self, store: ParamStore, group=None):
        self.ps = store
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if store.world_size != self.world_size or store.rank != self.rank:
            raise ValueError("ParamStore world_size/rank do not match the process group")
        self.is_cuda = store.device.type == "cuda"
        # RCCL ("nccl") has the fused tensor collectives; gloo (CPU tests, single-GPU multi-process tests) does not
        self.fused = dist.get_backend(group) == "nccl"
        self.side = torch.cuda.Stream(device=store.device) if self.is_cuda else None
        self.param_events: dict[str, object] = {}
        self._pending_grads = False

    # ---- streams (CPU/gloo tests run everything inline)
    def _on_side(self):
        import contextlib

        if not self.is_cuda:
            return contextlib.nullcontext()
        self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    # ---- parameters
    def start_param_gather(self):
        """All-gather every big unit's bf16 mirror (forward order) after the optimizer updated the owned slices."""
        with self._on_side():
            for u in self.ps.units:
                if not self.ps.sharded(u):
                    continue
                full = self.ps.full16[u.name]
                a, b = self.ps.shard_range(u)
                self._all_gather(full, full[a:b])
                if self.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.param_events[u.name] = ev

    def wait_unit(self, name: str):
        ev = self.param_events.pop(name, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # ---- gradients
    def grads_ready(self, name: str):
        u = self.ps.unit_by_name[name]
        with self._on_side():
            if self.ps.sharded(u):
                self._reduce_scatter(self.ps.gshard[name], self.ps.grad[name])
            else:
                dist.all_reduce(self.ps.grad[name], op=dist.ReduceOp.SUM, group=self.group)
        self._pending_grads = True

    def finish_grads(self):
        if self.is_cuda and self._pending_grads:
            torch.cuda.current_stream().wait_stream(self.side)
        self._pending_grads = False

    # ---- backend shims
    def _all_gather(self, full: torch.Tensor, mine: torch.Tensor):
        if self.fused:
            dist.all_gather_into_tensor(full, mine, group=self.group)  # in place: `mine` is full[rank*n:(rank+1)*n]
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            dist.all_gather(parts, mine.clone(), group=self.group)
            full.copy_(torch.cat(parts))

    def _reduce_scatter(self, out: torch.Tensor, full: torch.Tensor):
        if self.fused:
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)
        else:
            tmp = full.clone()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            n = out.numel()
            out.copy_(tmp[self.rank * n:(self.rank + 1) * n])

    # ---- small collectives on the compute stream
    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def sharded_embed_gather(self, rows, lo, hi, tokens, x0, Lt, Dv, Pn, dst_off, scale):
        B = tokens.shape[0]
        N = self.world_size
        all_tok = torch.empty((N * B, Lt), dtype=torch.int32, device=tokens.device)
        self._all_gather(all_tok.view(-1), tokens.reshape(-1))
        part = torch.empty((N * B * Lt, Dv), dtype=torch.bfloat16, device=tokens.device)
        hip.embed_gather(rows, all_tok, part, N * B * Lt, Lt, Dv, Lt, 0, scale, lo, hi)
        mine = torch.empty((B * Lt, Dv), dtype=torch.bfloat16, device=tokens.device)
        self._reduce_scatter(mine.view(-1), part.view(-1))
        hip.copy_rows_bf16(mine, x0, B * Lt, Lt, Dv, Lt, 0, Pn, dst_off)
