"""Per-unit step pipeline on a side HIP stream: gradient reduction, gradient-norm partial sums, the fused optimizer
and (with FSDP) the parameter all-gather — overlapped with the compute stream.  One process per GPU;
torch.distributed backend "nccl" (= RCCL over xGMI) when world_size > 1.

Reference: parameters / optimizer state / EMA sharded by `fsdp_sharding` over the `fsdp` mesh axis and XLA GSPMD
inserting a per-layer parameter all-gather in the forward, a second one in the rematerialised backward, and a
gradient reduce-scatter (src/lap/training/mh_sharding.py:14-100, scripts/train.py:532-537); the
optax update runs after the backward inside the same jit (train.py:363-396).

MI355X-first schedule (ZeRO-3 state, resident bf16 replicas, optimizer off the critical path):
  * each rank owns a contiguous 1/N slice of every big unit's f32 master / Adam m,v / EMA (lap_amd/params.py);
  * `grads_ready(unit)` — called by the backward the moment a unit's gradients are final — enqueues on the side
    stream: reduce-scatter (sum) of that unit's f32 gradient buffer (all-reduce for the small replicated unit) and
    the unit's contribution to the global gradient norm.  Both overlap with the rest of the backward;
  * `run_optimizer(...)` schedules, unit by unit IN FORWARD ORDER on the side stream: fused clip+AdamW+EMA on the
    owned slice (which also writes the bf16 values into the unit's bf16 mirror), then the in-place all-gather of
    that mirror, then an event.  The next step's forward waits per unit (`wait_unit`), so the HBM-bound optimizer
    and the xGMI all-gather of later units hide under the forward of earlier ones.  The pass is PACED: only the first
    `LAP_OPT_LOOKAHEAD` units are enqueued at once; unit k + lookahead is released (a side-stream wait on a compute-stream
    event) when the forward reaches unit k.  Enqueued all at once, the 120 GB of optimizer traffic land on the first
    ~25 ms of the next forward — the SigLIP tower, whose short-K GEMMs and head-size-72 attention are the most
    HBM-sensitive kernels of the step (measured 1.5-2x slower in situ than isolated); paced, each Gemma layer's update
    runs beside the compute-bound projections of the layer before it;
  * 288 GB of HBM per GPU keeps all gathered bf16 weights (6.7 GB) resident, so the backward needs NO second
    all-gather: 2 x 5.9 GB of xGMI traffic per step instead of the reference's 3 x 5.9 GB;
  * the embedding gather needs f32 rows (gemma.py:148-151): every rank looks up the rows it owns for ALL ranks'
    tokens, and one bf16 reduce-scatter (sum of one non-zero and N-1 zero rows: exact) hands each rank its rows.
"""
from __future__ import annotations

import contextlib

import torch

from lap_amd import hip
from lap_amd.params import ParamStore


class UnitPipeline:
    """world_size == 1 pipeline; FsdpComm adds the collectives."""

    world_size = 1
    rank = 0

    def __init__(self, store: ParamStore):
        self.ps = store
        self.is_cuda = store.device.type == "cuda"
        import os
        # LAP_OPT_PRIORITY=low: the optimizer's stream at the lowest HIP priority — its blocks are dispatched when the compute
        # stream has none waiting, i.e. under the persistent GEMM blocks rather than beside the bandwidth-bound kernels
        prio = os.environ.get("LAP_OPT_PRIORITY", "normal")
        if not self.is_cuda:
            self.side = None
        elif prio in ("low", "high"):
            self.side = hip.stream_with_hip_priority(store.device, prio)
        else:
            self.side = torch.cuda.Stream(device=store.device)
        self.unit_events: dict[str, object] = {}
        self.sumsq = torch.zeros(2, dtype=torch.float32, device=store.device)   # [sharded units, replicated unit]
        self.scal = torch.zeros(8, dtype=torch.float32, device=store.device)
        self.gnorm = torch.zeros((), dtype=torch.float32, device=store.device)
        self._opt_done = None
        self._pending = False
        import os
        # units of the current optimizer pass that are scheduled but not yet enqueued (paced release, see module docstring);
        # lookahead 0 = enqueue the whole pass at once
        self.lookahead = int(os.environ.get("LAP_OPT_LOOKAHEAD", "9"))
        self.pace_in_layer = os.environ.get("LAP_OPT_PACE", "gemm") == "gemm"
        self._todo_args = None
        # single rank: a weight gradient is final when its GEMM stores it, so the assembly weight-gradient kernels add its sum of squares
        # to the norm themselves (lap_gemm_wgrad_f32) and the per-unit pass skips those tensors; under FSDP the norm is that of the
        # REDUCED gradient and stays a pass over the shard.  LAP_FOLD_SUMSQ=0: off (A/B)
        self.fold_sumsq = self.is_cuda and self.world_size == 1 and os.environ.get("LAP_FOLD_SUMSQ", "1") != "0"
        self.folded = set()
        self._bwd_open = False      # a backward has added to `sumsq` and no optimizer pass has consumed it yet (before_backward)
        # release order = the order in which the forward first needs the units: the store's order, except that the adaRMS
        # bank (built last) is needed second — the model issues embed_suffix (on its second stream) ahead of the SigLIP tower
        sched = [u for u in store.units if u.name != "ada"]
        names = [u.name for u in sched]
        if "ada" in store.unit_by_name:
            sched.insert(1 if names and names[0] == "small" else 0, store.unit_by_name["ada"])
        self._sched = sched
        self._unit_pos = {u.name: k for k, u in enumerate(sched)}
        import os
        self.hold_big = os.environ.get("LAP_OPT_HOLD_EMBED", "0") == "1"     # A/B switch (measured: 282.3 vs 282.3 ms: what the tower gains, the wait costs)
        self._first_big = self._unit_pos.get("embed", len(sched))     # (schedule: small, ada, img..., img_head, embed, llm...)
        self._released = len(sched)      # units _sched[0 : _released] of the current pass are enqueued
        store._quiesce = self.synchronize

    # ---- stream helpers (CPU/gloo tests run everything inline)
    def _on_side(self, wait_compute: bool = True):
        if not self.is_cuda:
            return contextlib.nullcontext()
        if wait_compute:
            self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    # ---- step protocol -------------------------------------------------------------------------------------------
    def begin_step(self):
        """Start of a train step: the replicated f32 unit is needed (and its gradient buffer re-zeroed) first."""
        self.wait_unit("small")
        self.folded = set()      # tensors whose weight-gradient GEMM has already added sum(g^2) to the norm (fold_sumsq)

    def before_backward(self):
        """The previous optimizer pass must be done READING the gradient buffers before the backward rewrites them."""
        self.flush()
        if self.is_cuda and self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
            self._opt_done = None
        if self._bwd_open:
            # The previous backward was not followed by an optimizer pass (`LAP.loss_and_grad` used directly: gradient checks, an
            # exception mid-step, a future accumulation loop): its partial sums of squares would inflate the next global norm and
            # clip factor.  run_optimizer is the only place that clears them on the normal path; do it here, ordered
            # behind that backward's last partial on the side stream and in front of this backward's first.
            if self.is_cuda:
                torch.cuda.current_stream().wait_stream(self.side)
            self.sumsq.zero_()
            if self.is_cuda:
                self.side.wait_stream(torch.cuda.current_stream())
            self.folded = set()
        self._bwd_open = True

    def wait_unit(self, name: str, also=None):
        """The current stream (and `also`, the model's second stream) waits until the unit's parameters are the updated ones."""
        if self._released < len(self._sched):
            self._release(self._unit_pos[name] + 1 + (0 if self.pace_in_layer else max(self.lookahead, 1)), paced=True)
        ev = self.unit_events.pop(name, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            if also is not None:
                also.wait_event(ev)

    def pace(self, name: str):
        """Called by the model in front of a unit's longest MFMA-bound GEMM (LAP_OPT_PACE=gemm): the updates of the next
        `lookahead` units are released HERE, so that their HBM traffic runs beside that GEMM instead of beside the unit's first
        (bandwidth-bound) norm kernel."""
        if self.pace_in_layer and self._released < len(self._sched):
            upto = self._unit_pos[name] + 1 + max(self.lookahead, 1)
            if self.hold_big and name.startswith("img"):
                # the SigLIP tower's short kernels crawl beside a big unit's update (its last three blocks took 3.5 ms each instead of 1
                # beside the embedding table's 20 GB): the lookahead stops at the tower's own units; the table is updated with the
                # chip to itself when the compute stream asks for it (wait_unit("embed"))
                upto = min(upto, self._first_big)
            self._release(upto, paced=True)

    def _unfolded(self, u, ranges):
        """`ranges` of the unit's buffer minus the tensors whose sum of squares is already in (single rank: shard = whole unit)"""
        cut = sorted((t.offset, t.offset + t.numel) for t in u.tensors if t.name in self.folded) if self.folded else []
        if not cut:
            return ranges
        out = []
        for a, b in ranges:
            pos = a
            for ca, cb in cut:
                if cb <= pos or ca >= b:
                    continue
                if ca > pos:
                    out.append((pos, ca))
                pos = max(pos, cb)
            if pos < b:
                out.append((pos, b))
        return out

    def _reduce_grads(self, u):
        pass  # single rank: gradients are already complete

    def grads_ready(self, name: str, also=None):
        """The unit's gradients are complete once the current stream (and `also`, the model's weight-gradient stream) get here."""
        u = self.ps.unit_by_name[name]
        if not self.ps.unit_trainable(u):
            return    # fully frozen unit: no reduction, no norm contribution, no update
        if also is not None and self.is_cuda:
            for st in (also if isinstance(also, (list, tuple)) else [also]):
                self.side.wait_stream(st)
        with self._on_side():
            self._reduce_grads(u)
            if self.is_cuda:  # (the CPU/gloo tests exercise the collectives only; kernels need a GPU)
                slot = 0 if (self.ps.sharded(u) or self.world_size == 1) else 1
                g = self.ps.gshard[name]
                for a, b in self._unfolded(u, self.ps.local_train_ranges(u)):     # one range (the whole shard) unless a freeze filter is set
                    hip.sumsq_f32(g[a:b], self.sumsq[slot:slot + 1])
        self._pending = True

    def finish_grads(self):
        """Compute stream waits for every reduction / norm partial (needed for the returned grad_norm)."""
        if self.is_cuda and self._pending:
            torch.cuda.current_stream().wait_stream(self.side)
        self._pending = False

    def _after_unit_update(self, u):
        pass

    def run_optimizer(self, lr, bc1, bc2, ema_decay, ema_on, opt):
        """Side stream: global norm -> per-unit fused update (+ all-gather) in forward order, one event per unit."""
        ps = self.ps
        self.flush()        # (nothing left normally: the backward released the previous pass; its scalars are overwritten below)
        ps.version += 1     # derived copies of the weights (fp8 mirrors) are stale from here on
        h = torch.tensor([0.0, lr, bc1, bc2, ema_decay, 1.0 if ema_on else 0.0, 0.0, 0.0], dtype=torch.float32)
        if self.is_cuda:
            h = h.pin_memory()
        with self._on_side():
            self._reduce_norm()
            self.scal.copy_(h, non_blocking=True)
            self.scal[0:1] = self.sumsq[0:1] + self.sumsq[1:2]
            # the next step's partial sums start from zero HERE: in front of every unit update of this pass on the side stream, so that
            # any stream that has waited for one of this pass's units (all of them do, in the next forward) is ordered behind the
            # clearing — the weight-gradient GEMMs add their share from the compute / weight-gradient streams (fold_sumsq)
            self.sumsq.zero_()
            self._bwd_open = False
            gnorm = self.scal[0].sqrt()      # a fresh tensor per step: callers keep the infos of many steps
            self.gnorm = gnorm
            if self.is_cuda:
                norm_ready = torch.cuda.Event()
                norm_ready.record(self.side)
        # the per-unit updates: build_specs puts the small replicated unit first, then forward order
        self._todo_args = opt
        self._released = 0
        import os
        serial = os.environ.get("LAP_OPT_SERIAL", "0") == "1"     # A/B: the whole pass alone on the chip, the compute stream waits for it
        self._release(self.lookahead if (self.lookahead > 0 and self.is_cuda and not serial) else len(self._sched), paced=False)
        if serial and self.is_cuda and self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
        if self.is_cuda:
            torch.cuda.current_stream().wait_event(norm_ready)   # so that `gnorm` can be read from the compute stream
            gnorm.record_stream(torch.cuda.current_stream())
        return gnorm

    def _release(self, upto: int, paced: bool):
        """Enqueue the updates of units [_released, upto) on the side stream.  paced: they may not start before the compute
        stream has reached this point (the forward of a unit `lookahead` positions earlier)."""
        ps = self.ps
        upto = min(upto, len(self._sched))
        if upto <= self._released:
            return
        opt = self._todo_args
        with self._on_side(wait_compute=paced):
            for u in self._sched[self._released:upto]:
                lo, _ = ps.shard_range(u)
                ema = ps.ema.get(u.name)
                lo16 = ps.lo16.get(u.name)
                for a, b in ps.local_train_ranges(u):   # shard coordinates; frozen tensors are skipped altogether
                    p16 = ps.full16[u.name][lo + a:lo + b] if u.big else None
                    hip.adamw_ema(ps.master[u.name][a:b], ps.m[u.name][a:b], ps.v[u.name][a:b], ema[a:b] if ema is not None else None,
                                  ps.gshard[u.name][a:b], p16, self.scal, opt.b1, opt.b2, opt.eps, opt.weight_decay, opt.clip_gradient_norm,
                                  p16lo=lo16[lo + a:lo + b] if lo16 is not None else None)
                if ps.unit_trainable(u):
                    self._after_unit_update(u)
                if self.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.unit_events[u.name] = ev
            self._released = upto
            if upto == len(self._sched) and self.is_cuda:
                self._opt_done = torch.cuda.Event()
                self._opt_done.record(self.side)

    def flush(self):
        """Enqueue whatever is left of the current optimizer pass (before the backward rewrites the gradient buffers,
        before anything reads the optimizer state on the host side)."""
        self._release(len(self._sched), paced=False)

    def _reduce_norm(self):
        pass

    def param_sumsq(self, select) -> torch.Tensor:
        """Sum of squares of the f32 master values of the tensors `select(name, shape)` accepts, over ALL ranks: every
        rank adds the part of each tensor that lies in its shard, the partial sums are all-reduced (scripts/train.py:401-415
        reports optax.global_norm of the kernel parameters at every logging step, sharded or not)."""
        ps = self.ps
        self.synchronize()
        acc = torch.zeros(2, dtype=torch.float32, device=ps.device)     # [sharded units, replicated unit]
        for u in ps.units:
            lo, hi = ps.shard_range(u)
            slot = 0 if (ps.sharded(u) or self.world_size == 1) else 1
            for t in u.tensors:
                if not select(t.name, t.shape):
                    continue
                a, b = max(t.offset, lo), min(t.offset + t.numel, hi)
                if a < b:
                    v = ps.master[u.name][a - lo:b - lo]
                    if self.is_cuda:
                        hip.sumsq_f32(v, acc[slot:slot + 1])
                    else:
                        acc[slot] += (v.double() ** 2).sum().float()
        acc[1] /= self.world_size       # the replicated unit is counted once
        return self.all_reduce_sum(acc).sum()

    def synchronize(self):
        self.flush()
        if self.is_cuda:
            self.side.synchronize()

    # ---- small collectives (identity on one rank)
    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def sharded_embed_gather(self, *a, **k):
        raise RuntimeError("sharded_embed_gather is only used with world_size > 1")


class FsdpComm(UnitPipeline):
    def __init__(self, store: ParamStore, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        super().__init__(store)
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if store.world_size != self.world_size or store.rank != self.rank:
            raise ValueError("ParamStore world_size/rank do not match the process group")
        # RCCL ("nccl") has the fused tensor collectives; gloo (CPU tests, single-GPU multi-process tests) does not
        self.fused = dist.get_backend(group) == "nccl"
        # In-place all-gather (the input is this rank's slice OF the output buffer) is the documented NCCL / RCCL
        # in-place form and saves a 1/N staging copy per unit; LAP_FSDP_INPLACE_GATHER=0 selects the two-buffer form
        # (gather from a private copy of the slice) should an RCCL build mishandle the aliasing.  bench.py / train.py
        # run with the default (in place).
        import os

        self.inplace_gather = os.environ.get("LAP_FSDP_INPLACE_GATHER", "1") != "0"
        # The assembly GEMM kernels are persistent (one block per CU) but draw their tiles from per-XCD counters, so a collective
        # kernel that owns a few CUs costs them about what it costs the HIP tiles (tools/probes/coresident.py: +10 % with 16 CUs
        # taken, against +70 % for static tile lists); they stay on under RCCL.  LAP_GEMM_NO_ASM=1 is the switch if a SCALE run
        # says otherwise.

    # ---- gradients
    def _reduce_grads(self, u):
        if self.ps.sharded(u):
            g, gs = self.ps.grad[u.name], self.ps.gshard[u.name]
            if g.dtype != gs.dtype:
                # LAP_FSDP_REDUCE_F32=1: the bf16 buffer is widened into one shared f32 staging buffer and the ring sums in f32 (twice
                # the bytes on xGMI).  Default: bf16 on the wire — RCCL's ring adds in f32 but rounds the running sum to bf16 at each of
                # the N - 1 hops, sqrt(N - 1) x 2^-9 / sqrt(3) relative L2 on top of the single rounding the weight-gradient epilogue
                # does (tests/test_fsdp_cpu.py::test_bf16_ring_reduction_error_bound: 3.7e-3 at N = 8 against 1.7e-3 for one rounding).
                # That is also what GSPMD does with the reference's bf16 dot outputs (the partial products of `w.astype(bf16)` are reduced
                # across the batch shards in the dot's own dtype, gemma.py:307,318), so bf16 stays the default.
                if getattr(self, "_stage32", None) is None or self._stage32.numel() < g.numel():
                    self._stage32 = torch.empty(max(self.ps.padded(x) for x in self.ps.units if self.ps.sharded(x) and x.name != "embed"),
                                                dtype=torch.float32, device=g.device)
                st = self._stage32[:g.numel()]
                st.copy_(g)
                g = st
            self._reduce_scatter(gs, g)
        else:
            self.dist.all_reduce(self.ps.grad[u.name], op=self.dist.ReduceOp.SUM, group=self.group)

    def _reduce_norm(self):
        # slot 0: partial sums over the shards -> sum over ranks.  slot 1: the replicated unit's sum, computed by
        # every rank from identical gradients but with rank-dependent atomic ordering (differs in the last ulp):
        # average it over ranks too, so that every rank applies the bit-identical clip factor and replicas never drift.
        self.sumsq[1:2] /= self.world_size
        self.dist.all_reduce(self.sumsq, op=self.dist.ReduceOp.SUM, group=self.group)

    # ---- parameters
    def _after_unit_update(self, u):
        if self.ps.sharded(u):
            full = self.ps.full16[u.name]
            a, b = self.ps.shard_range(u)
            self._all_gather(full, full[a:b])
            lo = self.ps.lo16.get(u.name)      # the embedding table's residual plane travels like the mirror (+1.05 GB per step)
            if lo is not None:
                self._all_gather(lo, lo[a:b])

    def start_param_gather(self):
        """Stand-alone gather of every bf16 mirror (tests / after loading weights)."""
        self.flush()
        with self._on_side():
            for u in self.ps.units:
                self._after_unit_update(u)
                if self.is_cuda and self.ps.sharded(u):
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.unit_events[u.name] = ev

    # ---- backend shims
    def _all_gather(self, full: torch.Tensor, mine: torch.Tensor):
        if self.fused:
            # in place: `mine` is full[rank*n:(rank+1)*n]; the fallback reads it from a detached copy instead
            self.dist.all_gather_into_tensor(full, mine if self.inplace_gather else mine.clone(), group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            self.dist.all_gather(parts, mine.clone(), group=self.group)
            full.copy_(torch.cat(parts))

    def _reduce_scatter(self, out: torch.Tensor, full: torch.Tensor):
        if self.fused:
            self.dist.reduce_scatter_tensor(out, full, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            tmp = full.clone()
            self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM, group=self.group)
            n = out.numel()
            out.copy_(tmp[self.rank * n:(self.rank + 1) * n])

    # ---- small collectives on the compute stream
    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
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
