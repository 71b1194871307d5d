"""Parameter store of the MI355X engine: flat per-unit HBM buffers, FSDP partitioning, and the key map
to/from the reference's parameter tree.

Reference: parameters are a pytree (names in SURVEY.md §8 a1: `PaliGemma/llm/layers/attn/q_einsum/w` with a
leading depth axis from nn.scan, ...), float32, FSDP-sharded per array by `fsdp_sharding`
(src/lap/training/mh_sharding.py:80-100 -> openpi: arrays >= 4 MiB are split on their largest divisible axis,
smaller ones replicated).  Here:

  * every projection is stored as Wt[out][in] (k-contiguous for the forward GEMM, see csrc/gemm.hip) and the
    q|k|v and gate|up projections are packed row-wise so one GEMM produces the fused activation;
  * tensors are grouped in UNITS (one per SigLIP block, one per joint Gemma layer, embedding, adaRMS bank,
    image head).  A unit is one contiguous f32 master buffer with same-shaped Adam m / v / EMA / gradient
    buffers and a bf16 compute mirror: one optimizer launch, one all-gather and one reduce-scatter per unit;
  * everything that is consumed in f32 (norm scales, biases, the f32 SigLIP stem and action head — all
    "< 4 MiB, replicated" arrays of the reference) lives in one small replicated unit.

With world_size N > 1 each rank owns a contiguous 1/N slice of every big unit's master / m / v / EMA (ZeRO-3);
the bf16 mirror is all-gathered on a side stream before use and gradients are reduce-scattered after the
unit's backward (see lap_amd/fsdp.py).
"""
from __future__ import annotations

import dataclasses
import math

import torch

from lap_amd.config import LAPConfig, get_gemma_config, get_siglip_config

ADA_SLOTS_PER_LAYER = 2  # pre_attention_norm_1, pre_ffw_norm_1 ; last slot = final_norm_1


@dataclasses.dataclass
class TensorSpec:
    name: str
    shape: tuple
    init_std: float  # 0 -> zeros ; <0 -> ones + N(0, |std|) is not used: ones handled by `fill`
    fill: float = 0.0
    offset: int = 0
    valid: tuple | None = None   # the reference's extent inside `shape` when the engine pads the tensor with zeros (see siglip_mlp_pad)

    @property
    def numel(self) -> int:
        return math.prod(self.shape)


@dataclasses.dataclass
class UnitSpec:
    name: str
    tensors: list
    big: bool  # True: bf16-consumed matrices (sharded under FSDP); False: f32-consumed, replicated
    numel: int = 0


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def siglip_mlp_pad(mlp_dim: int) -> int:
    """The engine's width of the SigLIP MLP: So400m's 4304 (siglip_gemma3.py:35-37) is 16.8 tiles of 256 as an output and 33.6
    k-tiles of 128 as a contraction, which keeps fc2 forward and both data gradients off the assembly GEMM kernels.  Padded to
    4352 = 17 x 256 with ZERO rows of fc1 (+ zero bias entries) and zero columns of fc2: gelu(0) = 0 feeds zero products into
    fc2 (same f32 sums, k ascending), and every gradient of a padded entry is an exact zero, so AdamW (decay * 0 included)
    leaves them at zero.  The reference tree never sees the padding (reference_to_engine / engine_to_reference).
    LAP_SIGLIP_PAD=0: off (A/B)."""
    import os
    if os.environ.get("LAP_SIGLIP_PAD", "1") == "0" or mlp_dim % 128 == 0 or mlp_dim < 1024:
        return mlp_dim
    return _align(mlp_dim, 256)


def build_specs(cfg: LAPConfig) -> list[UnitSpec]:
    v, e, s = get_gemma_config(cfg.paligemma_variant), get_gemma_config(cfg.action_expert_variant), get_siglip_config(cfg.siglip_variant)
    if (v.depth, v.num_heads, v.num_kv_heads, v.head_dim) != (e.depth, e.num_heads, e.num_kv_heads, e.head_dim):
        raise ValueError("experts must share depth / heads / head_dim (gemma.py:169-171,411)")
    NH, HD, L = v.num_heads, v.head_dim, v.depth
    QKV = (NH + 2 * v.num_kv_heads) * HD
    T = (cfg.image_size // s.patch) ** 2
    pdim = s.patch * s.patch * 3
    mp = siglip_mlp_pad(s.mlp_dim)
    units: list[UnitSpec] = []
    small = [
        TensorSpec("img/stem_w", (s.width, pdim), pdim ** -0.5),
        TensorSpec("img/stem_b", (s.width,), 0.0),
        TensorSpec("img/pos", (T, s.width), s.width ** -0.5),
    ]
    for l in range(s.depth):
        small += [TensorSpec(f"img/{l}/ln1_g", (s.width,), 0.0, 1.0), TensorSpec(f"img/{l}/ln1_b", (s.width,), 0.0),
                  TensorSpec(f"img/{l}/bqkv", (3 * s.width,), 0.0), TensorSpec(f"img/{l}/bo", (s.width,), 0.0),
                  TensorSpec(f"img/{l}/ln2_g", (s.width,), 0.0, 1.0), TensorSpec(f"img/{l}/ln2_b", (s.width,), 0.0),
                  TensorSpec(f"img/{l}/b1", (mp,), 1e-6, valid=(s.mlp_dim,)), TensorSpec(f"img/{l}/b2", (s.width,), 1e-6)]
        units.append(UnitSpec(f"img{l}", [
            TensorSpec(f"img/{l}/wqkv", (3 * s.width, s.width), s.width ** -0.5),
            TensorSpec(f"img/{l}/wo", (s.width, s.width), s.width ** -0.5),
            TensorSpec(f"img/{l}/w1", (mp, s.width), (2.0 / (s.width + s.mlp_dim)) ** 0.5, valid=(s.mlp_dim, s.width)),
            TensorSpec(f"img/{l}/w2", (s.width, mp), (2.0 / (s.width + s.mlp_dim)) ** 0.5, valid=(s.width, s.mlp_dim))], True))
    small += [TensorSpec("img/norm_g", (s.width,), 0.0, 1.0), TensorSpec("img/norm_b", (s.width,), 0.0),
              TensorSpec("img/head_b", (v.width,), 0.0)]
    units.append(UnitSpec("img_head", [TensorSpec("img/head_w", (v.width, s.width), s.width ** -0.5)], True))
    units.append(UnitSpec("embed", [TensorSpec("llm/embed", (cfg.vocab_size, v.width), 0.01)], True))
    for l in range(L):
        small += [TensorSpec(f"llm/{l}/n_attn", (v.width,), 0.0), TensorSpec(f"llm/{l}/n_ffw", (v.width,), 0.0)]
        if not cfg.pi05:       # pi0: the expert's norms are plain RMSNorms (`use_adarms=[False, False]`, lap.py:51)
            small += [TensorSpec(f"llm/{l}/n_attn1", (e.width,), 0.0), TensorSpec(f"llm/{l}/n_ffw1", (e.width,), 0.0)]
        units.append(UnitSpec(f"llm{l}", [
            TensorSpec(f"llm/{l}/wqkv0", (QKV, v.width), v.width ** -0.5),
            TensorSpec(f"llm/{l}/wo0", (v.width, NH * HD), (NH * HD) ** -0.5),
            TensorSpec(f"llm/{l}/wgu0", (2 * v.mlp_dim, v.width), v.width ** -0.5),
            TensorSpec(f"llm/{l}/wd0", (v.width, v.mlp_dim), v.mlp_dim ** -0.5),
            TensorSpec(f"llm/{l}/wqkv1", (QKV, e.width), e.width ** -0.5),
            TensorSpec(f"llm/{l}/wo1", (e.width, NH * HD), (NH * HD) ** -0.5),
            TensorSpec(f"llm/{l}/wgu1", (2 * e.mlp_dim, e.width), e.width ** -0.5),
            TensorSpec(f"llm/{l}/wd1", (e.width, e.mlp_dim), e.mlp_dim ** -0.5)], True))
    nslots = ADA_SLOTS_PER_LAYER * L + 1
    small += [TensorSpec("llm/final_norm", (v.width,), 0.0)]
    ad, w = cfg.action_dim, e.width
    small += [TensorSpec("act/in_w", (w, ad), ad ** -0.5), TensorSpec("act/in_b", (w,), 0.0)]
    if cfg.pi05:
        small += [TensorSpec("ada/b", (nslots * 3 * e.width,), 0.0)]
        units.append(UnitSpec("ada", [TensorSpec("ada/w", (nslots * 3 * e.width, e.width), 0.0)], True))  # zero-init (gemma.py:128)
        small += [TensorSpec("act/time_in_w", (w, w), w ** -0.5), TensorSpec("act/time_in_b", (w,), 0.0),
                  TensorSpec("act/time_out_w", (w, w), w ** -0.5), TensorSpec("act/time_out_b", (w,), 0.0)]
    else:       # pi0 (lap.py:56-61): state token + action/time MLP, no adaRMS bank
        small += [TensorSpec("llm/final_norm1", (e.width,), 0.0),
                  TensorSpec("act/state_w", (w, ad), ad ** -0.5), TensorSpec("act/state_b", (w,), 0.0),
                  TensorSpec("act/atime_in_w", (w, 2 * w), (2 * w) ** -0.5), TensorSpec("act/atime_in_b", (w,), 0.0),
                  TensorSpec("act/atime_out_w", (w, w), w ** -0.5), TensorSpec("act/atime_out_b", (w,), 0.0)]
    small += [TensorSpec("act/out_w", (ad, w), w ** -0.5), TensorSpec("act/out_b", (ad,), 0.0)]
    units.insert(0, UnitSpec("small", small, False))
    for u in units:
        off = 0
        for t in u.tensors:
            t.offset = off
            off += _align(t.numel)
        u.numel = off
    return units


class ParamStore:
    """Device-resident parameters / optimizer state / gradients as flat per-unit buffers."""

    def __init__(self, cfg: LAPConfig, device="cuda", *, world_size=1, rank=0, with_optimizer=True, with_ema=True,
                 with_grads=True):
        self.cfg = cfg
        self.device = torch.device(device)
        self.world_size, self.rank = world_size, rank
        self.units = build_specs(cfg)
        self.unit_by_name = {u.name: u for u in self.units}
        self.tensor_unit = {t.name: u for u in self.units for t in u.tensors}
        self.tensor_spec = {t.name: t for u in self.units for t in u.tensors}
        self.full16: dict[str, torch.Tensor] = {}   # unit -> bf16 mirror [padded numel] (big units)
        # unit -> bf16 residual plane, lo = bf16(master - float(mirror)): only the embedding table has one.  The LM head multiplies
        # by the F32 table (gemma.py:153-154); hi + lo carries 16 of its mantissa bits into two bf16 MFMA products (model.py)
        self.lo16: dict[str, torch.Tensor] = {}
        self.master: dict[str, torch.Tensor] = {}   # unit -> f32 master (this rank's shard for big units if N>1)
        self.m: dict[str, torch.Tensor] = {}
        self.v: dict[str, torch.Tensor] = {}
        self.ema: dict[str, torch.Tensor] = {}
        self.grad: dict[str, torch.Tensor] = {}     # unit -> full gradient buffer [padded numel], f32 or bf16 (grad_dtype)
        self.gshard: dict[str, torch.Tensor] = {}   # unit -> gradient shard of the same dtype (aliases grad when N == 1)
        self.version = 0                            # bumped whenever parameter values change (derived copies re-quantise)
        self.frozen: dict[str, bool] = {}           # engine tensor -> excluded from gradient / optimizer (set_frozen)
        self._train_ranges: dict[str, list] = {}    # unit -> [(a, b)] trainable ranges in unit coordinates
        self._quiesce = None                        # set by the step pipeline (UnitPipeline.synchronize), see quiesce()
        for u in self.units:
            n = self.padded(u)
            sh = self.shard_numel(u)
            z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt, device=self.device)
            self.master[u.name] = z(sh)
            if u.big:
                self.full16[u.name] = z(n, torch.bfloat16)
                if u.name == "embed":
                    self.lo16[u.name] = z(n, torch.bfloat16)
            if with_grads:
                gdt = self.grad_dtype(u)
                self.grad[u.name] = z(n, gdt)
                # LAP_FSDP_REDUCE_F32=1: bf16 gradient buffers are widened before the reduce-scatter and summed in f32 (a bf16
                # ring rounds the partial sum at every hop; see fsdp.FsdpComm._reduce_grads), so the shard the optimizer reads is f32
                self.gshard[u.name] = self.grad[u.name] if self.sharded(u) is False else z(sh, torch.float32 if self.reduce_f32() else gdt)
            if with_optimizer:
                self.m[u.name], self.v[u.name] = z(sh), z(sh)
            if with_ema:
                self.ema[u.name] = z(sh)

    # ---- geometry
    @staticmethod
    def grad_dtype(u: UnitSpec) -> torch.dtype:
        """bf16 for the units whose tensors are all weights of bf16 GEMMs with ONE weight-gradient product each (SigLIP blocks, image
        head, joint Gemma layers, the adaRMS bank): the reference multiplies `w.astype(bf16)` (gemma.py:307,318; lora.Einsum; Flax
        Dense with dtype=bf16), so the cotangent that reaches its f32 master is a bf16-rounded product — here the weight-gradient
        GEMM's epilogue rounds once and stores 2 bytes.  -11.3 GB of HBM traffic per train step at N = 1 (the GEMMs' stores and
        the optimizer's reads) and half the FSDP reduce-scatter volume.  f32 stays where gradients are ACCUMULATED: the embedding
        table (LM-head product + scatter-add of the prompt tokens' rows) and the small replicated unit (atomics).
        LAP_GRAD_BF16=0: f32 everywhere (A/B runs, the pre-round-5 layout)."""
        import os
        if u.big and u.name != "embed" and os.environ.get("LAP_GRAD_BF16", "1") != "0":
            return torch.bfloat16
        return torch.float32

    @staticmethod
    def reduce_f32() -> bool:
        import os
        return os.environ.get("LAP_FSDP_REDUCE_F32", "0") == "1"

    def sharded(self, u: UnitSpec) -> bool:
        return u.big and self.world_size > 1

    def padded(self, u: UnitSpec) -> int:
        q = 64 * self.world_size
        if u.name == "embed":  # shard boundaries on whole vocabulary rows (sharded gather, lap_amd/fsdp.py)
            q = self.world_size * u.tensors[0].shape[1]
        return (u.numel + q - 1) // q * q

    def shard_numel(self, u: UnitSpec) -> int:
        return self.padded(u) // self.world_size if self.sharded(u) else self.padded(u)

    def shard_range(self, u: UnitSpec) -> tuple[int, int]:
        if not self.sharded(u):
            return 0, self.padded(u)
        sh = self.shard_numel(u)
        return self.rank * sh, (self.rank + 1) * sh

    # ---- trainable / frozen partition (openpi TrainConfig.freeze_filter, scripts/train.py:225-240,358-363)
    def set_frozen(self, is_frozen=None):
        """`is_frozen`: predicate over the REFERENCE's parameter paths (None: everything trainable).  An engine tensor
        packs one or more reference arrays (`engine_sources`); all of them must fall on the same side.  Frozen tensors
        are kept out of the optimizer, the gradient norm and the weight-gradient GEMMs, and their values are rounded to
        bfloat16 as the reference stores them (train.py:225-231)."""
        self.frozen = {}
        src = engine_sources(self.cfg)
        for name in self.tensor_spec:
            flags = {bool(is_frozen(k)) for k in src[name]} if is_frozen is not None else {False}
            if len(flags) != 1:
                raise ValueError(f"freeze filter splits engine tensor {name} (packed from {src[name]})")
            self.frozen[name] = flags.pop()
        self._train_ranges = {}
        for u in self.units:
            ranges = []
            for t in u.tensors:
                if self.frozen[t.name]:
                    continue
                a, b = t.offset, t.offset + _align(t.numel)
                if ranges and ranges[-1][1] == a:
                    ranges[-1] = (ranges[-1][0], b)
                else:
                    ranges.append((a, b))
            if ranges and ranges[-1][1] == u.numel:        # the unit's tail padding belongs to its last tensor
                ranges[-1] = (ranges[-1][0], self.padded(u))
            self._train_ranges[u.name] = ranges
            lo, hi = self.shard_range(u)
            for t in u.tensors:                               # frozen values live at bf16 precision
                if self.frozen[t.name]:
                    a, b = max(t.offset, lo), min(t.offset + t.numel, hi)
                    if a < b:
                        v = self.master[u.name][a - lo:b - lo]
                        v.copy_(v.to(torch.bfloat16).to(torch.float32))
        if any(self.frozen.values()):
            self.sync_ema_from_master()

    def is_trainable(self, name: str) -> bool:
        return not self.frozen.get(name, False)

    def unit_trainable(self, u: UnitSpec) -> bool:
        return bool(self._train_ranges.get(u.name, [(0, 1)]))

    def local_train_ranges(self, u: UnitSpec) -> list[tuple[int, int]]:
        """Trainable ranges of `u` intersected with this rank's shard, in SHARD coordinates."""
        lo, hi = self.shard_range(u)
        out = []
        for a, b in self._train_ranges.get(u.name, [(0, self.padded(u))]):
            a, b = max(a, lo), min(b, hi)
            if a < b:
                out.append((a - lo, b - lo))
        return out

    # ---- views
    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        t = self.tensor_spec[name]
        return buf[t.offset:t.offset + t.numel].view(t.shape)

    def w16(self, name: str) -> torch.Tensor:
        """bf16 compute view of a big-unit matrix (valid after the unit has been gathered)."""
        return self._view(self.full16[self.tensor_unit[name].name], name)

    def f32(self, name: str) -> torch.Tensor:
        """f32 view of a replicated (small unit) tensor, or of a big tensor when world_size == 1."""
        u = self.tensor_unit[name]
        if self.sharded(u):
            raise RuntimeError(f"{name} is sharded; no full f32 view")
        return self._view(self.master[u.name], name)

    def embed_rows(self) -> tuple[torch.Tensor, int, int]:
        """(f32 rows [lo, hi) of the embedding table held by this rank, lo, hi) — gemma.py:148-151 gathers from the
        f32 parameter; under FSDP every rank looks up the rows it owns (lap_amd/fsdp.py)."""
        u = self.tensor_unit["llm/embed"]
        D = u.tensors[0].shape[1]
        a, b = self.shard_range(u)
        return self.master[u.name].view(-1, D), a // D, b // D

    def g(self, name: str) -> torch.Tensor:
        return self._view(self.grad[self.tensor_unit[name].name], name)

    def names(self):
        return list(self.tensor_spec)

    def numel(self) -> int:
        return sum(t.numel for t in self.tensor_spec.values())

    # ---- initialisation / import / export
    def _write_full(self, u: UnitSpec, full: torch.Tensor):
        """full: f32 [padded] on device -> master shard (+ bf16 mirror)."""
        a, b = self.shard_range(u)
        self.version += 1
        self.master[u.name].copy_(full[a:b])
        if u.big:
            self.full16[u.name].copy_(full)  # dtype cast
            if u.name in self.lo16:
                self.lo16[u.name].copy_(full - self.full16[u.name].to(torch.float32))

    def init_random(self, seed: int = 0):
        """Reference initialisers in distribution family (lecun-normal GEMMs, zeros for norm scales / adaRMS,
        N(0, 0.01) embedding; gemma.py:121,128,143-146,183-198,305-317; siglip_gemma3.py:49-54,66-69)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for u in self.units:
            full = torch.zeros(self.padded(u), dtype=torch.float32, device=self.device)
            for t in u.tensors:
                v = full[t.offset:t.offset + t.numel]
                if t.init_std > 0 and t.valid is not None and tuple(t.valid) != tuple(t.shape):
                    # (the same draws as the unpadded tensor would get; the padding stays zero)
                    tmp = torch.empty(t.valid, dtype=torch.float32, device=self.device).normal_(0.0, t.init_std, generator=g)
                    v.view(t.shape)[tuple(slice(0, n) for n in t.valid)] = tmp
                elif t.init_std > 0:
                    v.normal_(0.0, t.init_std, generator=g)
                elif t.fill:
                    v.fill_(t.fill)
            self._write_full(u, full)
            del full
        self.sync_ema_from_master()

    def quiesce(self):
        """Host-side readers / writers of the state first let the step pipeline finish: its optimizer pass is released
        unit by unit while the next forward runs (lap_amd/fsdp.py), so after a train step part of it may not even be
        enqueued yet.  Collective under FSDP (every rank calls the readers below together)."""
        if self._quiesce is not None:
            self._quiesce()

    def sync_ema_from_master(self):
        self.quiesce()
        for k, e in self.ema.items():
            e.copy_(self.master[k])

    def refresh_mirror_local(self):
        """bf16 mirror <- f32 master for world_size == 1 (the optimizer kernel does this itself each step)."""
        self.version += 1     # derived copies of the weights (fp8 mirrors) are stale from here on
        for u in self.units:
            if u.big and not self.sharded(u):
                self.full16[u.name].copy_(self.master[u.name])
        self.refresh_lo_shard()

    def refresh_lo_shard(self):
        """Residual plane of this rank's slice of the embedding table from its f32 master slice and the bf16 mirror (after the
        masters were written directly: checkpoint restore).  Under FSDP the caller gathers the plane like the mirror."""
        for name, lo in self.lo16.items():
            a, b = self.shard_range(self.unit_by_name[name])
            lo[a:b].copy_(self.master[name] - self.full16[name][a:b].to(torch.float32))

    def w16lo(self, name: str) -> torch.Tensor | None:
        """The bf16 residual plane of a matrix that has one (the embedding table), or None."""
        u = self.tensor_unit[name]
        return self._view(self.lo16[u.name], name) if u.name in self.lo16 else None

    def load_reference_tree(self, P: dict):
        """Fill from a parameter tree in the reference's names / layouts (see module docstring and
        `reference_key_map`).  P values: torch tensors or numpy arrays."""
        self.quiesce()
        eng = reference_to_engine(self.cfg, P)
        missing = set(self.tensor_spec) - set(eng)
        if missing:
            raise KeyError(f"reference tree is missing tensors for: {sorted(missing)[:5]} ...")
        for u in self.units:
            full = torch.zeros(self.padded(u), dtype=torch.float32, device=self.device)
            for t in u.tensors:
                src = eng[t.name]
                if tuple(src.shape) != tuple(t.shape):
                    raise ValueError(f"{t.name}: shape {tuple(src.shape)} != expected {t.shape}")
                full[t.offset:t.offset + t.numel].copy_(src.reshape(-1).to(torch.float32))
            self._write_full(u, full)
        self.sync_ema_from_master()

    def gather_master_full(self, which: str = "master") -> dict[str, torch.Tensor]:
        """Engine-layout f32 tensors (CPU) from master or EMA; all-gathers shards when world_size > 1."""
        import torch.distributed as dist

        self.quiesce()
        src = {"master": self.master, "ema": self.ema, "m": self.m, "v": self.v}[which]
        out = {}
        for u in self.units:
            buf = src[u.name]
            if self.sharded(u):
                if dist.get_backend() == "nccl":
                    full = torch.empty(self.padded(u), dtype=torch.float32, device=self.device)
                    dist.all_gather_into_tensor(full, buf)
                else:   # gloo (CPU tests) has no fused tensor collective
                    parts = [torch.empty_like(buf) for _ in range(self.world_size)]
                    dist.all_gather(parts, buf.contiguous())
                    full = torch.cat(parts)
                buf = full
            for t in u.tensors:
                out[t.name] = buf[t.offset:t.offset + t.numel].view(t.shape).cpu()
        return out

    def to_reference_tree(self, which: str = "master") -> dict[str, torch.Tensor]:
        return engine_to_reference(self.cfg, self.gather_master_full(which))


# ================================================================================= reference key map
def engine_sources(cfg: LAPConfig) -> dict[str, list[str]]:
    """Engine tensor -> the reference arrays it is packed from (the inverse view of `reference_to_engine`)."""
    v, e, s = _cfgs(cfg)
    blk = "PaliGemma/img/Transformer/encoderblock"
    mha = f"{blk}/MultiHeadDotProductAttention_0"
    lay = "PaliGemma/llm/layers"
    out = {"img/stem_w": ["PaliGemma/img/embedding/kernel"], "img/stem_b": ["PaliGemma/img/embedding/bias"],
           "img/pos": ["PaliGemma/img/pos_embedding"],
           "img/norm_g": ["PaliGemma/img/Transformer/encoder_norm/scale"], "img/norm_b": ["PaliGemma/img/Transformer/encoder_norm/bias"],
           "img/head_w": ["PaliGemma/img/head/kernel"], "img/head_b": ["PaliGemma/img/head/bias"],
           "llm/embed": ["PaliGemma/llm/embedder/input_embedding"], "llm/final_norm": ["PaliGemma/llm/final_norm/scale"]}
    for l in range(s.depth):
        out[f"img/{l}/ln1_g"], out[f"img/{l}/ln1_b"] = [f"{blk}/LayerNorm_0/scale"], [f"{blk}/LayerNorm_0/bias"]
        out[f"img/{l}/ln2_g"], out[f"img/{l}/ln2_b"] = [f"{blk}/LayerNorm_1/scale"], [f"{blk}/LayerNorm_1/bias"]
        out[f"img/{l}/wqkv"] = [f"{mha}/{n}/kernel" for n in ("query", "key", "value")]
        out[f"img/{l}/bqkv"] = [f"{mha}/{n}/bias" for n in ("query", "key", "value")]
        out[f"img/{l}/wo"], out[f"img/{l}/bo"] = [f"{mha}/out/kernel"], [f"{mha}/out/bias"]
        out[f"img/{l}/w1"], out[f"img/{l}/b1"] = [f"{blk}/MlpBlock_0/Dense_0/kernel"], [f"{blk}/MlpBlock_0/Dense_0/bias"]
        out[f"img/{l}/w2"], out[f"img/{l}/b2"] = [f"{blk}/MlpBlock_0/Dense_1/kernel"], [f"{blk}/MlpBlock_0/Dense_1/bias"]
    if cfg.pi05:
        ada = [f"{lay}/{nm}/Dense_0" for nm in ("pre_attention_norm_1", "pre_ffw_norm_1")] + ["PaliGemma/llm/final_norm_1/Dense_0"]
        out["ada/w"], out["ada/b"] = [a + "/kernel" for a in ada], [a + "/bias" for a in ada]
    else:
        out["llm/final_norm1"] = ["PaliGemma/llm/final_norm_1/scale"]
    for l in range(v.depth):
        for i in range(2):
            sfx = "" if i == 0 else f"_{i}"
            out[f"llm/{l}/wqkv{i}"] = [f"{lay}/attn/q_einsum{sfx}/w", f"{lay}/attn/kv_einsum{sfx}/w"]
            out[f"llm/{l}/wo{i}"] = [f"{lay}/attn/attn_vec_einsum{sfx}/w"]
            out[f"llm/{l}/wgu{i}"] = [f"{lay}/mlp{sfx}/gating_einsum"]
            out[f"llm/{l}/wd{i}"] = [f"{lay}/mlp{sfx}/linear"]
        out[f"llm/{l}/n_attn"], out[f"llm/{l}/n_ffw"] = [f"{lay}/pre_attention_norm/scale"], [f"{lay}/pre_ffw_norm/scale"]
        if not cfg.pi05:
            out[f"llm/{l}/n_attn1"], out[f"llm/{l}/n_ffw1"] = [f"{lay}/pre_attention_norm_1/scale"], [f"{lay}/pre_ffw_norm_1/scale"]
    heads = (("time_in", "time_mlp_in"), ("time_out", "time_mlp_out")) if cfg.pi05 else \
        (("state", "state_proj"), ("atime_in", "action_time_mlp_in"), ("atime_out", "action_time_mlp_out"))
    for nm, ref in (("in", "action_in_proj"), *heads, ("out", "action_out_proj")):
        out[f"act/{nm}_w"], out[f"act/{nm}_b"] = [f"{ref}/kernel"], [f"{ref}/bias"]
    return out


def _is_action_expert_key(k: str) -> bool:
    """Reference parameters that exist only with `enable_action_training` (lap.py:40-62: the second Gemma expert's `_1` arrays, the
    adaRMS Dense layers, the action / time projections); without it the reference builds `gemma.Module(configs=[paligemma])` alone."""
    return k.startswith(("action_in_proj/", "action_out_proj/", "time_mlp_in/", "time_mlp_out/", "state_proj/", "action_time_mlp_in/",
                         "action_time_mlp_out/")) or (k.startswith("PaliGemma/llm/") and "_1/" in k)


def reference_shapes(cfg: LAPConfig) -> dict[str, tuple]:
    """Shapes of the reference's parameter tree, derived from the engine specs through the key map
    (shape-only `meta` tensors: nothing is allocated)."""
    E = {t.name: torch.empty(t.shape, dtype=torch.float32, device="meta") for u in build_specs(cfg) for t in u.tensors}
    return {k: tuple(v.shape) for k, v in engine_to_reference(cfg, E).items()}


def _cfgs(cfg):
    return get_gemma_config(cfg.paligemma_variant), get_gemma_config(cfg.action_expert_variant), get_siglip_config(cfg.siglip_variant)


def reference_to_engine(cfg: LAPConfig, P: dict) -> dict[str, torch.Tensor]:
    """Reference tree -> engine tensors.  Pure layout transforms, f32."""
    full_shapes = None

    def T(k):
        nonlocal full_shapes
        if k not in P and not cfg.enable_action_training and _is_action_expert_key(k):
            # (the reference has no action expert without action training: the engine's expert tensors are never read there — zeros)
            if full_shapes is None:
                E0 = {t.name: torch.empty(t.shape, dtype=torch.float32, device="meta") for u in build_specs(cfg) for t in u.tensors}
                full_shapes = {kk: tuple(vv.shape) for kk, vv in _engine_to_reference_full(cfg, E0).items()}
            return torch.zeros(full_shapes[k], dtype=torch.float32)
        x = P[k]
        return torch.as_tensor(x).to(torch.float32)

    v, e, s = _cfgs(cfg)
    L, NH, HD = v.depth, v.num_heads, v.head_dim
    out = {}
    pdim = s.patch * s.patch * 3
    mp = siglip_mlp_pad(s.mlp_dim)
    out["img/stem_w"] = T("PaliGemma/img/embedding/kernel").reshape(pdim, s.width).t().contiguous()
    out["img/stem_b"] = T("PaliGemma/img/embedding/bias")
    out["img/pos"] = T("PaliGemma/img/pos_embedding")[0]
    blk = "PaliGemma/img/Transformer/encoderblock"
    mha = f"{blk}/MultiHeadDotProductAttention_0"
    for l in range(s.depth):
        out[f"img/{l}/ln1_g"], out[f"img/{l}/ln1_b"] = T(f"{blk}/LayerNorm_0/scale")[l], T(f"{blk}/LayerNorm_0/bias")[l]
        out[f"img/{l}/ln2_g"], out[f"img/{l}/ln2_b"] = T(f"{blk}/LayerNorm_1/scale")[l], T(f"{blk}/LayerNorm_1/bias")[l]
        out[f"img/{l}/wqkv"] = torch.cat([T(f"{mha}/{n}/kernel")[l].reshape(s.width, s.width).t() for n in ("query", "key", "value")], 0).contiguous()
        out[f"img/{l}/bqkv"] = torch.cat([T(f"{mha}/{n}/bias")[l].reshape(-1) for n in ("query", "key", "value")], 0)
        out[f"img/{l}/wo"] = T(f"{mha}/out/kernel")[l].reshape(s.width, s.width).t().contiguous()
        out[f"img/{l}/bo"] = T(f"{mha}/out/bias")[l]
        padz = mp - s.mlp_dim          # zero rows of fc1 / zero bias entries / zero columns of fc2 (siglip_mlp_pad)
        out[f"img/{l}/w1"] = torch.nn.functional.pad(T(f"{blk}/MlpBlock_0/Dense_0/kernel")[l].t(), (0, 0, 0, padz)).contiguous()
        out[f"img/{l}/b1"] = torch.nn.functional.pad(T(f"{blk}/MlpBlock_0/Dense_0/bias")[l], (0, padz))
        out[f"img/{l}/w2"] = torch.nn.functional.pad(T(f"{blk}/MlpBlock_0/Dense_1/kernel")[l].t(), (0, padz)).contiguous()
        out[f"img/{l}/b2"] = T(f"{blk}/MlpBlock_0/Dense_1/bias")[l]
    out["img/norm_g"], out["img/norm_b"] = T("PaliGemma/img/Transformer/encoder_norm/scale"), T("PaliGemma/img/Transformer/encoder_norm/bias")
    out["img/head_w"] = T("PaliGemma/img/head/kernel").t().contiguous()
    out["img/head_b"] = T("PaliGemma/img/head/bias")
    out["llm/embed"] = T("PaliGemma/llm/embedder/input_embedding")
    lay = "PaliGemma/llm/layers"
    ada_w, ada_b = [], []
    for l in range(L):
        for i, c in enumerate((v, e)):
            sfx = "" if i == 0 else f"_{i}"
            q = T(f"{lay}/attn/q_einsum{sfx}/w")[l].permute(0, 2, 1).reshape(NH * HD, c.width)
            kv = T(f"{lay}/attn/kv_einsum{sfx}/w")[l]  # [2, K, D, H]
            k = kv[0].permute(0, 2, 1).reshape(-1, c.width)
            vv = kv[1].permute(0, 2, 1).reshape(-1, c.width)
            out[f"llm/{l}/wqkv{i}"] = torch.cat([q, k, vv], 0).contiguous()
            out[f"llm/{l}/wo{i}"] = T(f"{lay}/attn/attn_vec_einsum{sfx}/w")[l].reshape(NH * HD, c.width).t().contiguous()
            ge = T(f"{lay}/mlp{sfx}/gating_einsum")[l]
            out[f"llm/{l}/wgu{i}"] = torch.cat([ge[0].t(), ge[1].t()], 0).contiguous()
            out[f"llm/{l}/wd{i}"] = T(f"{lay}/mlp{sfx}/linear")[l].t().contiguous()
        out[f"llm/{l}/n_attn"] = T(f"{lay}/pre_attention_norm/scale")[l]
        out[f"llm/{l}/n_ffw"] = T(f"{lay}/pre_ffw_norm/scale")[l]
        if not cfg.pi05:
            out[f"llm/{l}/n_attn1"] = T(f"{lay}/pre_attention_norm_1/scale")[l]
            out[f"llm/{l}/n_ffw1"] = T(f"{lay}/pre_ffw_norm_1/scale")[l]
            continue
        for nm in ("pre_attention_norm_1", "pre_ffw_norm_1"):
            ada_w.append(T(f"{lay}/{nm}/Dense_0/kernel")[l].t())
            ada_b.append(T(f"{lay}/{nm}/Dense_0/bias")[l])
    out["llm/final_norm"] = T("PaliGemma/llm/final_norm/scale")
    out["act/in_w"], out["act/in_b"] = T("action_in_proj/kernel").t().contiguous(), T("action_in_proj/bias")
    if cfg.pi05:
        ada_w.append(T("PaliGemma/llm/final_norm_1/Dense_0/kernel").t())
        ada_b.append(T("PaliGemma/llm/final_norm_1/Dense_0/bias"))
        out["ada/w"] = torch.cat(ada_w, 0).contiguous()
        out["ada/b"] = torch.cat(ada_b, 0)
        out["act/time_in_w"], out["act/time_in_b"] = T("time_mlp_in/kernel").t().contiguous(), T("time_mlp_in/bias")
        out["act/time_out_w"], out["act/time_out_b"] = T("time_mlp_out/kernel").t().contiguous(), T("time_mlp_out/bias")
    else:
        out["llm/final_norm1"] = T("PaliGemma/llm/final_norm_1/scale")
        out["act/state_w"], out["act/state_b"] = T("state_proj/kernel").t().contiguous(), T("state_proj/bias")
        out["act/atime_in_w"], out["act/atime_in_b"] = T("action_time_mlp_in/kernel").t().contiguous(), T("action_time_mlp_in/bias")
        out["act/atime_out_w"], out["act/atime_out_b"] = T("action_time_mlp_out/kernel").t().contiguous(), T("action_time_mlp_out/bias")
    out["act/out_w"], out["act/out_b"] = T("action_out_proj/kernel").t().contiguous(), T("action_out_proj/bias")
    return out


def engine_to_reference(cfg: LAPConfig, E: dict) -> dict[str, torch.Tensor]:
    """Inverse of reference_to_engine (checkpoint export: the `params` item of the reference's layout).  Without action training
    (`vla0_*` configs) the reference's tree has no action expert (lap.py:64-74): its keys are left out."""
    P = _engine_to_reference_full(cfg, E)
    if not cfg.enable_action_training:
        P = {k: v for k, v in P.items() if not _is_action_expert_key(k)}
    return P


def _engine_to_reference_full(cfg: LAPConfig, E: dict) -> dict[str, torch.Tensor]:
    v, e, s = _cfgs(cfg)
    L, NH, HD, KV = v.depth, v.num_heads, v.head_dim, v.num_kv_heads
    hd = s.width // s.num_heads
    P = {}
    P["PaliGemma/img/embedding/kernel"] = E["img/stem_w"].t().reshape(s.patch, s.patch, 3, s.width).contiguous()
    P["PaliGemma/img/embedding/bias"] = E["img/stem_b"]
    P["PaliGemma/img/pos_embedding"] = E["img/pos"][None]
    blk = "PaliGemma/img/Transformer/encoderblock"
    mha = f"{blk}/MultiHeadDotProductAttention_0"
    st = lambda f: torch.stack([f(l) for l in range(s.depth)], 0)
    P[f"{blk}/LayerNorm_0/scale"], P[f"{blk}/LayerNorm_0/bias"] = st(lambda l: E[f"img/{l}/ln1_g"]), st(lambda l: E[f"img/{l}/ln1_b"])
    P[f"{blk}/LayerNorm_1/scale"], P[f"{blk}/LayerNorm_1/bias"] = st(lambda l: E[f"img/{l}/ln2_g"]), st(lambda l: E[f"img/{l}/ln2_b"])
    for j, n in enumerate(("query", "key", "value")):
        P[f"{mha}/{n}/kernel"] = st(lambda l: E[f"img/{l}/wqkv"][j * s.width:(j + 1) * s.width].t().reshape(s.width, s.num_heads, hd))
        P[f"{mha}/{n}/bias"] = st(lambda l: E[f"img/{l}/bqkv"][j * s.width:(j + 1) * s.width].reshape(s.num_heads, hd))
    P[f"{mha}/out/kernel"] = st(lambda l: E[f"img/{l}/wo"].t().reshape(s.num_heads, hd, s.width))
    P[f"{mha}/out/bias"] = st(lambda l: E[f"img/{l}/bo"])
    md = s.mlp_dim      # (the engine's zero padding of the MLP width stays behind: siglip_mlp_pad)
    P[f"{blk}/MlpBlock_0/Dense_0/kernel"], P[f"{blk}/MlpBlock_0/Dense_0/bias"] = st(lambda l: E[f"img/{l}/w1"][:md].t()), st(lambda l: E[f"img/{l}/b1"][:md])
    P[f"{blk}/MlpBlock_0/Dense_1/kernel"], P[f"{blk}/MlpBlock_0/Dense_1/bias"] = st(lambda l: E[f"img/{l}/w2"][:, :md].t()), st(lambda l: E[f"img/{l}/b2"])
    P["PaliGemma/img/Transformer/encoder_norm/scale"], P["PaliGemma/img/Transformer/encoder_norm/bias"] = E["img/norm_g"], E["img/norm_b"]
    P["PaliGemma/img/head/kernel"], P["PaliGemma/img/head/bias"] = E["img/head_w"].t().contiguous(), E["img/head_b"]
    P["PaliGemma/llm/embedder/input_embedding"] = E["llm/embed"]
    lay = "PaliGemma/llm/layers"
    sl = lambda f: torch.stack([f(l) for l in range(L)], 0)
    for i, c in enumerate((v, e)):
        sfx = "" if i == 0 else f"_{i}"
        P[f"{lay}/attn/q_einsum{sfx}/w"] = sl(lambda l: E[f"llm/{l}/wqkv{i}"][:NH * HD].reshape(NH, HD, c.width).permute(0, 2, 1))
        P[f"{lay}/attn/kv_einsum{sfx}/w"] = sl(lambda l: torch.stack([
            E[f"llm/{l}/wqkv{i}"][NH * HD:(NH + KV) * HD].reshape(KV, HD, c.width).permute(0, 2, 1),
            E[f"llm/{l}/wqkv{i}"][(NH + KV) * HD:].reshape(KV, HD, c.width).permute(0, 2, 1)], 0))
        P[f"{lay}/attn/attn_vec_einsum{sfx}/w"] = sl(lambda l: E[f"llm/{l}/wo{i}"].t().reshape(NH, HD, c.width))
        P[f"{lay}/mlp{sfx}/gating_einsum"] = sl(lambda l: torch.stack([E[f"llm/{l}/wgu{i}"][:c.mlp_dim].t(), E[f"llm/{l}/wgu{i}"][c.mlp_dim:].t()], 0))
        P[f"{lay}/mlp{sfx}/linear"] = sl(lambda l: E[f"llm/{l}/wd{i}"].t())
    P[f"{lay}/pre_attention_norm/scale"] = sl(lambda l: E[f"llm/{l}/n_attn"])
    P[f"{lay}/pre_ffw_norm/scale"] = sl(lambda l: E[f"llm/{l}/n_ffw"])
    W3 = 3 * e.width
    P["PaliGemma/llm/final_norm/scale"] = E["llm/final_norm"]
    P["action_in_proj/kernel"], P["action_in_proj/bias"] = E["act/in_w"].t().contiguous(), E["act/in_b"]
    if cfg.pi05:
        for j, nm in enumerate(("pre_attention_norm_1", "pre_ffw_norm_1")):
            P[f"{lay}/{nm}/Dense_0/kernel"] = sl(lambda l: E["ada/w"][(2 * l + j) * W3:(2 * l + j + 1) * W3].t())
            P[f"{lay}/{nm}/Dense_0/bias"] = sl(lambda l: E["ada/b"][(2 * l + j) * W3:(2 * l + j + 1) * W3])
        P["PaliGemma/llm/final_norm_1/Dense_0/kernel"] = E["ada/w"][2 * L * W3:].t().contiguous()
        P["PaliGemma/llm/final_norm_1/Dense_0/bias"] = E["ada/b"][2 * L * W3:]
        P["time_mlp_in/kernel"], P["time_mlp_in/bias"] = E["act/time_in_w"].t().contiguous(), E["act/time_in_b"]
        P["time_mlp_out/kernel"], P["time_mlp_out/bias"] = E["act/time_out_w"].t().contiguous(), E["act/time_out_b"]
    else:       # pi0 (lap.py:51,56-61)
        P[f"{lay}/pre_attention_norm_1/scale"] = sl(lambda l: E[f"llm/{l}/n_attn1"])
        P[f"{lay}/pre_ffw_norm_1/scale"] = sl(lambda l: E[f"llm/{l}/n_ffw1"])
        P["PaliGemma/llm/final_norm_1/scale"] = E["llm/final_norm1"]
        P["state_proj/kernel"], P["state_proj/bias"] = E["act/state_w"].t().contiguous(), E["act/state_b"]
        P["action_time_mlp_in/kernel"], P["action_time_mlp_in/bias"] = E["act/atime_in_w"].t().contiguous(), E["act/atime_in_b"]
        P["action_time_mlp_out/kernel"], P["action_time_mlp_out/bias"] = E["act/atime_out_w"].t().contiguous(), E["act/atime_out_b"]
    P["action_out_proj/kernel"], P["action_out_proj/bias"] = E["act/out_w"].t().contiguous(), E["act/out_b"]
    return {k: v.contiguous() for k, v in P.items()}
