"""CPU oracle for the LAP hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (lap_amd/) never does; it fails loudly if the HIP library is missing.

PARITY UNPINNED.  The reference (lihzha/lap) is a JAX/Flax program that cannot be imported here
(no jax/flax wheels, Python 3.10, and third_party/openpi — pinned nowhere in the mount — is empty;
SURVEY.md §0 F3-F5), and it ships no tests or golden vectors (F2).  This file is therefore a
line-by-line restatement in PyTorch (CPU, float32) of the cited reference lines; items restated
from memory of the missing `openpi` submodule are marked [UPSTREAM-RECALL] and isolated in one
function each.  It is cross-checked against independent implementations that DO run here
(HuggingFace transformers' Gemma decoder layer and SigLIP vision encoder with shared random
weights, tests/test_oracle_hf.py) and against analytic invariants.

Two numeric modes:
  * emulate_bf16=False: every op in float32 — the mathematical function of the reference.
  * emulate_bf16=True : values are rounded to bfloat16 wherever the reference's dtype flow
    (SURVEY.md §8 a-bis) produces a bfloat16 tensor, op by op as the Flax code is written.

Parameters use the reference's own tree paths and array layouts (lap.py:35-91 naming via
gemma.py:567-574), flattened with '/', so the key map doubles as the checkpoint interchange map.
"""
from __future__ import annotations

import dataclasses
import math

import torch

PALIGEMMA_VOCAB_SIZE = 257_152  # gemma.py:40
BIG_NEG = -2.3819763e38  # gemma.py:258


# --------------------------------------------------------------------------- configs
@dataclasses.dataclass(frozen=True)
class GemmaCfg:  # gemma.py:43-109
    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    num_kv_heads: int
    head_dim: int


GEMMA = {
    "dummy": GemmaCfg(64, 4, 128, 8, 1, 16),
    "gemma_300m": GemmaCfg(1024, 18, 4096, 8, 1, 256),
    "gemma_2b": GemmaCfg(2048, 18, 16384, 8, 1, 256),
}


@dataclasses.dataclass(frozen=True)
class SiglipCfg:  # siglip_gemma3.py:589-664 (decode_variant); patch 14
    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    patch: int = 14


SIGLIP = {
    "So400m/14": SiglipCfg(1152, 27, 4304, 16),
    "mu/14": SiglipCfg(32, 1, 128, 2),
}


@dataclasses.dataclass(frozen=True)
class OracleCfg:
    paligemma_variant: str = "gemma_2b"
    action_expert_variant: str = "gemma_300m"
    siglip_variant: str = "So400m/14"
    action_dim: int = 7
    action_horizon: int = 16
    max_token_len: int = 180
    image_size: int = 224
    image_keys: tuple = ("base_0_rgb", "left_wrist_0_rgb")
    vocab_size: int = PALIGEMMA_VOCAB_SIZE
    language_loss_weight: float = 1.0
    action_loss_weight: float = 1.0
    stop_action_to_vlm_grad: bool = False
    pi05: bool = True                          # lap_config.py:36; False = pi0 suffix: state token + action/time MLP, plain RMSNorm in the expert (lap.py:46-61)
    enable_action_training: bool = True        # lap_config.py:40-47: flow-matching loss + action-expert stream (lap.py:426-462,557-569)
    enable_langact_training: bool = True       # language-action cross entropy (lap.py:462-556)
    enable_vqa_training: bool = False          # lap_config.py:40-47 / lap.py:101-115
    enable_prediction_training: bool = False
    vqa_loss_weight: float = 0.1
    prediction_loss_weight: float = 1.0
    vqa_loss_weights_by_id: tuple = ()         # ((dataset id, weight), ...): lap.py:107-115 after the registry lookup
    emulate_bf16: bool = False
    # BASELINE.json config 5 (not a reference mode: lap_config.py:24 knows bfloat16 only): the VLM expert's projections
    # multiply e4m3-quantised operands (per-tensor scale 448 / amax, f32 accumulation), everything else as emulate_bf16
    emulate_fp8: bool = False

    @property
    def vlm(self) -> GemmaCfg:
        return GEMMA[self.paligemma_variant]

    @property
    def expert(self) -> GemmaCfg:
        return GEMMA[self.action_expert_variant]

    @property
    def img(self) -> SiglipCfg:
        return SIGLIP[self.siglip_variant]

    @property
    def n_img_tokens(self) -> int:
        return (self.image_size // self.img.patch) ** 2


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _mk_round(cfg: OracleCfg):
    if cfg.emulate_bf16 or cfg.emulate_fp8:
        return _RoundSTE.apply
    return lambda x: x


class _RoundWeightSTE(torch.autograd.Function):
    """`w.astype(bf16)` of an f32 master weight that feeds a bf16 dot (gemma.py:307,318; lora.Einsum; Flax Dense with dtype=bf16): the
    forward rounds the weight, and the cotangent that comes back through the cast is the bf16 OUTPUT of the weight-gradient dot — it is
    rounded to bf16 once before it reaches the f32 master.  (`_RoundSTE` passes the f32 cotangent through: right for activations, whose
    rounding points the backward of this restatement does not model, and one rounding short for weights.)"""
    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _mk_round_w(cfg: OracleCfg):
    if cfg.emulate_bf16 or cfg.emulate_fp8:
        return _RoundWeightSTE.apply
    return lambda x: x


class _Fp8STE(torch.autograd.Function):
    """Quantise-dequantise to OCP e4m3 with the per-tensor scale 448 / amax (straight-through in the backward)."""
    @staticmethod
    def forward(ctx, x):
        s = 448.0 / x.detach().abs().max().clamp_min(1e-12)
        return (x * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) / s

    @staticmethod
    def backward(ctx, g):
        return g


def _mk_q8(cfg: OracleCfg, expert: int):
    """Operand quantiser of expert `expert`'s projections: fp8 for the VLM (expert 0) in emulate_fp8 mode, identity otherwise."""
    if cfg.emulate_fp8 and expert == 0:
        return _Fp8STE.apply
    return lambda x: x


# --------------------------------------------------------------------------- init
def init_params(cfg: OracleCfg, seed: int = 0, zero_init_like_reference: bool = False) -> dict[str, torch.Tensor]:
    """Random parameters with the reference's names / shapes.

    Initialisers follow gemma.py:121,128,143-146,183-198,305-317 and siglip_gemma3.py:49-54,66-69
    in distribution family (lecun/xavier scale); norm scales and adaRMS Dense are zero in the
    reference — with zero_init_like_reference=False they get small random values instead so that
    parity tests exercise those paths.
    """
    g = torch.Generator().manual_seed(seed)
    P: dict[str, torch.Tensor] = {}

    def normal(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def small(*shape):
        return torch.zeros(*shape) if zero_init_like_reference else normal(*shape, std=0.05)

    v, e = cfg.vlm, cfg.expert
    L = v.depth
    P["PaliGemma/llm/embedder/input_embedding"] = normal(cfg.vocab_size, v.width, std=0.02 if cfg.vocab_size > 10000 else 0.3)
    for i, c in enumerate((v, e)):
        sfx = "" if i == 0 else f"_{i}"
        P[f"PaliGemma/llm/layers/attn/q_einsum{sfx}/w"] = normal(L, c.num_heads, c.width, c.head_dim, std=c.width ** -0.5)
        P[f"PaliGemma/llm/layers/attn/kv_einsum{sfx}/w"] = normal(L, 2, c.num_kv_heads, c.width, c.head_dim, std=c.width ** -0.5)
        P[f"PaliGemma/llm/layers/attn/attn_vec_einsum{sfx}/w"] = normal(L, c.num_heads, c.head_dim, c.width, std=(c.num_heads * c.head_dim) ** -0.5)
        P[f"PaliGemma/llm/layers/mlp{sfx}/gating_einsum"] = normal(L, 2, c.width, c.mlp_dim, std=c.width ** -0.5)
        P[f"PaliGemma/llm/layers/mlp{sfx}/linear"] = normal(L, c.mlp_dim, c.width, std=c.mlp_dim ** -0.5)
        if i == 0:
            P["PaliGemma/llm/layers/pre_attention_norm/scale"] = small(L, c.width)
            P["PaliGemma/llm/layers/pre_ffw_norm/scale"] = small(L, c.width)
            P["PaliGemma/llm/final_norm/scale"] = small(c.width)
        elif not cfg.pi05:  # pi0: `use_adarms=[False, False]` (lap.py:51) — the expert's norms are plain RMSNorms with a scale (gemma.py:121)
            P[f"PaliGemma/llm/layers/pre_attention_norm{sfx}/scale"] = small(L, c.width)
            P[f"PaliGemma/llm/layers/pre_ffw_norm{sfx}/scale"] = small(L, c.width)
            P[f"PaliGemma/llm/final_norm{sfx}/scale"] = small(c.width)
        else:  # adaRMS (pi05): Dense(width -> 3*width), zero-init kernel, zero bias (gemma.py:128)
            for nm in ("pre_attention_norm", "pre_ffw_norm"):
                P[f"PaliGemma/llm/layers/{nm}{sfx}/Dense_0/kernel"] = small(L, c.width, 3 * c.width) * (0.2 if not zero_init_like_reference else 1)
                P[f"PaliGemma/llm/layers/{nm}{sfx}/Dense_0/bias"] = small(L, 3 * c.width)
            P[f"PaliGemma/llm/final_norm{sfx}/Dense_0/kernel"] = small(c.width, 3 * c.width) * (0.2 if not zero_init_like_reference else 1)
            P[f"PaliGemma/llm/final_norm{sfx}/Dense_0/bias"] = small(3 * c.width)
    s = cfg.img
    hd = s.width // s.num_heads
    T = cfg.n_img_tokens
    P["PaliGemma/img/embedding/kernel"] = normal(s.patch, s.patch, 3, s.width, std=(s.patch * s.patch * 3) ** -0.5)
    P["PaliGemma/img/embedding/bias"] = normal(s.width, std=0.02)
    P["PaliGemma/img/pos_embedding"] = normal(1, T, s.width, std=s.width ** -0.5)
    blk = "PaliGemma/img/Transformer/encoderblock"
    for ln in ("LayerNorm_0", "LayerNorm_1"):
        P[f"{blk}/{ln}/scale"] = 1.0 + normal(s.depth, s.width, std=0.05)
        P[f"{blk}/{ln}/bias"] = normal(s.depth, s.width, std=0.05)
    for nm in ("query", "key", "value"):
        P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/kernel"] = normal(s.depth, s.width, s.num_heads, hd, std=s.width ** -0.5)
        P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/bias"] = normal(s.depth, s.num_heads, hd, std=0.02)
    P[f"{blk}/MultiHeadDotProductAttention_0/out/kernel"] = normal(s.depth, s.num_heads, hd, s.width, std=s.width ** -0.5)
    P[f"{blk}/MultiHeadDotProductAttention_0/out/bias"] = normal(s.depth, s.width, std=0.02)
    P[f"{blk}/MlpBlock_0/Dense_0/kernel"] = normal(s.depth, s.width, s.mlp_dim, std=s.width ** -0.5)
    P[f"{blk}/MlpBlock_0/Dense_0/bias"] = normal(s.depth, s.mlp_dim, std=0.02)
    P[f"{blk}/MlpBlock_0/Dense_1/kernel"] = normal(s.depth, s.mlp_dim, s.width, std=s.mlp_dim ** -0.5)
    P[f"{blk}/MlpBlock_0/Dense_1/bias"] = normal(s.depth, s.width, std=0.02)
    P["PaliGemma/img/Transformer/encoder_norm/scale"] = 1.0 + normal(s.width, std=0.05)
    P["PaliGemma/img/Transformer/encoder_norm/bias"] = normal(s.width, std=0.05)
    P["PaliGemma/img/head/kernel"] = normal(s.width, v.width, std=s.width ** -0.5)
    P["PaliGemma/img/head/bias"] = normal(v.width, std=0.02)
    # action head: nnx.Linear (lap.py:52-62), f32
    ad, w = cfg.action_dim, e.width
    P["action_in_proj/kernel"] = normal(ad, w, std=ad ** -0.5)
    P["action_in_proj/bias"] = normal(w, std=0.02)
    if cfg.pi05:
        for nm in ("time_mlp_in", "time_mlp_out"):
            P[f"{nm}/kernel"] = normal(w, w, std=w ** -0.5)
            P[f"{nm}/bias"] = normal(w, std=0.02)
    else:       # lap.py:56-61
        P["state_proj/kernel"] = normal(ad, w, std=ad ** -0.5)
        P["state_proj/bias"] = normal(w, std=0.02)
        P["action_time_mlp_in/kernel"] = normal(2 * w, w, std=(2 * w) ** -0.5)
        P["action_time_mlp_in/bias"] = normal(w, std=0.02)
        P["action_time_mlp_out/kernel"] = normal(w, w, std=w ** -0.5)
        P["action_time_mlp_out/bias"] = normal(w, std=0.02)
    P["action_out_proj/kernel"] = normal(w, ad, std=w ** -0.5)
    P["action_out_proj/bias"] = normal(ad, std=0.02)
    return P


# --------------------------------------------------------------------------- small pieces
def gelu_tanh(x):  # flax nn.gelu default approximate=True
    return torch.nn.functional.gelu(x, approximate="tanh")


def make_attn_mask(input_mask: torch.Tensor, mask_ar: torch.Tensor) -> torch.Tensor:
    """[UPSTREAM-RECALL] openpi pi0.make_attn_mask: tokens attend to valid tokens whose cumulative
    mask_ar is <= theirs.  input_mask, mask_ar: bool [B,T] -> bool [B,T,T]."""
    cs = torch.cumsum(mask_ar.to(torch.int64), dim=1)
    attn = cs[:, None, :] <= cs[:, :, None]
    valid = input_mask[:, None, :] & input_mask[:, :, None]
    return attn & valid


def posemb_sincos(t: torch.Tensor, dim: int, min_period: float, max_period: float) -> torch.Tensor:
    """[UPSTREAM-RECALL] openpi pi0.posemb_sincos: f32 [B] -> [B, dim] = concat[sin, cos]."""
    frac = torch.linspace(0.0, 1.0, dim // 2, dtype=torch.float32)
    period = min_period * (max_period / min_period) ** frac
    ang = t.to(torch.float32)[:, None] * (1.0 / period * 2 * math.pi)[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)


def apply_rope(x: torch.Tensor, positions: torch.Tensor, max_wavelength: float = 10_000.0) -> torch.Tensor:
    """gemma.py:548-564.  x [B,L,H,D] (f32 here), positions int [B,L]; result f32 (caller rounds)."""
    d = x.shape[-1]
    fe = (2.0 / d) * torch.arange(d // 2, dtype=torch.float32)
    timescale = max_wavelength ** fe
    rad = positions[..., None].to(torch.float32) / timescale[None, None, :]
    rad = rad[..., None, :]
    sin, cos = torch.sin(rad), torch.cos(rad)
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)


def rmsnorm(x, scale=None, cond=None, dense_k=None, dense_b=None, r=lambda t: t, rw=None):
    """gemma.py:113-131.  Returns (normed, gate or None)."""
    var = torch.mean(torch.square(x), dim=-1, keepdim=True)
    normed = x * torch.reciprocal(torch.sqrt(var + 1e-6))
    if cond is None:
        return r(normed * (1 + scale)), None
    modulation = r(r(r(cond) @ (rw or r)(dense_k)) + r(dense_b))  # nn.Dense(dtype=bf16)
    sc, sh, gate = torch.chunk(modulation[:, None, :], 3, dim=-1)
    return r(normed * r(1 + sc) + sh), gate


# --------------------------------------------------------------------------- SigLIP
def siglip_forward(P, cfg: OracleCfg, image: torch.Tensor, collect: dict | None = None) -> torch.Tensor:
    """openpi.models.siglip._Module.__call__ (missing) restated from the in-tree Gemma3 variant
    siglip_gemma3.py:382-545 minus its soft-embedding RMSNorm (:432) and with the PaliGemma head bias.
    image f32 [B,H,W,3] in [-1,1] -> [B, 256, vlm.width]."""
    r, rw = _mk_round(cfg), _mk_round_w(cfg)
    s = cfg.img
    B, H, W, C = image.shape
    p = s.patch
    gh, gw = H // p, W // p
    # conv 14x14/14 VALID == im2col matmul, f32 (:401-408)
    patches = image.reshape(B, gh, p, gw, p, C).permute(0, 1, 3, 2, 4, 5).reshape(B, gh * gw, p * p * C)
    x = patches @ P["PaliGemma/img/embedding/kernel"].reshape(p * p * C, s.width) + P["PaliGemma/img/embedding/bias"]
    x = x + P["PaliGemma/img/pos_embedding"]  # f32 (:418)
    x = r(x)  # cast to dtype_mm (:436)
    if collect is not None:
        collect["img/stem"] = x
    hd = s.width // s.num_heads
    blk = "PaliGemma/img/Transformer/encoderblock"

    def ln(v, sc, bi):
        return r(torch.nn.functional.layer_norm(v, (s.width,), sc, bi, eps=1e-6))

    for l in range(s.depth):
        mha = f"{blk}/MultiHeadDotProductAttention_0"
        y = ln(x, P[f"{blk}/LayerNorm_0/scale"][l], P[f"{blk}/LayerNorm_0/bias"][l])
        q = r(r(torch.einsum("btd,dnh->btnh", y, rw(P[f"{mha}/query/kernel"][l]))) + r(P[f"{mha}/query/bias"][l]))
        k = r(r(torch.einsum("btd,dnh->btnh", y, rw(P[f"{mha}/key/kernel"][l]))) + r(P[f"{mha}/key/bias"][l]))
        v = r(r(torch.einsum("btd,dnh->btnh", y, rw(P[f"{mha}/value/kernel"][l]))) + r(P[f"{mha}/value/bias"][l]))
        q = r(q / r(torch.tensor(math.sqrt(hd))))  # flax dot_product_attention_weights
        logits = r(torch.einsum("bqnh,bknh->bnqk", q, k))
        probs = r(torch.softmax(logits, dim=-1))
        enc = r(torch.einsum("bnqk,bknh->bqnh", probs, v))
        y = r(r(torch.einsum("bqnh,nhd->bqd", enc, rw(P[f"{mha}/out/kernel"][l]))) + r(P[f"{mha}/out/bias"][l]))
        x = r(x + y)
        y = ln(x, P[f"{blk}/LayerNorm_1/scale"][l], P[f"{blk}/LayerNorm_1/bias"][l])
        h = r(r(y @ rw(P[f"{blk}/MlpBlock_0/Dense_0/kernel"][l])) + r(P[f"{blk}/MlpBlock_0/Dense_0/bias"][l]))
        h = r(gelu_tanh(h))
        y = r(r(h @ rw(P[f"{blk}/MlpBlock_0/Dense_1/kernel"][l])) + r(P[f"{blk}/MlpBlock_0/Dense_1/bias"][l]))
        x = r(x + y)
        if collect is not None:
            collect[f"img/block{l:02d}"] = x
    x = ln(x, P["PaliGemma/img/Transformer/encoder_norm/scale"], P["PaliGemma/img/Transformer/encoder_norm/bias"])
    if collect is not None:
        collect["img/encoded"] = x
    x = r(r(x @ rw(P["PaliGemma/img/head/kernel"])) + r(P["PaliGemma/img/head/bias"]))
    if collect is not None:
        collect["img/out"] = x
    return x


# --------------------------------------------------------------------------- Gemma (multi-expert)
def gemma_forward(P, cfg: OracleCfg, embedded, positions, mask, adarms_cond=None, kv_cache=None, collect=None):
    """gemma.Module.__call__ (gemma.py:455-531) + Block (336-387) + Attention (167-290).
    embedded: [x0 or None, x1 or None]; positions int [B,T]; mask bool [B,T,S];
    kv_cache: list of (K [B,S0,1,H], V) per layer or None.  Returns (outs, new_cache)."""
    r, rw = _mk_round(cfg), _mk_round_w(cfg)
    cfgs = (cfg.vlm, cfg.expert)
    if adarms_cond is None:
        adarms_cond = [None, None]
    xs = [r(e) if e is not None else None for e in embedded]  # astype(embed_dtype) (:494)
    if collect is not None:
        for i, x in enumerate(xs):
            if x is not None:
                collect[f"llm/in{i}"] = x
    L = cfg.vlm.depth
    new_cache = []
    nkv = cfg.vlm.num_kv_heads
    G = cfg.vlm.num_heads // nkv
    lay = "PaliGemma/llm/layers"
    for l in range(L):
        pre, gates = [], []
        for i, x in enumerate(xs):
            if x is None:
                pre.append(None); gates.append(None); continue
            sfx = "" if i == 0 else f"_{i}"
            if adarms_cond[i] is None:
                y, gate = rmsnorm(x, scale=P[f"{lay}/pre_attention_norm{sfx}/scale"][l], r=r)
            else:
                y, gate = rmsnorm(x, cond=adarms_cond[i], dense_k=P[f"{lay}/pre_attention_norm{sfx}/Dense_0/kernel"][l],
                                  dense_b=P[f"{lay}/pre_attention_norm{sfx}/Dense_0/bias"][l], r=r, rw=rw)
            pre.append(y); gates.append(gate)
        qs, ks, vs = [], [], []
        for i, x in enumerate(pre):
            if x is None:
                continue
            sfx = "" if i == 0 else f"_{i}"
            q8 = _mk_q8(cfg, i)
            wq, wkv = rw(P[f"{lay}/attn/q_einsum{sfx}/w"][l]), rw(P[f"{lay}/attn/kv_einsum{sfx}/w"][l])
            if cfg.emulate_fp8 and i == 0:   # the engine quantises the packed q|k|v weight with ONE scale
                flat = q8(torch.cat([wq.reshape(-1), wkv.reshape(-1)]))
                wq, wkv = flat[:wq.numel()].view_as(wq), flat[wq.numel():].view_as(wkv)
            xq = q8(x)
            qs.append(r(torch.einsum("btd,ndh->btnh", xq, wq)))
            kv = r(torch.einsum("bsd,xkdh->xbskh", xq, wkv))
            ks.append(kv[0]); vs.append(kv[1])
        q = torch.cat(qs, dim=1); k = torch.cat(ks, dim=1); v = torch.cat(vs, dim=1)
        q = r(apply_rope(q, positions))
        q = r(q * r(torch.tensor(cfg.vlm.head_dim ** -0.5)))  # bf16 multiply (:216)
        k = r(apply_rope(k, positions))
        if kv_cache is not None:  # suffix-only decode: concat [cache | new] (:228-230)
            ck, cv = kv_cache[l]
            k = torch.cat([ck, k], dim=1); v = torch.cat([cv, v], dim=1)
        new_cache.append((k, v))
        B, T = q.shape[:2]
        qg = q.reshape(B, T, nkv, G, -1)
        if cfg.stop_action_to_vlm_grad and xs[0] is not None and xs[1] is not None:
            n0 = xs[0].shape[1]
            k_c = torch.cat([k[:, :n0].detach(), k[:, n0:]], dim=1)
            v_c = torch.cat([v[:, :n0].detach(), v[:, n0:]], dim=1)
            logits = torch.cat([torch.einsum("btkgh,bskh->bkgts", qg[:, :n0], k),
                                torch.einsum("btkgh,bskh->bkgts", qg[:, n0:], k_c)], dim=3)
        else:
            logits = torch.einsum("btkgh,bskh->bkgts", qg, k)  # f32 (:235)
            v_c = None
        if mask.shape != (B, T, k.shape[1]):
            raise ValueError(f"Attention mask with shape {mask.shape} but shapes for q and k are: {q.shape} and {k.shape}")
        masked = torch.where(mask[:, None, None, :, :], logits, torch.tensor(BIG_NEG))
        probs = r(torch.softmax(masked, dim=-1))
        if v_c is not None:
            n0 = xs[0].shape[1]
            enc = torch.cat([torch.einsum("bkgts,bskh->btkgh", probs[:, :, :, :n0], v),
                             torch.einsum("bkgts,bskh->btkgh", probs[:, :, :, n0:], v_c)], dim=1)
        else:
            enc = torch.einsum("bkgts,bskh->btkgh", probs, v)
        enc = r(enc).reshape(B, T, nkv * G, -1)
        outs, start = [], 0
        for i, x in enumerate(xs):
            if x is None:
                outs.append(None); continue
            sfx = "" if i == 0 else f"_{i}"
            end = start + x.shape[1]
            q8 = _mk_q8(cfg, i)
            outs.append(r(torch.einsum("btnh,nhd->btd", q8(enc[:, start:end]), q8(rw(P[f"{lay}/attn/attn_vec_einsum{sfx}/w"][l])))))
            start = end
        xs = [_gated_residual(x, y, g, r) for x, y, g in zip(xs, outs, gates)]
        outs, gates = [], []
        for i, x in enumerate(xs):
            if x is None:
                outs.append(None); gates.append(None); continue
            sfx = "" if i == 0 else f"_{i}"
            if adarms_cond[i] is None:
                y, gate = rmsnorm(x, scale=P[f"{lay}/pre_ffw_norm{sfx}/scale"][l], r=r)
            else:
                y, gate = rmsnorm(x, cond=adarms_cond[i], dense_k=P[f"{lay}/pre_ffw_norm{sfx}/Dense_0/kernel"][l],
                                  dense_b=P[f"{lay}/pre_ffw_norm{sfx}/Dense_0/bias"][l], r=r, rw=rw)
            q8 = _mk_q8(cfg, i)
            wg = q8(rw(P[f"{lay}/mlp{sfx}/gating_einsum"][l]))      # gate | up share one scale (packed weight)
            yq = q8(y)
            ff_gate = r(yq @ wg[0])
            ff1 = r(yq @ wg[1])
            act = r(r(gelu_tanh(ff_gate)) * ff1)
            outs.append(r(q8(act) @ q8(rw(P[f"{lay}/mlp{sfx}/linear"][l]))))
            gates.append(gate)
        xs = [_gated_residual(x, y, g, r) for x, y, g in zip(xs, outs, gates)]
        if collect is not None:
            for i, x in enumerate(xs):
                if x is not None:
                    collect[f"llm/layer{l:02d}/x{i}"] = x
    final = []
    for i, x in enumerate(xs):
        if x is None:
            final.append(None); continue
        sfx = "" if i == 0 else f"_{i}"
        if adarms_cond[i] is None:
            final.append(rmsnorm(x, scale=P[f"PaliGemma/llm/final_norm{sfx}/scale"], r=r)[0])
        else:
            final.append(rmsnorm(x, cond=adarms_cond[i], dense_k=P[f"PaliGemma/llm/final_norm{sfx}/Dense_0/kernel"],
                                 dense_b=P[f"PaliGemma/llm/final_norm{sfx}/Dense_0/bias"], r=r, rw=rw)[0])
    return final, new_cache


def _gated_residual(x, y, gate, r):  # gemma.py:577-583
    if x is None:
        return None
    if gate is None:
        return r(x + y)
    return r(x + r(y * gate))


# --------------------------------------------------------------------------- LAP
def embed_prefix(P, cfg, obs, collect=None):
    """lap.py:118-170.  obs: dict(images{key:[B,H,W,3]}, image_masks{key:[B]}, tokenized_prompt [B,L] int,
    tokenized_prompt_mask [B,L] bool, tokenized_langact_mask [B,L] bool or None)."""
    r = _mk_round(cfg)
    toks, imask, armask = [], [], []
    for name in cfg.image_keys:
        it = siglip_forward(P, cfg, obs["images"][name], collect if name == cfg.image_keys[0] else None)
        toks.append(it)
        B, S = it.shape[:2]
        imask.append(obs["image_masks"][name][:, None].expand(B, S))
        armask.append(torch.zeros(B, S, dtype=torch.bool))
    table = P["PaliGemma/llm/embedder/input_embedding"]
    emb = table[obs["tokenized_prompt"].long()] * math.sqrt(cfg.vlm.width)  # gemma.py:148-151
    toks.append(r(emb))
    imask.append(obs["tokenized_prompt_mask"])
    la = obs.get("tokenized_langact_mask")
    armask.append(la if la is not None else torch.zeros_like(obs["tokenized_prompt_mask"]))
    return torch.cat(toks, 1), torch.cat(imask, 1), torch.cat(armask, 1)


def embed_suffix(P, cfg, noisy_actions, timestep, state=None):
    """[UPSTREAM-RECALL] openpi Pi0.embed_suffix.  f32.  pi05: action tokens + the time MLP's output as the adaRMS condition.
    pi0 (`pi05=False`, parameters of lap.py:56-61): a state token `state_proj(obs.state)` in front (an autoregressive block of
    its own: prefix tokens do not see it, action tokens do), the action tokens mixed with the time embedding through
    `action_time_mlp_in([action | time]) -> swish -> action_time_mlp_out`, no adaRMS condition."""
    action_tokens = noisy_actions @ P["action_in_proj/kernel"] + P["action_in_proj/bias"]
    time_emb = posemb_sincos(timestep, cfg.expert.width, 4e-3, 4.0)
    if not cfg.pi05:
        B, S = action_tokens.shape[:2]
        state_token = (state.to(torch.float32) @ P["state_proj/kernel"] + P["state_proj/bias"])[:, None, :]
        at = torch.cat([action_tokens, time_emb[:, None, :].expand(B, S, -1)], dim=-1)
        at = torch.nn.functional.silu(at @ P["action_time_mlp_in/kernel"] + P["action_time_mlp_in/bias"])
        at = at @ P["action_time_mlp_out/kernel"] + P["action_time_mlp_out/bias"]
        tokens = torch.cat([state_token, at], dim=1)
        ar = torch.zeros(S + 1, dtype=torch.bool)
        ar[0] = ar[1] = True
        return tokens, torch.ones(B, S + 1, dtype=torch.bool), ar, None
    time_emb = torch.nn.functional.silu(time_emb @ P["time_mlp_in/kernel"] + P["time_mlp_in/bias"])
    time_emb = torch.nn.functional.silu(time_emb @ P["time_mlp_out/kernel"] + P["time_mlp_out/bias"])
    B, S = action_tokens.shape[:2]
    suffix_mask = torch.ones(B, S, dtype=torch.bool)
    ar = torch.zeros(S, dtype=torch.bool)
    ar[0] = True
    return action_tokens, suffix_mask, ar, time_emb


def build_masks_positions(cfg, obs, prefix_mask, prefix_ar, suffix_mask, suffix_ar):
    """lap.py:303-377: prefix_mask_action, combined mask, combined positions."""
    la = obs.get("tokenized_langact_mask")
    if la is None:
        pma = prefix_mask
    else:
        n_img = prefix_mask.shape[1] - la.shape[1]
        full = torch.cat([torch.zeros(la.shape[0], n_img, dtype=torch.bool), la], 1)
        pma = prefix_mask & ~full
    prefix_attn = make_attn_mask(prefix_mask, prefix_ar)
    B, Pn = prefix_mask.shape
    S = suffix_mask.shape[1]
    combined = torch.zeros(B, Pn + S, Pn + S, dtype=torch.bool)
    combined[:, :Pn, :Pn] = prefix_attn
    inp = torch.cat([pma, suffix_mask], 1)
    ar = torch.cat([torch.zeros_like(pma), suffix_ar], 1)
    action_mask = make_attn_mask(inp, ar)
    combined[:, Pn:, :] = action_mask[:, Pn:, :]
    ppos = torch.cumsum(prefix_mask.long(), 1) - 1
    spos = pma.long().sum(-1, keepdim=True) + torch.cumsum(suffix_mask.long(), -1) - 1
    return pma, combined, torch.cat([ppos, spos], 1)


def compute_loss(P, cfg: OracleCfg, obs, actions, noise, time, collect=None):
    """lap.py:380-602 (image augmentation off), with the random draws (noise ~ N(0,1), time ~ Beta(1.5,1)*.999+.001,
    lap.py:193-194) supplied by the caller.  All three branches of the loss assembly:
      * action + langact training (LAP-3B): both streams, cross entropy + flow matching;
      * enable_action_training=False (VLA-0 style, lap.py:426-462): `llm([prefix])` with the prefix's own mask and positions,
        cross entropy only, final loss = sum / active samples (lap.py:590-596);
      * enable_langact_training=False (pi0 style): both streams, flow matching only (lang_term = 0, lap.py:579-589).
    Returns (loss, metrics)."""
    r = _mk_round(cfg)
    B = actions.shape[0]
    act_on, lang_on = cfg.enable_action_training, cfg.enable_langact_training
    prefix_tokens, prefix_mask, prefix_ar = embed_prefix(P, cfg, obs, collect)
    if act_on:
        te = time[:, None, None]
        x_t = te * noise + (1 - te) * actions
        u_t = noise - actions
        suffix_tokens, suffix_mask, suffix_ar1, cond = embed_suffix(P, cfg, x_t, time, state=obs.get("state"))
        suffix_ar = suffix_ar1[None].expand(B, -1)
        _, mask, positions = build_masks_positions(cfg, obs, prefix_mask, prefix_ar, suffix_mask, suffix_ar)
        (pre0, pre1), _ = gemma_forward(P, cfg, [prefix_tokens, suffix_tokens], positions, mask, [None, cond], collect=collect)
    else:       # lap.py:431-455: prefix_mask_action = prefix_mask, mask = make_attn_mask(prefix), positions = cumsum - 1
        mask = make_attn_mask(prefix_mask, prefix_ar)
        positions = torch.cumsum(prefix_mask.long(), 1) - 1
        (pre0, pre1), _ = gemma_forward(P, cfg, [prefix_tokens, None], positions, mask, [None, None], collect=collect)
    if collect is not None:
        collect["llm/out0"] = pre0
        collect["llm/out1"] = pre1
        collect["mask"] = mask
        collect["positions"] = positions
    sm = obs.get("sample_mask")
    smb = sm if sm is not None else torch.ones(B, dtype=torch.bool)
    vqa = obs.get("is_vqa_sample") if cfg.enable_vqa_training else None
    pred = obs.get("is_prediction_sample") if cfg.enable_prediction_training else None
    lang_loss = torch.zeros(B)
    lang_ps = torch.zeros(B)
    vqa_mask = pred_mask = None
    if lang_on:
        # language loss (lap.py:209-289)
        tok = obs["tokenized_prompt"].long()
        tgt = tok[:, 1:]
        pl = pre0[:, :-1][:, -tgt.shape[1]:]
        logits = pl @ P["PaliGemma/llm/embedder/input_embedding"].t()  # bf16 x f32 -> f32 (gemma.py:153-154)
        loss_mask = obs["tokenized_langact_mask"][:, 1:] & obs["tokenized_prompt_mask"][:, 1:] & obs["token_loss_mask"][:, 1:]
        lm = loss_mask.to(torch.float32)
        if sm is not None:
            lm = lm * sm[:, None].to(torch.float32)
        logp = torch.log_softmax(logits, dim=-1)
        token_pplx = logp.gather(-1, tgt[..., None]).squeeze(-1)
        lang_loss = -(token_pplx * lm).sum(-1) / torch.clamp(lm.sum(-1), min=1)
        # combination (lap.py:472-556)
        if cfg.enable_vqa_training or cfg.enable_prediction_training:
            vqa_raw = vqa if vqa is not None else torch.zeros(B, dtype=torch.bool)
            pred_raw = pred if pred is not None else torch.zeros(B, dtype=torch.bool)
            lang_mask = ~(vqa_raw | pred_raw) & smb
            vqa_mask, pred_mask = vqa_raw & smb, pred_raw & smb
            vqa_w = torch.full((B,), cfg.vqa_loss_weight)
            if cfg.enable_vqa_training and cfg.vqa_loss_weights_by_id and obs.get("vqa_dataset_id") is not None:
                for did, wgt in cfg.vqa_loss_weights_by_id:
                    vqa_w = torch.where(obs["vqa_dataset_id"] == did, torch.tensor(float(wgt)), vqa_w)
            lang_ps = vqa_w * lang_loss * vqa_mask + cfg.prediction_loss_weight * lang_loss * pred_mask + cfg.language_loss_weight * lang_loss * lang_mask
        else:
            lang_ps = cfg.language_loss_weight * lang_loss
    metrics = {"lang_loss": lang_loss.mean(), "per_sample_lang": lang_loss}
    if act_on:
        # action loss (lap.py:291-301) and its sample mask (lap.py:557-569: the masks as they stand at this point, i.e. AND-ed
        # with the sample mask only when the langact branch ran with VQA / prediction mixing)
        v_t = pre1[:, -cfg.action_horizon:] @ P["action_out_proj/kernel"] + P["action_out_proj/bias"]
        act_loss = torch.mean(torch.square(v_t - u_t), dim=(-1, -2))
        act_mask = torch.ones(B, dtype=torch.bool)
        for msk, raw in ((vqa_mask, vqa), (pred_mask, pred)):
            m_ = msk if msk is not None else raw
            if m_ is not None:
                act_mask = act_mask & ~m_
        amf = act_mask.to(torch.float32)
        action_term = (cfg.action_loss_weight * act_loss * amf).sum() / torch.clamp(amf.sum(), min=1.0)
        if lang_on:
            lang_term = lang_ps.sum() / torch.clamp(sm.to(torch.float32).sum(), min=1.0) if sm is not None else lang_ps.mean()
        else:
            lang_term = 0.0
        loss = lang_term + action_term
        metrics.update(action_loss=act_loss.mean(), per_sample_action=act_loss, v_t=v_t, u_t=u_t)
    elif lang_on and sm is not None:      # lap.py:590-593
        loss = lang_ps.sum() / torch.clamp(sm.to(torch.float32).sum(), min=1.0)
    else:
        loss = lang_ps.mean()
    return loss, metrics


def sample_actions(P, cfg: OracleCfg, obs, noise, num_steps: int = 10, collect=None):
    """lap.py:605-675: prefix prefill -> KV cache -> Euler integration from t=1 to 0."""
    dt = -1.0 / num_steps
    B = noise.shape[0]
    prefix_tokens, prefix_mask, prefix_ar = embed_prefix(P, cfg, obs)
    prefix_attn = make_attn_mask(prefix_mask, prefix_ar)
    positions = torch.cumsum(prefix_mask.long(), 1) - 1
    _, cache = gemma_forward(P, cfg, [prefix_tokens, None], positions, prefix_attn, [None, None])
    x_t, t = noise.clone(), 1.0
    step = 0
    while t >= -dt / 2:
        suffix_tokens, suffix_mask, suffix_ar1, cond = embed_suffix(P, cfg, x_t, torch.full((B,), t, dtype=torch.float32), state=obs.get("state"))
        suffix_attn = make_attn_mask(suffix_mask, suffix_ar1[None].expand(B, -1))
        pmask = prefix_mask[:, None, :].expand(B, suffix_tokens.shape[1], -1)
        full = torch.cat([pmask, suffix_attn], dim=-1)
        pos = prefix_mask.long().sum(-1)[:, None] + torch.cumsum(suffix_mask.long(), -1) - 1
        (_, out1), _ = gemma_forward(P, cfg, [None, suffix_tokens], pos, full, [None, cond], kv_cache=cache)
        v_t = out1[:, -cfg.action_horizon:] @ P["action_out_proj/kernel"] + P["action_out_proj/bias"]
        if collect is not None:
            collect[f"v_t/{step}"] = v_t
        x_t = x_t + dt * v_t
        t = t + dt
        step += 1
    return x_t


def left_to_right_align(x, input_mask, attn_mask):
    """[UPSTREAM-RECALL] openpi pi0_fast.left_to_right_align (vmapped per sample): roll every sequence so that the
    span up to its last valid token ends at the right edge; padding ends up on the left."""
    xs, ms, as_ = [], [], []
    T = input_mask.shape[1]
    for b in range(x.shape[0]):
        seqlen = int((input_mask[b].long() * torch.arange(T)).max().item()) + 1
        xs.append(torch.roll(x[b], -seqlen, 0))
        ms.append(torch.roll(input_mask[b], -seqlen, 0))
        as_.append(torch.roll(attn_mask[b], (-seqlen, -seqlen), (0, 1)))
    return torch.stack(xs), torch.stack(ms), torch.stack(as_)


EOS_TOKEN = 1  # lap.py: `self.EOS_TOKEN` (PaliGemma tokenizer <eos>)


def sample_tokens(P, cfg: OracleCfg, obs, max_decoding_steps: int = 390, temperature: float = 0.0, collect=None,
                  eos_token: int = EOS_TOKEN):
    """lap.py:678-766: right-aligned VLM prefill -> KV cache -> greedy single-token decode with expert 0 only until every
    sample has produced EOS or `max_decoding_steps` tokens.  temperature > 0 (categorical sampling with the JAX PRNG)
    is not restated: it cannot be reproduced bit for bit without that generator."""
    if temperature > 0.0:
        raise NotImplementedError("temperature sampling depends on the JAX PRNG stream")
    table = P["PaliGemma/llm/embedder/input_embedding"]
    r = _mk_round(cfg)
    prefix_tokens, prefix_mask, prefix_ar = embed_prefix(P, cfg, obs)
    prefix_attn = make_attn_mask(prefix_mask, prefix_ar)
    prefix_tokens, prefix_mask, prefix_attn = left_to_right_align(prefix_tokens, prefix_mask, prefix_attn)
    B, prefill_size = prefix_mask.shape
    prefill_len = prefix_mask.long().sum(-1)
    prefix_start = prefill_size - prefill_len
    positions = torch.cumsum(prefix_mask.long(), -1) - 1
    (pre, _), cache = gemma_forward(P, cfg, [prefix_tokens, None], positions, prefix_attn, [None, None])
    last_logit = pre[:, -1:] @ table.t()                      # Embedder.decode (gemma.py:153-154), f32
    out = torch.zeros(B, max_decoding_steps, dtype=torch.int32)
    eos = torch.zeros(B, dtype=torch.bool)
    step = 0
    while (not bool(eos.all())) and step < max_decoding_steps:
        token = torch.argmax(last_logit, dim=-1).to(torch.int32)          # [B, 1]
        if collect is not None:
            collect[f"logit/{step}"] = last_logit[:, 0]
        out[:, step] = token[:, 0]
        eos = eos | (token[:, 0] == eos_token)
        emb = r(table[token.long()] * math.sqrt(cfg.vlm.width))           # llm(token, method="embed")
        pos = prefill_len[:, None] + step
        ar = torch.arange(prefill_size + step + 1)                        # cached keys + the fresh one
        mask = (ar[None, None, :] >= prefix_start[:, None, None]) & (ar[None, None, :] < prefill_size + step + 1)
        (pre, _), cache = gemma_forward(P, cfg, [emb, None], pos, mask, [None, None], kv_cache=cache)
        last_logit = pre @ table.t()
        step += 1
    return out


# --------------------------------------------------------------------------- optimizer / schedules
def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.95, eps=1e-8, wd=1e-4, clip_scale=1.0):
    """optax.chain(clip_by_global_norm, adamw) for one tensor, step counted from 1
    ([UPSTREAM-RECALL] openpi optimizer.AdamW defaults b1=.9 b2=.95 eps=1e-8; wd from config.py:517-519)."""
    g = g * clip_scale
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    upd = (m / (1 - b1 ** step)) / (torch.sqrt(v / (1 - b2 ** step)) + eps) + wd * p
    return p - lr * upd, m, v


def clip_scale(global_norm: float, max_norm: float = 1.0) -> float:
    return 1.0 if global_norm < max_norm else max_norm / global_norm


def cosine_lr(step, warmup_steps, peak_lr, decay_steps, decay_lr):
    """[UPSTREAM-RECALL] openpi CosineDecaySchedule -> optax.warmup_cosine_decay_schedule(
    init_value=peak/(warmup+1), peak_value=peak, warmup_steps, decay_steps, end_value=decay_lr)."""
    init = peak_lr / (warmup_steps + 1)
    if step < warmup_steps:
        return init + (peak_lr - init) * step / warmup_steps
    frac = min(max((step - warmup_steps) / max(decay_steps - warmup_steps, 1), 0.0), 1.0)
    cos = 0.5 * (1 + math.cos(math.pi * frac))
    return decay_lr + (peak_lr - decay_lr) * cos


def ema_decay_for_step(kind, step, ema_decay, start_step, num_train_steps):
    """training/config.py:549-589 (get_ema_decay_for_step) + :372-504."""
    if ema_decay is None or kind == "disabled":
        return 0.0, False
    if kind == "cosine_delayed":
        dur = max(num_train_steps - start_step, 1)
        prog = min(max((step - start_step) / dur, 0.0), 1.0)
        return ema_decay * (1 - math.cos(math.pi * prog)) / 2, step >= start_step
    if kind == "constant":
        return ema_decay, True
    if kind == "delayed":
        if start_step <= 0 or step >= start_step:
            return ema_decay, True
        return 0.0, False
    raise ValueError(kind)


# ------------------------------------------------------------------------------ train-time image augmentation
def augment_images(img: torch.Tensor, par: torch.Tensor) -> torch.Tensor:
    """models/model_adapter.py:118-151 [UPSTREAM-RECALL of augmax]: Chain(RandomCrop 95 %, Resize, Rotate, ColorJitter) on
    f32 [B,H,W,3] images in [-1,1] with the random parameters given (layout of lap_augment_images: crop offset x, y, crop
    width, height, cos, sin, brightness, contrast, saturation, skip).  One composed coordinate map + bilinear sampling with
    zero fill, then brightness / contrast tone curve / HSV saturation per pixel."""
    B, H, W, _ = img.shape
    par = par.to(torch.float32)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    xc, yc = (xs + 0.5 - 0.5 * W)[None], (ys + 0.5 - 0.5 * H)[None]
    q = lambda i: par[:, i].view(B, 1, 1)
    xr, yr = q(4) * xc - q(5) * yc, q(5) * xc + q(4) * yc
    sx = q(0) + 0.5 * q(2) + xr * (q(2) / W) - 0.5
    sy = q(1) + 0.5 * q(3) + yr * (q(3) / H) - 0.5
    x0, y0 = torch.floor(sx), torch.floor(sy)
    ax, ay = (sx - x0)[..., None], (sy - y0)[..., None]
    unit = img.to(torch.float32) * 0.5 + 0.5
    out = torch.zeros_like(unit)
    bidx = torch.arange(B).view(B, 1, 1).expand(B, H, W)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = (x0 + dx).long(), (y0 + dy).long()
            ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H))[..., None]
            wgt = (ax if dx else 1 - ax) * (ay if dy else 1 - ay)
            out = out + torch.where(ok, wgt * unit[bidx, yy.clamp(0, H - 1), xx.clamp(0, W - 1)], torch.zeros(()))
    bri = lambda v, b: torch.where(b < 0, v * (1 + b), v * (1 - b) + b)
    b_, c_, s_ = (par[:, i].view(B, 1, 1, 1) for i in (6, 7, 8))
    v = bri(out, b_)
    slant = torch.tan((c_ + 1.0) * (math.pi / 4))
    safe = torch.where((slant - 1).abs() < 1e-6, torch.full_like(slant, 2.0), slant)
    p1 = (safe - safe * safe) / (2 * (1 - safe * safe))
    curve = torch.where(v < p1, v / safe, torch.where(v > 1 - p1, v / safe + 1 - 1 / safe, safe * (v - 0.5) + 0.5))
    v = torch.where((slant - 1).abs() < 1e-6, v, curve)
    mx, mn = v.max(-1, keepdim=True).values, v.min(-1, keepdim=True).values
    sat = torch.where(mx > 0, (mx - mn) / mx.clamp_min(1e-30), torch.zeros(()))
    k = torch.where(sat > 0, bri(sat, s_).clamp(0, 1) / sat.clamp_min(1e-30), torch.zeros(()))
    v = (mx - (mx - v) * k).clamp(0, 1) * 2 - 1
    skip = (par[:, 9] != 0).view(B, 1, 1, 1)
    return torch.where(skip, img.to(torch.float32), v)
