"""ctypes binding of liblap_hip.so — the C ABI declared in include/lap_hip.h.

This module is the only place where Python touches the kernels.  It takes torch tensors
(device memory + the current HIP stream are PyTorch's job: plumbing), checks dtype /
contiguity, and passes raw device pointers.  There is NO fallback: if the shared library
is missing or a launch is rejected, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import pathlib

import torch

import os as _os

# LAP_HIP_LIB_VARIANT=<v>: load lap_amd/liblap_hip_<v>.so (a probe build with the ablation bits of LAP_GEMM_EXPERIMENTAL compiled in,
# `python -m lap_amd.build --variant=<v>`) instead of the production library.  Timing probes only; a missing file fails like the default.
_LIB_PATH = pathlib.Path(__file__).resolve().parent / (
    f"liblap_hip_{_os.environ['LAP_HIP_LIB_VARIANT']}.so" if _os.environ.get("LAP_HIP_LIB_VARIANT") else "liblap_hip.so")

GEMM_OUT_F32 = 1
GEMM_ACCUM = 2
GEMM_GELU = 4
GEMM_BIAS_F32 = 8
GEMM_PARTIALS = 16
GEMM_GELU_BF16 = 32
GEMM_GELU_EXP2 = 128
GEMM_GEGLU = 64


class LapHipError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} not found: build it with `python -m lap_amd.build` (hipcc --offload-arch=gfx950). "
            "lap_amd has no CPU / eager fallback."
        )
    return C.CDLL(str(_LIB_PATH))


_lib = _load()

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class AttnFwdArgs(C.Structure):
    _fields_ = [
        ("q", _vp * 2), ("o", _vp * 2), ("k", _vp * 2), ("v", _vp * 2),
        ("q_len", _i * 2), ("k_len", _i * 2),
        ("q_rs", _i * 2), ("kv_rs", _i * 2), ("o_rs", _i * 2),
        ("qinfo", _vp), ("kinfo", _vp), ("lse", _vp),
        ("scratch", _vp), ("scratch_floats", _ll),
        ("scale", _f), ("nsplit", _i),
        ("B", _i), ("NH", _i), ("NKV", _i), ("HD", _i),
    ]


CHAIN_MAX_DEPTH = 32    # LAP_CHAIN_MAX_DEPTH (include/lap_hip.h)


class ServeChainArgs(C.Structure):
    _fields_ = [
        ("depth", _i), ("B", _i), ("S", _i), ("D", _i), ("H", _i), ("NH", _i), ("HD", _i), ("prefix_len", _i),
        ("x_in", _vp), ("x_out", _vp),
        ("mod", _vp), ("mod_slot_stride", _i),
        ("wqkv", _vp * CHAIN_MAX_DEPTH), ("wo", _vp * CHAIN_MAX_DEPTH), ("wgu", _vp * CHAIN_MAX_DEPTH), ("wd", _vp * CHAIN_MAX_DEPTH),
        ("cache_k", _vp * CHAIN_MAX_DEPTH), ("cache_v", _vp * CHAIN_MAX_DEPTH),
        ("kv_rs", _i),
        ("rope_table", _vp), ("qinfo", _vp), ("kinfo", _vp),
        ("q_scale", _f), ("eps", _f),
        ("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("xa", _vp), ("act", _vp),
        ("attn_scratch", _vp), ("attn_scratch_floats", _ll),
        ("counters", _vp), ("debug_clock", _vp),
        ("packed", _i), ("xs", _vp),
        ("tp_slabs", _vp), ("tp_xs", _vp), ("tp_xn", _vp), ("tp_k", _vp), ("tp_v", _vp),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("q", _vp * 2), ("o", _vp * 2), ("d_o", _vp * 2), ("k", _vp * 2), ("v", _vp * 2),
        ("dq", _vp * 2), ("dk", _vp * 2), ("dv", _vp * 2),
        ("q_len", _i * 2), ("k_len", _i * 2),
        ("q_rs", _i * 2), ("kv_rs", _i * 2), ("o_rs", _i * 2),
        ("qinfo", _vp), ("kinfo", _vp), ("lse", _vp), ("delta", _vp),
        ("scratch", _vp), ("scratch_floats", _ll),
        ("scale", _f), ("hsplit", _i),
        ("B", _i), ("NH", _i), ("NKV", _i), ("HD", _i), ("stop_q1_to_k0", _i),
    ]


# name -> argtypes; every entry point of include/lap_hip.h must be listed here (tests check it).
SIGNATURES: dict[str, list] = {
    "lap_abi_version": [],
    "lap_gemm_bf16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp],
    "lap_gemm_bf16_ex": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "lap_gemm_set_debug": [_i],
    "lap_gemm_asm": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_ok": [_i, _i, _i, _i, _i, _i, _i, _i, _i],
    "lap_gemm_asm_launch_counts": [C.POINTER(C.c_longlong), _i],
    "lap_gemm_asm_bias": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_bias_ok": [_i, _i, _i, _i, _i, _i],
    "lap_gemm_asm_res": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_res_ok": [_i, _i, _i, _i, _i, _i, _i],
    "lap_gemm_asm_geglu_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_geglu_bwd_ok": [_i, _i, _i, _i, _i, _i],
    "lap_gemm_asm_geglu_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_geglu_fwd_ok": [_i, _i, _i, _i, _i, _i, _i],
    "lap_gemm_asm_wgrad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "lap_gemm_wgrad_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _vp],
    "lap_gemm_asm_wgrad_b16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "lap_gemm_wgrad_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _ll, _vp],
    "lap_gemm_asm_bias_gelu": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_asm_gelu_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_gemm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "lap_amax_bf16": [_vp, _ll, _i, _ll, _vp, _vp],
    "lap_quantize_fp8": [_vp, _ll, _i, _ll, _vp, _vp, _ll, _vp, _vp],
    "lap_quantize_fp8_weight": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "lap_gemm_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp],
    "lap_rmsnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "lap_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lap_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "lap_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_layernorm_bwd_sum": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_rope_split_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "lap_rope_split_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "lap_geglu_fwd": [_vp, _vp, _i, _i, _vp],
    "lap_geglu_bwd": [_vp, _vp, _vp, _i, _i, _vp],
    "lap_geglu_fwd_ld": [_vp, _vp, _i, _i, _i, _i, _vp],
    "lap_geglu_bwd_ld": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lap_gelu_fwd": [_vp, _vp, _ll, _vp],
    "lap_gelu_bwd": [_vp, _vp, _vp, _ll, _vp],
    "lap_embed_gather": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "lap_embed_scatter_add": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "lap_gated_residual_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lap_gated_residual_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lap_colsum_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "lap_colsum_f32": [_vp, _vp, _i, _i, _i, _vp],
    "lap_cast_f32_to_bf16": [_vp, _vp, _ll, _vp],
    "lap_split_f32_hilo": [_vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "lap_cast_bf16_to_f32": [_vp, _vp, _ll, _vp],
    "lap_add_bf16": [_vp, _vp, _vp, _ll, _vp],
    "lap_copy2d_bf16": [_vp, _vp, _i, _i, _i, _i, _vp],
    "lap_copy_rows_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lap_im2col_patch": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lap_augment_images": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_add_posemb_cast": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_add_posemb_cast_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_attention_fwd": [C.POINTER(AttnFwdArgs), _vp],
    "lap_attention_serve_splits": [_i, _i, _i, _i, _i],
    "lap_attention_serve": [C.POINTER(AttnFwdArgs), _vp],
    "lap_attention_bwd": [C.POINTER(AttnBwdArgs), _vp],
    "lap_attention_set_variant": [_i],
    "lap_rope_table": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lap_fused_reduce_rope_split": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "lap_fused_reduce_geglu": [_vp, _i, _vp, _i, _i, _vp],
    "lap_fused_reduce_residual_norm": [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp],
    "lap_fused_reduce_norm": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "lap_serve_infos": [C.POINTER(_vp), _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "lap_serve_qkv_rope": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp],
    "lap_serve_gate_up": [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp],
    "lap_serve_proj_residual": [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "lap_serve_set_variant": [_i],
    "lap_serve_embed_actions": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_serve_chain_ok": [_i, _i, _i, _i, _i, _i, _i, _i],
    "lap_serve_chain_tp_ok": [_i, _i, _i, _i, _i, _i, _i, _i],
    "lap_serve_chain_counter_words": [],
    "lap_serve_chain_status": [_vp, C.POINTER(_i)],
    "lap_serve_chain": [C.POINTER(ServeChainArgs), _vp],
    "lap_serve_pack_weight": [_vp, _vp, _i, _i, _i, _i, _vp],
    "lap_serve_final_euler_embed": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "lap_panel_gemm_ok": [_i, _i, _i, _i],
    "lap_panel_gemm_pf": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _i, _vp, _ll, _vp],
    "lap_panel_gemm": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "lap_serve_final_euler": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp],
    "lap_ce_chunk_update": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lap_ce_chunk_grad": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lap_ce_chunk_grad_hilo": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lap_adamw_ema_hilo": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _vp],
    "lap_sumsq_f32": [_vp, _ll, _vp, _vp],
    "lap_sumsq_bf16": [_vp, _ll, _vp, _vp],
    "lap_adamw_ema_g16": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _vp],
    "lap_argmax_rows_f32": [_vp, _i, _i, _i, _vp, _vp],
    "lap_adamw_ema": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _vp],
    "lap_fm_mix": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "lap_posemb_sincos": [_vp, _vp, _i, _i, _f, _f, _vp],
    "lap_swish_fwd": [_vp, _vp, _ll, _vp],
    "lap_swish_bwd": [_vp, _vp, _vp, _ll, _vp],
    "lap_mse_fwd_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "lap_axpy_f32": [_vp, _vp, _f, _ll, _vp],
}

_fn = {}
for _name, _args in SIGNATURES.items():
    _f_ = getattr(_lib, _name)  # AttributeError here = header / library mismatch: fail loudly
    _f_.argtypes = _args
    _f_.restype = C.c_int
    _fn[_name] = _f_

ABI_VERSION = _fn["lap_abi_version"]()
LIB_PATH = str(_LIB_PATH)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _chk(rc: int, name: str) -> None:
    if rc != 0:
        raise LapHipError(f"{name} failed with code {rc}" + (" (LAP_ERR_ARG: rejected arguments)" if rc == 1001 else ""))


def call(name: str, *args) -> None:
    """Raw call: args are python ints/floats/pointers; the current torch stream is appended."""
    _chk(_fn[name](*args, _stream()), name)


def _req(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")


# ------------------------------------------------------------------------------ GEMM
_SCRATCH: dict = {}
_NSPLIT_ENV = __import__("os").environ.get("LAP_ATTN_NSPLIT")
_SERVE_ATTN = __import__("os").environ.get("LAP_SERVE_ATTN", "1") != "0"     # A/B switch: "0" = the generic key-split kernel


def _gemm_scratch(device, floats: int = 160 * 1024 * 1024):
    """Per-stream f32 scratch (640 MB) lent to the GEMM for two-phase split-K; its uses are ordered by that stream."""
    key = (device.type, device.index, torch.cuda.current_stream().cuda_stream)
    t = _SCRATCH.get(key)
    if t is None:
        t = torch.empty(floats, dtype=torch.float32, device=device)
        _SCRATCH[key] = t
    return t


def gemm(a, b, out, *, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, bias=None, residual=None, ldr=0,
         alpha=1.0, gelu=False, accum=False, tile=-1, ksplit=0):
    """out[M,N] = epi(alpha * opA . opB); see include/lap_hip.h lap_gemm_bf16 / lap_gemm_bf16_ex."""
    _req(a, torch.bfloat16, "A"); _req(b, torch.bfloat16, "B")
    flags = 0
    if out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    elif out.dtype != torch.bfloat16:
        raise TypeError("gemm out must be bf16 or f32")
    if accum:
        flags |= GEMM_ACCUM
    if gelu:
        flags |= GEMM_GELU | (GEMM_GELU_BF16 if gelu == "bf16" else 0)   # "bf16": round the pre-activation to bf16 first
    if bias is not None and bias.dtype == torch.float32:
        flags |= GEMM_BIAS_F32
    # the library decides on two-phase split-K itself when it is lent scratch (poorly filled grids, skinny-M serving); an
    # accumulating product gets it too when its output is small and its contraction long (the LM head's hi / lo data gradients:
    # 16 output tiles over K = 257,152 — unsplit that is 16 CUs walking 4,018 k-tiles; the reduce pass adds onto C)
    scratch = _gemm_scratch(a.device) if (ksplit == 0 and (not accum or (K >= 16384 and M * N <= (1 << 22)))) else None
    call("lap_gemm_bf16_ex", _p(a), _p(b), _p(out), _p(bias), _p(residual), M, N, K, lda, ldb, ldc, ldr, float(alpha),
         int(a_kc), int(b_kc), flags, tile, ksplit, _p(scratch), scratch.numel() * 4 if scratch is not None else 0)
    return out


def linear_fwd(x, wt, out=None, *, bias=None, residual=None, gelu=False, out_dtype=torch.bfloat16, tile=-1, ksplit=0):
    """y[M,out] = x[M,in] @ wt[out,in]^T (+bias)(gelu)(+residual)."""
    M, K = x.shape
    N = wt.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=x.device)
    return gemm(x, wt, out, M=M, N=N, K=K, lda=x.stride(0), ldb=wt.stride(0), ldc=out.stride(0), bias=bias,
                residual=residual, ldr=(residual.stride(0) if residual is not None else 0), gelu=gelu, tile=tile,
                ksplit=ksplit)


def linear_geglu(x, wgu, exp2=False):
    """act[M, H] = GeGLU(x @ wgu[2H, in]^T): the gate|up projection and lap_geglu_fwd in one launch (serving prefill, M <= 640).
    exp2: the GELU through v_exp / v_rcp (the training kernel lap_gemm_asm_geglu_fwd's arithmetic) instead of tanhf."""
    M, K = x.shape
    H = wgu.shape[0] // 2
    act = torch.empty((M, H), dtype=torch.bfloat16, device=x.device)
    call("lap_gemm_bf16_ex", _p(x), _p(wgu), _p(act), None, None, M, 2 * H, K, x.stride(0), wgu.stride(0), H, 0, 1.0, 1, 1,
         GEMM_GEGLU | (GEMM_GELU_EXP2 if exp2 else 0), -1, 0,
         None, 0)
    return act


def linear_dgrad(dy, wt, out=None, *, out_dtype=torch.bfloat16, accum=False, tile=-1, ksplit=0):
    """dx[M,in] = dy[M,out] @ wt[out,in]."""
    M, K = dy.shape
    N = wt.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=dy.device)
    return gemm(dy, wt, out, M=M, N=N, K=K, lda=dy.stride(0), ldb=wt.stride(0), ldc=out.stride(0), a_kc=True,
                b_kc=False, accum=accum, tile=tile, ksplit=ksplit)


def linear_wgrad(dy, x, out, *, accum=False, ksplit=0, tile=-1):
    """dWt[out,in] (f32) = dy[M,out]^T @ x[M,in].  ksplit > 1 needs accum=True (adds onto `out`)."""
    Mrows, Nout = dy.shape
    Kin = x.shape[1]
    return gemm(dy, x, out, M=Nout, N=Kin, K=Mrows, lda=dy.stride(0), ldb=x.stride(0), ldc=out.stride(0), a_kc=False,
                b_kc=False, accum=accum, ksplit=ksplit, tile=tile)


ASM_KERNELS = ("nt", "nn", "tn", "nt_bias", "tn_t", "nt_res", "nt_bias_res", "nn_geglu_bwd", "nt_geglu", "nn_gelu_bwd", "nt_bias_gelu",
               "tn_b16", "tn_t_b16")


def gemm_asm_launch_counts() -> dict:
    """Launches of each assembly GEMM kernel since the library was loaded (include/lap_hip.h: lap_gemm_asm_launch_counts)."""
    buf = (C.c_longlong * len(ASM_KERNELS))()
    _chk(_fn["lap_gemm_asm_launch_counts"](buf, len(ASM_KERNELS)), "lap_gemm_asm_launch_counts")
    return dict(zip(ASM_KERNELS, (int(v) for v in buf)))


def linear_wgrad_sumsq(dy, x, out, sumsq):
    """dWt[out, in] (f32, or bf16: ParamStore.grad_dtype) = dy^T x like linear_wgrad, and where the assembly kernel takes the product sum(dWt^2) is added to the
    one-element f32 tensor `sumsq` on the way out; returns whether that happened (else the caller still owes the norm a pass)."""
    import ctypes
    Mrows, Nout = dy.shape
    Kin = x.shape[1]
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(sumsq, torch.float32, "sumsq")
    if out.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("linear_wgrad_sumsq: out must be f32 or bf16")
    folded = ctypes.c_int(0)
    scratch = _gemm_scratch(dy.device)
    call("lap_gemm_wgrad_f32" if out.dtype == torch.float32 else "lap_gemm_wgrad_bf16", _p(dy), _p(x), _p(out), Nout, Kin, Mrows, dy.stride(0), x.stride(0), out.stride(0), _p(sumsq),
         ctypes.byref(folded), _p(scratch), scratch.numel() * 4)
    return bool(folded.value)


def split_f32_hilo(x, ld_out=None):
    """x f32 [rows, cols] -> (hi, lo) bf16 [rows, ld_out] with hi + lo ~ x (16 mantissa bits), zero padded columns."""
    _req(x, torch.float32, "x")
    rows, cols = x.shape
    ld_out = (cols + 7) // 8 * 8 if ld_out is None else ld_out
    hi = torch.empty((rows, ld_out), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty_like(hi)
    call("lap_split_f32_hilo", _p(x), rows, cols, x.stride(0), _p(hi), _p(lo), ld_out)
    return hi, lo


def argmax_rows(x, out=None):
    """x f32 [rows, n] (row stride x.stride(0)) -> int32 [rows], lowest index among ties."""
    _req(x, torch.float32, "x")
    if out is None:
        out = torch.empty((x.shape[0],), dtype=torch.int32, device=x.device)
    call("lap_argmax_rows_f32", _p(x), x.shape[0], x.shape[1], x.stride(0), _p(out))
    return out


def gemm_f32(a, b, out, *, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, bias=None, alpha=1.0, accum=False):
    for t, n in ((a, "A"), (b, "B"), (out, "C")):
        _req(t, torch.float32, n)
    call("lap_gemm_f32", _p(a), _p(b), _p(out), _p(bias), M, N, K, lda, ldb, ldc, float(alpha), int(a_kc), int(b_kc),
         int(accum))
    return out


# ------------------------------------------------------------------- normalisation
def rmsnorm_fwd(x, scale=None, mod=None, rows_per_sample=0, eps=1e-6, save_rstd=True, mod_ld=None):
    """mod: bf16 [B, >=3D] view (row stride = mod.stride(0)) holding scale|shift|gate; mod_ld=0 shares row 0
    between all samples (serving: the condition depends on the denoise time only)."""
    rows, D = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_rstd else None
    if mod_ld is None:
        mod_ld = mod.stride(0) if mod is not None else 0
    call("lap_rmsnorm_fwd", _p(x), _p(scale), _p(mod), _p(y), _p(rstd), rows, D, rows_per_sample, mod_ld, float(eps))
    return y, rstd


def rmsnorm_bwd(x, dy, rstd, scale=None, mod=None, rows_per_sample=0, dx=None, dscale=None, dmod=None, accum_dx=False):
    rows, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    call("lap_rmsnorm_bwd", _p(x), _p(scale), _p(mod), _p(rstd), _p(dy), _p(dx), _p(dscale), _p(dmod), rows, D,
         rows_per_sample, mod.stride(0) if mod is not None else 0, dmod.stride(0) if dmod is not None else 0,
         int(accum_dx))
    return dx


def layernorm_fwd(x, gamma, beta, eps=1e-6):
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("lap_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, float(eps))
    return y, mean, rstd


def layernorm_bwd(x, dy, gamma, mean, rstd, dgamma, dbeta, dx=None, accum_dx=False, dxsum=None):
    """dxsum: f32 [D] that the column sums of the returned dx are ADDED to (a bias gradient for free), or None."""
    rows, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    call("lap_layernorm_bwd_sum", _p(x), _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dx), _p(dgamma), _p(dbeta), _p(dxsum), rows, D,
         int(accum_dx))
    return dx


# --------------------------------------------------------------------- elementwise
def rope_split_fwd(qkv, pos, B, T_seg, T_total, seg_off, NH, HD, q_scale):
    rows = B * T_seg
    q = torch.empty((rows, NH * HD), dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty((rows, HD), dtype=torch.bfloat16, device=qkv.device)
    v = torch.empty((rows, HD), dtype=torch.bfloat16, device=qkv.device)
    call("lap_rope_split_fwd", _p(qkv), _p(pos), _p(q), _p(k), _p(v), B, T_seg, T_total, seg_off, NH, HD, float(q_scale))
    return q, k, v


def rope_split_bwd(dq, dk, dv, pos, B, T_seg, T_total, seg_off, NH, HD, q_scale):
    dqkv = torch.empty((B * T_seg, (NH + 2) * HD), dtype=torch.bfloat16, device=dq.device)
    call("lap_rope_split_bwd", _p(dq), _p(dk), _p(dv), _p(pos), _p(dqkv), B, T_seg, T_total, seg_off, NH, HD,
         float(q_scale))
    return dqkv


def _padded_rows(rows, cols, device, pad):
    """[rows, cols] bf16 view of a buffer whose rows are `pad` elements longer: see lap_geglu_fwd_ld in include/lap_hip.h"""
    return torch.empty((rows, cols + pad), dtype=torch.bfloat16, device=device)[:, :cols]


def _row_pad(cols):
    """64 elements when the dense row would be a multiple of 16 KiB (the bad strides measured: 32 and 64 KiB), else none"""
    return 64 if cols * 2 >= 16384 and (cols * 2) % 16384 == 0 else 0


def geglu_fwd(gu, pad=False):
    rows, H2 = gu.shape
    act = _padded_rows(rows, H2 // 2, gu.device, _row_pad(H2 // 2) if pad else 0)
    call("lap_geglu_fwd_ld", _p(gu), _p(act), rows, H2 // 2, gu.stride(0), act.stride(0))
    return act


def linear_bias_gelu_train_ok(x, wt, bias):
    M, K = x.shape
    N = wt.shape[0]
    return (x.dtype == wt.dtype == torch.bfloat16 and bias is not None and bias.dtype == torch.float32 and x.stride(1) == 1 and wt.stride(1) == 1
            and bool(_lib.lap_gemm_asm_bias_ok(M, N, K, x.stride(0), wt.stride(0), N)))


def linear_bias_gelu_train(x, wt, bias):
    """(h, a) = (x @ wt^T + bias, gelu(h)) in one launch, both kept (training: the backward pass reads h)"""
    M, K = x.shape
    N = wt.shape[0]
    h = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    a = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    call("lap_gemm_asm_bias_gelu", _p(x), _p(wt), _p(h), _p(a), _p(bias), M, N, K, x.stride(0), wt.stride(0), N)
    return h, a


def dgrad_gelu_bwd_ok(dy, w, h):
    M, K = dy.shape
    N = w.shape[1]
    return (dy.dtype == w.dtype == h.dtype == torch.bfloat16 and tuple(h.shape) == (M, N) and h.stride(1) == 1 and dy.stride(1) == 1 and w.stride(1) == 1
            and w.shape[0] == K and bool(_lib.lap_gemm_asm_ok(1, 0, 0, M, N, K, dy.stride(0), w.stride(0), h.stride(0))))


def linear_dgrad_gelu_bwd(dy, w, h):
    """d(h) = gelu_bwd(h, dy @ w) in one launch (the second Dense's data gradient with the GELU backward as its epilogue)"""
    M, K = dy.shape
    N = w.shape[1]
    if tuple(h.shape) != (M, N) or w.shape[0] != K:
        raise LapHipError(f"linear_dgrad_gelu_bwd: dy {tuple(dy.shape)}, w {tuple(w.shape)}, h {tuple(h.shape)} do not fit")
    dh = torch.empty((M, h.stride(0)), dtype=torch.bfloat16, device=h.device)[:, :N]
    call("lap_gemm_asm_gelu_bwd", _p(dy), _p(w), _p(dh), _p(h), M, N, K, dy.stride(0), w.stride(0), h.stride(0))
    return dh


def linear_geglu_train_ok(x, wgu, pad=True):
    M, K = x.shape
    N = wgu.shape[0]
    p = _row_pad(N) if pad else 0
    pa = _row_pad(N // 2) if pad else 0
    return (x.dtype == wgu.dtype == torch.bfloat16 and x.stride(1) == 1 and wgu.stride(1) == 1 and N % 256 == 0
            and bool(_lib.lap_gemm_asm_geglu_fwd_ok(M, N, K, x.stride(0), wgu.stride(0), N + p, N // 2 + pa)))


def linear_geglu_train(x, wgu, pad=True):
    """(gu, act) = (x @ wgu^T, GeGLU(gu)) in one launch, both kept (training: the backward pass reads gu); rows padded off the
    16 KiB strides like geglu_fwd / geglu_bwd do."""
    M, K = x.shape
    N = wgu.shape[0]
    gu = _padded_rows(M, N, x.device, _row_pad(N) if pad else 0)
    act = _padded_rows(M, N // 2, x.device, _row_pad(N // 2) if pad else 0)
    call("lap_gemm_asm_geglu_fwd", _p(x), _p(wgu), _p(gu), _p(act), M, N, K, x.stride(0), wgu.stride(0), gu.stride(0), act.stride(0))
    return gu, act


def dgrad_geglu_bwd_ok(dy, w, gu):
    """whether linear_dgrad_geglu_bwd takes (dy [M, K], w [K, N], gu [M, 2N] with any row stride >= 2N)"""
    M, K = dy.shape
    N = w.shape[1]
    return (dy.dtype == w.dtype == gu.dtype == torch.bfloat16 and gu.shape == (M, 2 * N) and gu.stride(1) == 1 and dy.stride(1) == 1 and w.stride(1) == 1
            and bool(_lib.lap_gemm_asm_geglu_bwd_ok(M, N, K, dy.stride(0), w.stride(0), gu.stride(0))))


def linear_dgrad_geglu_bwd(dy, w, gu):
    """dgu = geglu_bwd(gu, dy @ w) in one launch (the down projection's data gradient with the GeGLU backward as its epilogue);
    dgu gets gu's row stride."""
    M, K = dy.shape
    N = w.shape[1]
    if tuple(gu.shape) != (M, 2 * N) or w.shape[0] != K:
        raise LapHipError(f"linear_dgrad_geglu_bwd: dy {tuple(dy.shape)}, w {tuple(w.shape)}, gate|up {tuple(gu.shape)} do not fit")
    ld = gu.stride(0)
    dgu = torch.empty((M, ld), dtype=torch.bfloat16, device=gu.device)[:, :2 * N]
    call("lap_gemm_asm_geglu_bwd", _p(dy), _p(w), _p(dgu), _p(gu), M, N, K, dy.stride(0), w.stride(0), ld)
    return dgu


def geglu_bwd(gu, dact, pad=False):
    rows, H2 = gu.shape
    dgu = _padded_rows(rows, H2, gu.device, _row_pad(H2) if pad else 0)
    call("lap_geglu_bwd_ld", _p(gu), _p(dact), _p(dgu), rows, H2 // 2, gu.stride(0), dact.stride(0), dgu.stride(0))
    return dgu


def gelu_fwd(x):
    y = torch.empty_like(x)
    call("lap_gelu_fwd", _p(x), _p(y), x.numel())
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    call("lap_gelu_bwd", _p(x), _p(dy), _p(dx), x.numel())
    return dx


def embed_gather(table, tok, out, rows, T, D, dst_rps, dst_off, scale, row_lo=0, row_hi=None):
    """table: f32 rows [row_lo, row_hi) of the vocabulary (default: the whole table)."""
    if row_hi is None:
        row_hi = row_lo + table.shape[0]
    call("lap_embed_gather", _p(table), _p(tok), _p(out), rows, T, D, dst_rps, dst_off, float(scale), row_lo, row_hi)


def embed_scatter_add(dtable, tok, dout, rows, T, D, src_rps, src_off, scale):
    call("lap_embed_scatter_add", _p(dtable), _p(tok), _p(dout), rows, T, D, src_rps, src_off, float(scale))


def gated_residual_fwd(x, u, gate=None, rows_per_sample=0, ldg=0):
    y = torch.empty_like(x)
    call("lap_gated_residual_fwd", _p(x), _p(u), _p(gate), _p(y), x.shape[0], x.shape[1], rows_per_sample, ldg)
    return y


def gated_residual_bwd(dy, u, gate, rows_per_sample, ldg, dgate, ldg_out):
    du = torch.empty_like(dy)
    call("lap_gated_residual_bwd", _p(dy), _p(u), _p(gate), _p(du), _p(dgate), dy.shape[0], dy.shape[1],
         rows_per_sample, ldg, ldg_out)
    return du


def colsum(x, out, rows=None, cols=None, ld=None):
    """out[c] += sum_r x[r, c] (f32 atomics; caller zeroes)."""
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    ld = x.stride(0) if ld is None else ld
    call("lap_colsum_bf16" if x.dtype == torch.bfloat16 else "lap_colsum_f32", _p(x), _p(out), rows, cols, ld)


def cast_f32_to_bf16(x, y=None):
    if y is None:
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("lap_cast_f32_to_bf16", _p(x), _p(y), x.numel())
    return y


def cast_bf16_to_f32(x, y=None):
    if y is None:
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("lap_cast_bf16_to_f32", _p(x), _p(y), x.numel())
    return y


def add_bf16(a, b, y=None):
    if y is None:
        y = torch.empty_like(a)
    call("lap_add_bf16", _p(a), _p(b), _p(y), a.numel())
    return y


def copy2d_bf16(src, dst, rows, cols, lds, ldd):
    call("lap_copy2d_bf16", _p(src), _p(dst), rows, cols, lds, ldd)


def copy_rows_bf16(src, dst, rows, T, D, src_rps, src_off, dst_rps, dst_off, accumulate=False):
    call("lap_copy_rows_bf16", _p(src), _p(dst), rows, T, D, src_rps, src_off, dst_rps, dst_off, int(accumulate))


def im2col_patch(img, P):
    B, H, W, Cc = img.shape
    out = torch.empty((B * (H // P) * (W // P), P * P * Cc), dtype=torch.float32, device=img.device)
    call("lap_im2col_patch", _p(img), _p(out), B, H, W, Cc, P)
    return out


def add_posemb_cast(x, pos, T):
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device)
    call("lap_add_posemb_cast", _p(x), _p(pos), _p(y), rows, T, D)
    return y


def add_posemb_cast_bwd(dy, dpos, T):
    rows, D = dy.shape
    dx = torch.empty((rows, D), dtype=torch.float32, device=dy.device)
    call("lap_add_posemb_cast_bwd", _p(dy), _p(dx), _p(dpos), rows, T, D)
    return dx


# ----------------------------------------------------------------------- attention
def attention_set_variant(variant: int):
    """Test / benchmark knob: -1 automatic, 0 generic kernels, 1 / 2 row groups of the HD = 256 LDS-DMA kernels."""
    _chk(_fn["lap_attention_set_variant"](int(variant)), "lap_attention_set_variant")


def attention_fwd(q, k, v, q_len, k_len, B, NH, NKV, HD, qinfo=None, kinfo=None, need_lse=True, scale=1.0,
                  q_rs=(0, 0), kv_rs=(0, 0), nsplit_hint=None):
    """q/k/v: lists of up to two segment tensors (None for an empty segment).  q_rs / kv_rs: row strides in
    elements when q / k / v are column slices of a wider (fused qkv) buffer; outputs are packed [B*len, NH*HD]."""
    a = AttnFwdArgs()
    outs = []
    for s in range(2):
        qs = q[s] if s < len(q) else None
        a.q[s] = _p(qs)
        a.q_rs[s], a.kv_rs[s], a.o_rs[s] = q_rs[s], kv_rs[s], 0
        o = None
        if qs is not None:
            o = torch.empty((B * q_len[s], NH * HD), dtype=torch.bfloat16, device=qs.device)
        outs.append(o)
        a.o[s] = _p(o)
        a.k[s] = _p(k[s]) if s < len(k) and k[s] is not None else None
        a.v[s] = _p(v[s]) if s < len(v) and v[s] is not None else None
        a.q_len[s] = q_len[s] if s < len(q_len) else 0
        a.k_len[s] = k_len[s] if s < len(k_len) else 0
    for nm, inf in (("qinfo", qinfo), ("kinfo", kinfo)):
        if inf is not None and (inf.dtype != torch.int32 or not inf.is_contiguous()):
            raise TypeError(f"{nm} must be a contiguous int32 tensor, got {inf.dtype}")
    Tq = a.q_len[0] + a.q_len[1]
    dev = next(t for t in q if t is not None).device
    lse = torch.empty((B, NH, Tq), dtype=torch.float32, device=dev) if need_lse else None
    a.qinfo, a.kinfo, a.lse = _p(qinfo), _p(kinfo), _p(lse)
    a.scale = float(scale)
    a.B, a.NH, a.NKV, a.HD = B, NH, NKV, HD
    # batch-1 denoise step: a handful of suffix queries against [KV cache | fresh keys] -> the load-everything-up-front kernel
    if _SERVE_ATTN and HD == 256 and a.q_len[0] == 0 and 0 < a.q_len[1] <= 64 and not need_lse and nsplit_hint is None and scale > 0:
        ns = _fn["lap_attention_serve_splits"](a.k_len[0], a.k_len[1], B, NH, a.q_len[1])
        if 0 < ns <= 16:
            a.nsplit = ns
            scratch = torch.empty(ns * B * Tq * NH * (HD + 1), dtype=torch.float32, device=dev)
            a.scratch, a.scratch_floats = _p(scratch), scratch.numel()
            _chk(_fn["lap_attention_serve"](C.byref(a), _stream()), "lap_attention_serve")
            return outs, None
    # few query tiles (serving): split the key tiles over more blocks
    ntq = (a.q_len[0] + 63) // 64 + (a.q_len[1] + 63) // 64
    ntk = (a.k_len[0] + 63) // 64 + (a.k_len[1] + 63) // 64
    blocks = B * NH * ntq
    nsplit = 1
    if nsplit_hint is None and _NSPLIT_ENV and blocks < 128 and ntk > 1:
        nsplit_hint = min(int(_NSPLIT_ENV), ntk)     # tuning knob (tools/bench_serve_split.py)
    if nsplit_hint is not None:
        nsplit = nsplit_hint
    elif blocks < 128 and ntk > 1:
        nsplit = min((ntk + 1) // 2, max(1, 256 // blocks))   # >= 128 keys per split: the combine cost grows with the split count
    a.nsplit = nsplit
    if nsplit > 1:
        scratch = torch.empty(nsplit * B * Tq * NH * (HD + 1), dtype=torch.float32, device=dev)
        a.scratch, a.scratch_floats = _p(scratch), scratch.numel()
    a.B, a.NH, a.NKV, a.HD = B, NH, NKV, HD
    _chk(_fn["lap_attention_fwd"](C.byref(a), _stream()), "lap_attention_fwd")
    return outs, lse


def attention_bwd(q, k, v, o, d_o, lse, q_len, k_len, B, NH, NKV, HD, qinfo=None, kinfo=None, stop_q1_to_k0=False,
                  scale=1.0, q_rs=(0, 0), kv_rs=(0, 0), dq_out=None, dk_out=None, dv_out=None):
    """Gradients are written with the same row strides as q / k / v (pass dq_out/dk_out/dv_out views of a fused
    dqkv buffer when q_rs / kv_rs are set)."""
    a = AttnBwdArgs()
    dq, dk, dv = [], [], []
    for s in range(2):
        def g(lst):
            return lst[s] if lst is not None and s < len(lst) else None
        a.q[s], a.o[s], a.d_o[s] = _p(g(q)), _p(g(o)), _p(g(d_o))
        a.k[s], a.v[s] = _p(g(k)), _p(g(v))
        a.q_rs[s], a.kv_rs[s], a.o_rs[s] = q_rs[s], kv_rs[s], 0
        dq.append(g(dq_out) if g(dq_out) is not None else (torch.empty_like(g(q)) if g(q) is not None else None))
        dk.append(g(dk_out) if g(dk_out) is not None else (torch.empty_like(g(k)) if g(k) is not None else None))
        dv.append(g(dv_out) if g(dv_out) is not None else (torch.empty_like(g(v)) if g(v) is not None else None))
        a.dq[s], a.dk[s], a.dv[s] = _p(dq[s]), _p(dk[s]), _p(dv[s])
        a.q_len[s] = q_len[s] if s < len(q_len) else 0
        a.k_len[s] = k_len[s] if s < len(k_len) else 0
    for nm, inf in (("qinfo", qinfo), ("kinfo", kinfo)):
        if inf is not None and (inf.dtype != torch.int32 or not inf.is_contiguous()):
            raise TypeError(f"{nm} must be a contiguous int32 tensor, got {inf.dtype}")
    Tq = a.q_len[0] + a.q_len[1]
    delta = torch.empty((B, NH, Tq), dtype=torch.float32, device=lse.device)
    a.qinfo, a.kinfo, a.lse, a.delta = _p(qinfo), _p(kinfo), _p(lse), _p(delta)
    a.scale = float(scale)
    hpk = NH // NKV
    hsplit = 1
    if hpk > 1:  # MQA / GQA: more blocks for the dK/dV kernel
        hsplit = 4 if hpk % 4 == 0 else (2 if hpk % 2 == 0 else 1)
    a.hsplit = hsplit
    if hsplit > 1:
        Tk = a.k_len[0] + a.k_len[1]
        scratch = torch.empty(2 * hsplit * B * Tk * NKV * HD, dtype=torch.float32, device=lse.device)
        a.scratch, a.scratch_floats = _p(scratch), scratch.numel()
    a.B, a.NH, a.NKV, a.HD, a.stop_q1_to_k0 = B, NH, NKV, HD, int(stop_q1_to_k0)
    _chk(_fn["lap_attention_bwd"](C.byref(a), _stream()), "lap_attention_bwd")
    return dq, dk, dv


# ----------------------------------------------------------------- loss / optimizer
def ce_chunk_update(logits, target, m, l, tl, v0):
    rows, vc = logits.shape
    call("lap_ce_chunk_update", _p(logits), logits.stride(0), _p(target), _p(m), _p(l), _p(tl), rows, v0, vc)


def ce_chunk_grad(logits, target, m, l, w, dlogits, v0, dlogits_lo=None):
    """dlogits (bf16) = w * (softmax - onehot); with `dlogits_lo` also the bf16 residual plane (hi + lo ~ the f32 value)."""
    rows, vc = logits.shape
    if dlogits_lo is None:
        call("lap_ce_chunk_grad", _p(logits), logits.stride(0), _p(target), _p(m), _p(l), _p(w), _p(dlogits),
             dlogits.stride(0), rows, v0, vc)
    else:
        if dlogits_lo.stride(0) != dlogits.stride(0):
            raise ValueError("hi / lo planes must share the row stride")
        call("lap_ce_chunk_grad_hilo", _p(logits), logits.stride(0), _p(target), _p(m), _p(l), _p(w), _p(dlogits), _p(dlogits_lo),
             dlogits.stride(0), rows, v0, vc)


def sumsq_f32(x, out):
    """out[0] += sum x^2; x: an f32 or a bf16 gradient buffer (squares accumulated in f32 either way)."""
    if x.dtype == torch.bfloat16:
        call("lap_sumsq_bf16", _p(x), x.numel(), _p(out))
        return
    call("lap_sumsq_f32", _p(x), x.numel(), _p(out))


def adamw_ema(p, m, v, ema, g, p16, scalars, b1, b2, eps, wd, max_norm, p16lo=None):
    if g.dtype == torch.bfloat16:       # GEMM-produced weight gradients (round 5): bf16 gradient buffer
        if p16lo is not None:
            raise ValueError("the hi / lo unit (embedding table) keeps f32 gradients")
        call("lap_adamw_ema_g16", _p(p), _p(m), _p(v), _p(ema), _p(g), _p(p16), p.numel(), _p(scalars), float(b1), float(b2),
             float(eps), float(wd), float(max_norm))
        return
    if p16lo is not None:
        call("lap_adamw_ema_hilo", _p(p), _p(m), _p(v), _p(ema), _p(g), _p(p16), _p(p16lo), p.numel(), _p(scalars), float(b1), float(b2),
             float(eps), float(wd), float(max_norm))
        return
    call("lap_adamw_ema", _p(p), _p(m), _p(v), _p(ema), _p(g), _p(p16), p.numel(), _p(scalars), float(b1), float(b2),
         float(eps), float(wd), float(max_norm))


def fm_mix(noise, actions, t):
    B = noise.shape[0]
    n_per = noise.numel() // B
    x_t = torch.empty_like(noise); u_t = torch.empty_like(noise)
    call("lap_fm_mix", _p(noise), _p(actions), _p(t), _p(x_t), _p(u_t), B, n_per)
    return x_t, u_t


def posemb_sincos(t, D, min_period, max_period):
    out = torch.empty((t.shape[0], D), dtype=torch.float32, device=t.device)
    call("lap_posemb_sincos", _p(t), _p(out), t.shape[0], D, float(min_period), float(max_period))
    return out


def swish_fwd(x):
    y = torch.empty_like(x)
    call("lap_swish_fwd", _p(x), _p(y), x.numel())
    return y


def swish_bwd(x, dy):
    dx = torch.empty_like(x)
    call("lap_swish_bwd", _p(x), _p(dy), _p(dx), x.numel())
    return dx


def mse_fwd_bwd(v, u, coef=None, need_grad=True):
    B = v.shape[0]
    n_per = v.numel() // B
    per = torch.empty(B, dtype=torch.float32, device=v.device)
    dv = torch.empty_like(v) if need_grad else None
    call("lap_mse_fwd_bwd", _p(v), _p(u), _p(coef), _p(per), _p(dv), B, n_per)
    return per, dv


def axpy_f32(x, v, dt):
    call("lap_axpy_f32", _p(x), _p(v), float(dt), x.numel())


# ------------------------------------------------ split-K partials + fused consumers (batch-1 denoise step)
def linear_partials(x, wt, scratch, ksplit=None, tile=6):
    """Raw f32 partial products of y = x @ wt^T, [ksplit, M, N] in `scratch` (a 1-D f32 tensor); returns (view, ksplit)."""
    M, K = x.shape
    N = wt.shape[0]
    if ksplit is None:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        ksplit = max(1, min(K // 256, 256 // max(tiles, 1), 16))   # (K // 128, 384-512 blocks, K // 512 measured: no better)
    need = ksplit * M * N
    if scratch.numel() < need:
        raise ValueError("scratch too small for the requested split")
    call("lap_gemm_bf16_ex", _p(x), _p(wt), None, None, None, M, N, K, x.stride(0), wt.stride(0), N, 0, 1.0, 1, 1,
         GEMM_OUT_F32 | GEMM_PARTIALS, tile, ksplit, _p(scratch), scratch.numel() * 4)
    return scratch[:need].view(ksplit, M, N), ksplit


def serve_infos(img_masks, T_img, prompt_mask, langact_mask, S, suffix_idx):
    """Token info words / positions of the sampler (lap.py:624-654) in one launch: (qinfo_p, kinfo_p, ppos, qinfo_s, kinfo_all,
    pos_all), int32; masks are torch.bool tensors on the device."""
    B, Lt = prompt_mask.shape
    for t in (*img_masks, prompt_mask, *(() if langact_mask is None else (langact_mask,))):
        if t.dtype != torch.bool or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("serve_infos: contiguous cuda bool masks")
    Pn = len(img_masks) * T_img + Lt
    dev = prompt_mask.device
    mk = lambda n: torch.empty((B, n), dtype=torch.int32, device=dev)
    outs = (mk(Pn), mk(Pn), mk(Pn), mk(S), mk(Pn + S), mk(Pn + S))
    arr = (_vp * max(len(img_masks), 1))(*[_p(t) for t in img_masks])
    call("lap_serve_infos", arr, len(img_masks), T_img, _p(prompt_mask), _p(langact_mask), B, Lt, S, suffix_idx, *[_p(t) for t in outs])
    return outs


def fused_reduce_norm(part, ksplit, rows, D, *, bias=None, residual=None, norm=0, gamma=None, beta=None, eps=1e-6):
    """xn = bf16(sum of the split-K slabs (+ f32 bias) (+ bf16 residual)); h = RMSNorm (norm=1, gamma = the f32 scale) or
    LayerNorm (norm=2) of xn, or None (norm=0).  One pass: the serving prefill's reduce + epilogue + next norm."""
    xn = torch.empty((rows, D), dtype=torch.bfloat16, device=part.device)
    h = torch.empty_like(xn) if norm else None
    call("lap_fused_reduce_norm", _p(part), ksplit, _p(bias), _p(residual), norm, _p(gamma), _p(beta), _p(xn), _p(h), rows, D, float(eps))
    return xn, h


def augment_images(img, params):
    """img f32 [B, H, W, 3] in [-1, 1], params f32 [B, 12] (include/lap_hip.h) -> augmented copy."""
    _req(img, torch.float32, "img"); _req(params, torch.float32, "params")
    B, H, W, C = img.shape
    if C != 3 or params.shape != (B, 12) or not img.is_contiguous() or not params.is_contiguous():
        raise ValueError("augment_images: img [B, H, W, 3] and params [B, 12], both contiguous")
    out = torch.empty_like(img)
    call("lap_augment_images", _p(img), _p(out), _p(params), B, H, W)
    return out


def rope_table(pos, B, T_seg, T_total, seg_off, HD):
    """sin / cos of a segment's positions, f32 [B*T_seg, HD/2, 2] (hoisted out of the denoise loop's RoPE kernels)."""
    tab = torch.empty((B * T_seg, HD // 2, 2), dtype=torch.float32, device=pos.device)
    call("lap_rope_table", _p(pos), _p(tab), B, T_seg, T_total, seg_off, HD)
    return tab


def fused_reduce_rope_split(part, ksplit, pos, B, T_seg, T_total, seg_off, NH, HD, q_scale, table=None):
    rows = B * T_seg
    dev = part.device
    q = torch.empty((rows, NH * HD), dtype=torch.bfloat16, device=dev)
    k = torch.empty((rows, HD), dtype=torch.bfloat16, device=dev)
    v = torch.empty((rows, HD), dtype=torch.bfloat16, device=dev)
    call("lap_fused_reduce_rope_split", _p(part), ksplit, _p(pos), _p(table), _p(q), _p(k), _p(v), B, T_seg, T_total, seg_off, NH, HD,
         float(q_scale))
    return q, k, v


def fused_reduce_geglu(part, ksplit, rows, H):
    act = torch.empty((rows, H), dtype=torch.bfloat16, device=part.device)
    call("lap_fused_reduce_geglu", _p(part), ksplit, _p(act), rows, H)
    return act


def fused_reduce_residual_norm(part, ksplit, x, gate, ldg, mod, mod_ld, rows_per_sample, eps=1e-6):
    rows, D = x.shape
    xn = torch.empty_like(x)
    h = torch.empty_like(x) if mod is not None else None
    call("lap_fused_reduce_residual_norm", _p(part), ksplit, _p(x), _p(gate), ldg, _p(mod), mod_ld, _p(xn), _p(h), rows, D,
         rows_per_sample, float(eps))
    return xn, h


# ------------------------------------------------ skinny-M fused projections (batch-1 denoise step, csrc/serve_skinny.hip)
def serve_supported(D, HD, mlp_dim, NH):
    """Shapes the skinny kernels are built for (the LAP-3B action expert); anything else takes the partials + consumers path."""
    return D == 1024 and HD == 256 and NH * HD in (1024, 2048, 4096) and mlp_dim in (1024, 2048, 4096)


def serve_qkv_rope(x, mod, mod_ld, rps, wqkv, table, NH, HD, q_scale, eps=1e-6):
    M, D = x.shape
    q = torch.empty((M, NH * HD), dtype=torch.bfloat16, device=x.device)
    k = torch.empty((M, HD), dtype=torch.bfloat16, device=x.device)
    v = torch.empty((M, HD), dtype=torch.bfloat16, device=x.device)
    call("lap_serve_qkv_rope", _p(x), _p(mod), mod_ld, rps, _p(wqkv), _p(table), _p(q), _p(k), _p(v), M, D, NH, HD, float(q_scale), float(eps))
    return q, k, v


def serve_gate_up(x, mod, mod_ld, rps, wgu, eps=1e-6):
    M, D = x.shape
    H = wgu.shape[0] // 2
    act = torch.empty((M, H), dtype=torch.bfloat16, device=x.device)
    call("lap_serve_gate_up", _p(x), _p(mod), mod_ld, rps, _p(wgu), _p(act), M, D, H, float(eps))
    return act


def serve_proj_residual(a, w, x, gate, gate_ld, rps):
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    call("lap_serve_proj_residual", _p(a), _p(w), _p(x), _p(gate), gate_ld, rps, _p(out), M, N, K)
    return out


def serve_chain_ok(B, S, D, H, NH, HD, NKV, prefix_len) -> bool:
    """Shapes the one-launch denoise step (lap_serve_chain) is built for."""
    return bool(_fn["lap_serve_chain_ok"](B, S, D, H, NH, HD, NKV, prefix_len))


def serve_chain_counters(device) -> torch.Tensor:
    """The chain's barrier counters: zero before the first launch, left zero by every launch (allocate ONCE, outside stream capture)."""
    return torch.zeros(_fn["lap_serve_chain_counter_words"](), dtype=torch.int32, device=device)


def serve_chain_failed(counters: torch.Tensor) -> bool:
    """True if a block of some launch gave up waiting at a grid barrier (that launch's results are invalid).  Synchronises."""
    st = C.c_int(0)
    _chk(_fn["lap_serve_chain_status"](_p(counters), C.byref(st)), "lap_serve_chain_status")
    return bool(st.value)


PACK_QKV, PACK_GATE_UP, PACK_PLAIN = 1, 2, 3     # lap_serve_pack_weight kinds (= the skinny kernels' epilogue ids)


def serve_pack_weight(w, kind: int, HD: int = 0, out=None):
    """Fragment-packed image of a K-contiguous bf16 weight [N, K] for the packed chain (include/lap_hip.h: lap_serve_pack_weight)."""
    _req(w, torch.bfloat16, "w")
    N, K = w.shape
    if not w.is_contiguous():
        raise ValueError("serve_pack_weight: w must be contiguous")
    if out is None:
        out = torch.empty_like(w)
    call("lap_serve_pack_weight", _p(w), _p(out), N, K, kind, HD)
    return out


def panel_gemm_ok(M, N, K, ksplit=1) -> bool:
    return bool(_fn["lap_panel_gemm_ok"](M, N, K, ksplit))


def _nbytes(t):
    return 0 if t is None else t.numel() * t.element_size()


def panel_linear(x, wp, N, *, bias=None, residual=None, norm=0, gamma=None, beta=None, eps=1e-6, gelu=False, out=None, nt=0, prefetch=None):
    """y[M, N] = epi(norm(x)[M, K] @ W[N, K]^T) on the row-panel kernel (csrc/serve_panel.hip); wp = serve_pack_weight(W, PACK_PLAIN).
    prefetch: the weight tensor of the NEXT launch of the chain (pulled into the Infinity Cache beside this product)."""
    M, K = x.shape
    _req(x, torch.bfloat16, "x"); _req(wp, torch.bfloat16, "wp")
    if wp.numel() != N * K or not wp.is_contiguous():
        raise ValueError(f"panel_linear: wp must be the contiguous packed image of a [{N}, {K}] weight, got {tuple(wp.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    flags = 0
    if gelu:    # "bf16": the pre-activation rounded to bf16 first; "exp2": that, with the GELU through v_exp / v_rcp (the training kernels' form)
        flags |= GEMM_GELU | (GEMM_GELU_BF16 if gelu in ("bf16", "exp2") else 0) | (GEMM_GELU_EXP2 if gelu == "exp2" else 0)
    call("lap_panel_gemm_pf", _p(x), x.stride(0), _p(wp), _p(out), out.stride(0), _p(bias), _p(residual),
         residual.stride(0) if residual is not None else 0, norm, _p(gamma), _p(beta), float(eps), M, N, K, flags, 1, None, nt,
         _p(prefetch), _nbytes(prefetch))
    return out


def panel_partials(x, wp, N, scratch, ksplit, nt=0, prefetch=None):
    """Raw f32 partial products [ksplit, M, N] of x @ W^T on the row-panel kernel (the consumers: fused_reduce_norm,
    fused_reduce_rope_split); ksplit = 1: the whole product as one f32 slab."""
    M, K = x.shape
    _req(x, torch.bfloat16, "x"); _req(wp, torch.bfloat16, "wp")
    if wp.numel() != N * K or not wp.is_contiguous():
        raise ValueError(f"panel_partials: wp must be the contiguous packed image of a [{N}, {K}] weight, got {tuple(wp.shape)}")
    need = ksplit * M * N
    if scratch.numel() < need:
        raise ValueError("scratch too small for the requested split")
    call("lap_panel_gemm_pf", _p(x), x.stride(0), _p(wp), None, 0, None, None, 0, 0, None, None, 0.0, M, N, K, 0, ksplit, _p(scratch), nt,
         _p(prefetch), _nbytes(prefetch))
    return scratch[:need].view(ksplit, M, N), ksplit


def serve_chain_tp_ok(B, S, D, H, NH, HD, NKV, prefix_len) -> bool:
    """Shapes (and device) the tensor-parallel form of the packed chain takes (lap_serve_chain with packed = 2)."""
    return bool(_fn["lap_serve_chain_tp_ok"](B, S, D, H, NH, HD, NKV, prefix_len))


def serve_chain_scratch(device, D, H, NH, HD, tp: bool = False) -> dict:
    """The packed chain's activation buffers (64 rows each, zero: the pad rows are never written).  Allocate ONCE per sampler,
    outside stream capture; a launch may be replayed from a graph that holds these addresses.  tp: also the tensor-parallel
    form's per-XCD buffers and partial-sum slabs."""
    z = lambda cols: torch.zeros((64, cols), dtype=torch.bfloat16, device=device)
    d = dict(q=z(NH * HD), o=z(NH * HD), xa=z(D), act=z(H), xs=z(D))
    if tp:
        z8 = lambda cols: torch.zeros((8, 64, cols), dtype=torch.bfloat16, device=device)
        d.update(tp_slabs=torch.zeros((2, 8, 64, D), dtype=torch.float32, device=device), tp_xs=z8(D), tp_xn=z8(D), tp_k=z8(HD), tp_v=z8(HD))
    return d


def serve_chain(x_in, mod, slot_stride, weights, caches, rope_table, qinfo, kinfo, B, S, NH, HD, H, prefix_len, q_scale, counters,
                eps=1e-6, debug_clock=None, keep=None, packed_scratch=None, tp: bool = False):
    """All action-expert layers of one denoise step in one persistent launch (include/lap_hip.h: lap_serve_chain).
    weights: per layer (wqkv, wo, wgu, wd); caches: per layer (k, v) of the prefix.  Returns the residual stream [B*S, D].
    `packed_scratch` (serve_chain_scratch): the weights are lap_serve_pack_weight images and the activations between the stages are
    fragment-packed too (same bits, every operand load 1 KiB contiguous per wave instruction).  `tp` (with a scratch made with
    tp=True): the tensor-parallel form, lap_serve_chain packed = 2."""
    M, D = x_in.shape
    dev = x_in.device
    depth = len(weights)
    a = ServeChainArgs()
    a.depth, a.B, a.S, a.D, a.H, a.NH, a.HD, a.prefix_len = depth, B, S, D, H, NH, HD, prefix_len
    out = torch.empty_like(x_in)
    a.x_in, a.x_out, a.mod, a.mod_slot_stride = _p(x_in), _p(out), _p(mod), slot_stride
    for l, (wqkv, wo, wgu, wd) in enumerate(weights):
        a.wqkv[l], a.wo[l], a.wgu[l], a.wd[l] = _p(wqkv), _p(wo), _p(wgu), _p(wd)
        ck, cv = caches[l]
        a.cache_k[l], a.cache_v[l] = _p(ck), _p(cv)
    a.kv_rs = 0
    a.rope_table, a.qinfo, a.kinfo = _p(rope_table), _p(qinfo), _p(kinfo)
    a.q_scale, a.eps = float(q_scale), float(eps)
    bf = dict(dtype=torch.bfloat16, device=dev)
    k, v = torch.empty((M, HD), **bf), torch.empty((M, HD), **bf)
    if packed_scratch is not None:
        ps = packed_scratch
        q, o, xa, act = ps["q"], ps["o"], ps["xa"], ps["act"]
        a.packed, a.xs = 1, _p(ps["xs"])
        if tp:
            a.packed = 2
            a.tp_slabs, a.tp_xs, a.tp_xn, a.tp_k, a.tp_v = (_p(ps[n]) for n in ("tp_slabs", "tp_xs", "tp_xn", "tp_k", "tp_v"))
    else:
        q, o = torch.empty((M, NH * HD), **bf), torch.empty((M, NH * HD), **bf)
        xa, act = torch.empty((M, D), **bf), torch.empty((M, H), **bf)
        a.packed, a.xs = 0, None
    ns = _fn["lap_attention_serve_splits"](prefix_len, S, B, NH, S)
    scratch = torch.empty(ns * M * NH * (HD + 1), dtype=torch.float32, device=dev)
    a.q, a.k, a.v, a.o, a.xa, a.act = _p(q), _p(k), _p(v), _p(o), _p(xa), _p(act)
    a.attn_scratch, a.attn_scratch_floats, a.counters = _p(scratch), scratch.numel(), _p(counters)
    a.debug_clock = _p(debug_clock)
    if keep is not None:      # (tests / debugging: the last layer's intermediates)
        keep.update(q=q, k=k, v=v, o=o, xa=xa, act=act)
    _chk(_fn["lap_serve_chain"](C.byref(a), _stream()), "lap_serve_chain")
    return out


def serve_embed_actions(x_t, w_in, b_in):
    rows, ad = x_t.shape
    D = w_in.shape[0]
    tok = torch.empty((rows, D), dtype=torch.bfloat16, device=x_t.device)
    call("lap_serve_embed_actions", _p(x_t), _p(w_in), _p(b_in), _p(tok), rows, ad, D)
    return tok


def serve_final_euler(x, mod, mod_ld, rps, w_out, b_out, x_t, dt, v_out=None, eps=1e-6):
    rows, D = x.shape
    call("lap_serve_final_euler", _p(x), _p(mod), mod_ld, rps, _p(w_out), _p(b_out), _p(x_t), _p(v_out), rows, D, w_out.shape[0], float(dt), float(eps))


def serve_final_euler_embed(x, mod, mod_ld, rps, w_out, b_out, x_t, dt, v_out=None, eps=1e-6, w_in=None, b_in=None, tokens=None):
    """serve_final_euler (action_dim 32) and, with `tokens` (a [rows, D] bf16 tensor to fill), the next step's serve_embed_actions in one launch."""
    rows, D = x.shape
    call("lap_serve_final_euler_embed", _p(x), _p(mod), mod_ld, rps, _p(w_out), _p(b_out), _p(x_t), _p(v_out), rows, D, w_out.shape[0], float(dt), float(eps),
         _p(w_in), _p(b_in), _p(tokens))


def serve_set_variant(feature_tiles: int):
    """Tuning knob of lap_serve_proj_residual (tools/bench_skinny.py)."""
    _chk(_fn["lap_serve_set_variant"](int(feature_tiles)), "lap_serve_set_variant")


# ------------------------------------------------------------------------- fp8 GEMM path (csrc/gemm_fp8.hip, config 5)
def quantize_fp8(x):
    """bf16 [rows, cols] (row stride x.stride(0)) -> (uint8 [rows, cols] e4m3 bytes, f32 scale [1]); per-tensor current scaling."""
    _req(x, torch.bfloat16, "x")
    rows, cols = x.shape
    amax = torch.zeros(1, dtype=torch.float32, device=x.device)
    call("lap_amax_bf16", _p(x), rows, cols, x.stride(0), _p(amax))
    out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    call("lap_quantize_fp8", _p(x), rows, cols, x.stride(0), _p(amax), _p(out), cols, _p(scale))
    return out, scale


def quantize_fp8_weight(wt):
    """bf16 Wt [out, in] contiguous -> (W8 [out, in], W8t [in, out], scale [1])."""
    _req(wt, torch.bfloat16, "wt")
    if not wt.is_contiguous():
        raise ValueError("quantize_fp8_weight needs a contiguous weight")
    rows, cols = wt.shape
    amax = torch.zeros(1, dtype=torch.float32, device=wt.device)
    call("lap_amax_bf16", _p(wt), rows, cols, cols, _p(amax))
    w8 = torch.empty((rows, cols), dtype=torch.uint8, device=wt.device)
    w8t = torch.empty((cols, rows), dtype=torch.uint8, device=wt.device)
    scale = torch.empty(1, dtype=torch.float32, device=wt.device)
    call("lap_quantize_fp8_weight", _p(wt), rows, cols, _p(amax), _p(w8), _p(w8t), _p(scale))
    return w8, w8t, scale


def gemm_fp8(a8, sa, b8, sb, out=None, *, residual=None, out_dtype=torch.bfloat16, accum=False, alpha=1.0):
    """out[M, N] = alpha / (sa * sb) * a8[M, K] @ b8[N, K]^T (+ residual): fp8 e4m3 operands, f32 accumulate."""
    M, K = a8.shape
    N = b8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    flags = (GEMM_OUT_F32 if out.dtype == torch.float32 else 0) | (GEMM_ACCUM if accum else 0)
    call("lap_gemm_fp8", _p(a8), _p(b8), _p(out), _p(residual), _p(sa), _p(sb), M, N, K, a8.stride(0), b8.stride(0), out.stride(0),
         residual.stride(0) if residual is not None else 0, float(alpha), flags)
    return out


def stream_with_hip_priority(device, level: str):
    """A HIP stream of the LOWEST ("low") or HIGHEST ("high") priority the device offers, wrapped for torch (torch itself only
    creates streams of normal or higher priority).  hipStreamNonBlocking, like torch's own pool streams.  The wrapper keeps no
    ownership: the stream lives for the rest of the process (one per pipeline object)."""
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    if rt.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        raise LapHipError("hipDeviceGetStreamPriorityRange failed")
    handle = ctypes.c_void_p()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        rc = rt.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(least.value if level == "low" else greatest.value))
    if rc != 0 or not handle.value:
        raise LapHipError(f"hipStreamCreateWithPriority failed: {rc}")
    return torch.cuda.ExternalStream(handle.value, device=dev)
