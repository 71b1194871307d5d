"""LAP on MI355X: the model object behind `LAPConfig.create()` / `.load()`.

Keeps the reference's model surface (src/lap/models/lap.py):
    compute_loss(rng, observation, actions, *, train=False, ...) -> (loss, metrics)      lap.py:380-602
    sample_actions(rng, observation, *, num_steps=10, noise=None) -> [b, ah, ad]         lap.py:605-675
and adds `loss_and_grad(...)`, the fused forward + hand-written backward that the train step uses (the
reference gets it from nnx.value_and_grad, scripts/train.py:358-361).

Everything numeric is a call into liblap_hip.so (lap_amd/hip.py); torch only owns device memory, the stream,
and a few O(batch x tokens) integer tensors (masks -> per-token info words, positions).  There is no autograd
and no fallback path.  Activations needed by the backward are kept in HBM (288 GB per MI355X) instead of being
recomputed — the reference rematerialises every block (gemma.py:418-423 nothing_saveable), which costs a
fourth forward pass.

Joint two-expert transformer (gemma.py:455-531): the prefix stream (SigLIP tokens + prompt, width of the VLM)
and the suffix stream (action tokens, width of the action expert) keep separate activations and weights and
meet only inside the attention kernel, which takes both as segments.  In the train step the suffix stream's kernels (1,600 rows:
poorly filled grids, launch-latency bound) are issued on a second HIP stream and run under the prefix stream's GEMMs; the two
streams synchronise before and after each layer's attention (`LAP_DUAL_STREAM=0`: everything on one stream).
"""
from __future__ import annotations

import contextlib
import dataclasses
import math
import os

import torch

from lap_amd import hip
from lap_amd.config import LAPConfig, get_gemma_config, get_siglip_config
from lap_amd.observation import CoTObservation, preprocess_observation
from lap_amd.params import ParamStore

SUFFIX_IDX_BASE = 0x800000  # suffix ar-indices live above every prefix index (see _train_infos)


class _NullComm:
    """world_size == 1: parameters are always resident, gradients stay where they are written."""
    world_size = 1

    def wait_unit(self, name, also=None): pass
    def pace(self, name): pass
    def grads_ready(self, name, also=None): pass
    def before_backward(self): pass
    def all_reduce_sum(self, t): return t


def _gen(rng, device):
    if isinstance(rng, torch.Generator):
        return rng
    g = torch.Generator(device=device)
    g.manual_seed(int(rng) if rng is not None else 0)
    return g


class LAP:
    EOS_TOKEN = 1

    def __init__(self, config: LAPConfig, seed: int = 0, params: dict | None = None, device="cuda", store: ParamStore | None = None,
                 comm=None, with_optimizer: bool = False, with_ema: bool = False, with_grads: bool = True, gemm_dtype: str = "bf16"):
        self.config = config
        if gemm_dtype not in ("bf16", "fp8"):
            raise ValueError(f"gemm_dtype {gemm_dtype!r}: 'bf16' or 'fp8'")
        self.gemm_dtype = gemm_dtype
        self._w8: dict = {}     # fp8 mirrors of the VLM projections: name -> (store version, W8, W8t, scale)
        self.device = torch.device(device)
        self.v = get_gemma_config(config.paligemma_variant)
        self.e = get_gemma_config(config.action_expert_variant)
        self.s = get_siglip_config(config.siglip_variant)
        self.action_dim, self.action_horizon, self.max_token_len = config.action_dim, config.action_horizon, config.max_token_len
        self.comm = comm if comm is not None else _NullComm()
        if store is None:
            store = ParamStore(config, device, with_optimizer=with_optimizer, with_ema=with_ema, with_grads=with_grads)
            if params is not None:
                store.load_reference_tree(params)
            else:
                store.init_random(seed)
        self.ps = store
        self.n_img_tok = (config.image_size // self.s.patch) ** 2
        self.deterministic = True
        self.dual_stream = os.environ.get("LAP_DUAL_STREAM", "1") != "0"
        # serving prefill on the fused consumers (`_siglip_fwd_serve`, `_llm_prefill`); "0": the generic layer loops (A/B, tests)
        self.serve_fusions = os.environ.get("LAP_SERVE_FUSIONS", "1") != "0"
        # prefix stream, bf16: d(act) of the down projection goes straight into the GeGLU backward inside the assembly GEMM's epilogue
        self.fuse_geglu_bwd = os.environ.get("LAP_FUSE_GEGLU_BWD", "1") != "0" and gemm_dtype != "fp8"
        self.fuse_geglu_fwd = os.environ.get("LAP_FUSE_GEGLU_FWD", "1") != "0" and gemm_dtype != "fp8"
        self.fuse_gelu = os.environ.get("LAP_FUSE_GELU", "1") != "0"      # SigLIP MLP: GELU forward / backward inside the Dense GEMMs
        # first denoise step on a second stream beside the prefill (it needs layer l's K / V only at its layer l).  Measured, hipGraph
        # replay, same box, interleaved: 15.65 -> 16.30 ms per chunk — the step's 110 short kernels take CUs from the prefill's
        # load-bound GEMMs for longer than they save.  Kept as a switch, OFF by default.
        self.serve_overlap = os.environ.get("LAP_SERVE_OVERLAP", "0") != "0"
        # the 18 action-expert layers of a denoise step as ONE persistent launch (csrc/serve_chain.hpp) instead of 6 launches per
        # layer; bitwise equal to them.  LAP_SERVE_CHAIN=0: the separate launches (A/B runs, tests).
        self.serve_chain = os.environ.get("LAP_SERVE_CHAIN", "1") != "0"
        # ... on fragment-packed operands (round 4: every operand load 1 KiB contiguous per wave instruction; same bits).
        # LAP_SERVE_PACKED=0: the row-major chain of round 3 (A/B runs, tests)
        self.serve_packed = os.environ.get("LAP_SERVE_PACKED", "1") != "0"
        # ... mapped onto the chip as 8-way tensor parallelism over the XCDs (csrc/serve_chain_tp.hpp: two chip-wide seams per layer
        # instead of five; bf16-rounding-noise equal to the flat chain, not bitwise).  LAP_SERVE_TP=0: the flat packed chain
        self.serve_tp = os.environ.get("LAP_SERVE_TP", "0") != "0"
        # an Euler step's tail (final adaRMS + action_out_proj + x_t update) and the next step's action_in_proj in one launch
        # (csrc/serve_skinny.hip final_euler_embed_kernel; same arithmetic).  LAP_SERVE_EULER_EMBED=0: the two launches
        self.serve_euler_embed = os.environ.get("LAP_SERVE_EULER_EMBED", "1") != "0"
        # the serving prefill's small projections (SigLIP qkv / out / fc1 / fc2; Gemma's qkv / out opt-in, see _panel_llm) on the row-panel kernel
        # (csrc/serve_panel.hip: the rows of A resident in LDS, packed weights streamed into MFMA fragments, no barrier in the k-loop)
        # against packed weight images kept per parameter version (+1.0 GB for LAP-3B).  qkv / out / fc1 are bitwise equal to the
        # unsplit tiles they replace.  LAP_SERVE_PANEL=0: the LDS-tiled kernels (A/B runs, tests)
        self.serve_panel = os.environ.get("LAP_SERVE_PANEL", "1") != "0"
        self._prefill_pw: dict = {}     # name -> [version, packed image]
        # the prefill's fused GELU / GeGLU epilogues (SigLIP fc1 on the panel kernel, Gemma's gate|up tile) through v_exp / v_rcp, the training
        # kernels' arithmetic; "bf16": tanhf (A/B)
        self._panel_gelu = os.environ.get("LAP_SERVE_PANEL_GELU", "exp2")
        # which Gemma prefill projections take the panel kernel (q, o): none by default — in the chunk the out projection takes 23 us there
        # against 19.6 on the LDS tile (14.6 on cache-warm weights), qkv as one f32 slab 25 against the split-K tile's 17 (12.06 / 12.14 / 12.22 ms)
        self._panel_llm = os.environ.get("LAP_SERVE_PANEL_LLM", "")
        # feature tiles of 16 per wave for SigLIP's qkv / out / fc1 / fc2 on the panel kernel (tuning knob; sweep in the chunk: docs/EXPERIMENTS.md K)
        self._panel_nt = tuple(int(v) for v in os.environ.get("LAP_SERVE_PANEL_NT", "2,1,3,3").split(","))
        # ... and every panel launch pulls the NEXT launch's weights into the Infinity Cache with a fifth wave per block (the chain's
        # launches otherwise meet their weights HBM-cold).  LAP_SERVE_PREFETCH=0: off (A/B runs)
        self.serve_prefetch = os.environ.get("LAP_SERVE_PREFETCH", "1") != "0"
        self._chain_ctr = None
        self._chain_scratch = None
        self._packed_w = None       # [version, per layer (wqkv, wo, wgu, wd) packed images]
        self._den = None
        # K splits of the prefill's qkv / out / down projections, down's tile.  Round 5 sweep incl. the consumer (tools/probes/
        # prefill_down_sweep.py, us per GEMM + reduce / residual / norm): 256 x 256 x 8 65.0 | 320 x 128 (tile 19) x 8 60.8 | 320 x 256 x 16 64.5
        ks = os.environ.get("LAP_PREFILL_KS", "4,1,8,19").split(",")
        self._prefill_ks = tuple(int(k) for k in ks)
        self._sfx = None        # the suffix stream's HIP stream (created on first use)
        # which gradients leave the data-gradient path for a third stream (see _off_path): s SigLIP weights, b biases, q / g the
        # prefix stream's attention / MLP projections; "1" all, "0" none.  Measured (tools/ab3.sh, interleaved on one box): sb
        # -2.1 .. -3.3 ms per step in round 2 and -2.6 ms in round 3 (310.0 -> 307.4), s -1.3, q 0, all +12.6.  Default "sb":
        # step time is the decision variable (the compute stream's GEMMs then share the chip, so their EVENT-timed rate drops
        # by 4 % — bench.py's roofline figure is computed from isolated per-shape times for that reason).
        mode = os.environ.get("LAP_WGRAD_STREAM", "sb")
        self.wgrad_stream = "" if mode == "0" else mode
        self._wg = self._wg_obj = self._wg_main = None
        self._wg_dirty = False
        self._wg_ev: dict = {}
        if gemm_dtype == "fp8":
            dims = (self.v.width, self.v.num_heads * self.v.head_dim, self.v.mlp_dim, (self.v.num_heads + 2 * self.v.num_kv_heads) * self.v.head_dim)
            if any(d % 128 for d in dims):
                raise ValueError(f"gemm_dtype='fp8' needs projection dimensions that are multiples of 128, got {dims}")

    # ------------------------------------------------------------------ small helpers
    def _suffix_stream(self, *tensors):
        """The second HIP stream for the suffix (action-expert) side of a joint layer loop, or None when both streams of
        activations go down the current one (one of them absent, CPU tensors, stream capture, LAP_DUAL_STREAM=0).  It starts
        behind everything the current stream has been given so far; `tensors` are marked as used on it."""
        if not self.dual_stream or any(t is None or not t.is_cuda for t in tensors) or torch.cuda.is_current_stream_capturing():
            return None
        if self._sfx is None:
            self._sfx = torch.cuda.Stream(self.device, priority=-1)   # short kernels: never let them queue behind a full grid
        self._sfx.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            t.record_stream(self._sfx)
        return self._sfx

    @staticmethod
    def _handoff(src, dst, *tensors):
        """`dst` waits for what `src` has been given so far; `tensors` (allocated on src) are about to be used on dst."""
        if src is not None and dst is not None:
            dst.wait_stream(src)
            for t in tensors:
                if t is not None:
                    t.record_stream(dst)

    def W(self, name):
        return self.ps.w16(name)

    def F(self, name):
        return self.ps.f32(name)

    def G(self, name):
        return self.ps.g(name)

    def _wgrad(self, dy, x, name, **kw):
        """Weight gradient dWt = dy^T x into the gradient buffer of `name` — skipped for frozen parameters
        (scripts/train.py:358-361 differentiates w.r.t. the trainable filter only)."""
        if self.ps.is_trainable(name):
            mode = self.wgrad_stream
            kind = "s" if name.startswith("img/") else ("q" if name.endswith(("wqkv0", "wo0")) else "g")
            fold = not kw and getattr(self.comm, "fold_sumsq", False) and dy.dtype == torch.bfloat16

            def run():
                if fold:    # the weight gradient and, where the assembly kernel takes it, its share of the gradient norm in one launch
                    if hip.linear_wgrad_sumsq(dy, x, self.G(name), self.comm.sumsq[0:1]):
                        self.comm.folded.add(name)
                else:
                    hip.linear_wgrad(dy, x, self.G(name), **kw)
            if mode == "1" or kind in mode:
                with self._off_path(dy, x):
                    run()
            else:
                run()

    def _bgrad(self, dy, name):
        """Bias gradient = column sums of dy, next to the weight gradient."""
        mode = self.wgrad_stream
        if mode == "1" or "b" in mode:
            with self._off_path(dy):
                hip.colsum(dy, self.G(name))
        else:
            hip.colsum(dy, self.G(name))

    # Nothing in the backward waits for a weight / bias gradient except the optimizer, so the prefix stream's and SigLIP's are
    # issued on a third HIP stream: the data-gradient chain (the critical path) keeps the compute stream, and the weight-gradient
    # GEMMs fill the CUs its poorly filled last rounds leave idle.  The compute stream joins before it updates a dy in place and
    # before a unit's gradients are declared complete.  (LAP_WGRAD_STREAM=0: inline.)
    @contextlib.contextmanager
    def _off_path(self, *tensors):
        wg = self._wg
        cur = torch.cuda.current_stream()
        if wg is None or cur != self._wg_main:
            yield
            return
        wg.wait_stream(self._wg_main)
        for t in tensors:
            t.record_stream(wg)
        with torch.cuda.stream(wg):
            yield
        ev = torch.cuda.Event()
        ev.record(wg)
        self._wg_ev.setdefault(tensors[0].data_ptr(), []).append(ev)     # keyed by dy: the tensor the path may rewrite
        self._wg_dirty = True

    def _wg_join(self, dy=None):
        """The compute stream waits for the off-path readers of `dy` (about to be updated in place), or for all of them."""
        if self._wg is None or not self._wg_dirty or torch.cuda.current_stream() != self._wg_main:
            return
        if dy is not None:
            for ev in self._wg_ev.pop(dy.data_ptr(), ()):
                self._wg_main.wait_event(ev)
        else:
            self._wg_main.wait_stream(self._wg)
            self._wg_ev.clear()
            self._wg_dirty = False

    def _unit_done(self, name, sfx=None):
        """comm.grads_ready for a unit whose gradients may still be in flight on the second (`sfx`) or third stream: the
        communication / optimizer stream waits for those too, the compute stream does not."""
        also = [sfx] if sfx is not None else []
        if self._wg is not None and self._wg_dirty and torch.cuda.current_stream() == self._wg_main:
            also.append(self._wg)
        self.comm.grads_ready(name, also=also or None)

    def _wg_begin(self):
        """Start of a backward pass on the current stream: weight gradients go off the path from here on."""
        if self.wgrad_stream and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            if self._wg_obj is None:
                self._wg_obj = torch.cuda.Stream(self.device)
            self._wg, self._wg_main, self._wg_dirty = self._wg_obj, torch.cuda.current_stream(), False

    def _wg_end(self):
        self._wg_join()
        self._wg = None

    # ---- fp8 routing of the VLM expert's projections (BASELINE.json config 5; csrc/gemm_fp8.hip)
    def _w8_of(self, name):
        """(W8 [out][in], W8t [in][out], scale) of a VLM projection, re-quantised when the parameters changed."""
        ent = self._w8.get(name)
        if ent is None or ent[0] != self.ps.version:
            ent = (self.ps.version, *hip.quantize_fp8_weight(self.W(name)))
            self._w8[name] = ent
        return ent[1:]

    def _lin0(self, x, name, residual=None, out=None):
        """y = x @ Wt^T (+ residual) for a prefix-stream projection: bf16 MFMA GEMM, or e4m3 x e4m3 when gemm_dtype == 'fp8'."""
        if self.gemm_dtype == "fp8":
            x8, sx = hip.quantize_fp8(x)
            w8, _, sw = self._w8_of(name)
            return hip.gemm_fp8(x8, sx, w8, sw, residual=residual)
        if residual is not None and os.environ.get("LAP_UNFUSED_RESIDUAL", "0") == "1":
            # measurement switch: the reference rounds the projection to bf16 and then the sum to bf16
            # (gemma.py:285,582-583); the fused epilogue adds in f32 and rounds once
            return hip.add_bf16(residual, hip.linear_fwd(x, self.W(name)))
        return hip.linear_fwd(x, self.W(name), out, residual=residual)

    def _dgrad0(self, dy, name):
        """dx = dy @ Wt for a prefix-stream projection (the fp8 route multiplies by the transposed fp8 copy)."""
        if self.gemm_dtype == "fp8":
            d8, sd = hip.quantize_fp8(dy)
            _, w8t, sw = self._w8_of(name)
            return hip.gemm_fp8(d8, sd, w8t, sw)
        return hip.linear_dgrad(dy, self.W(name))

    def _prefix_frozen(self) -> bool:
        """True when no parameter reached by the prefix stream's backward is trainable (e.g. `get_vlm_freeze_filter`):
        the language-head, VLM and SigLIP backward passes are then skipped altogether."""
        fr = self.ps.frozen
        if not fr:
            return False
        return all(v for k, v in fr.items() if k.startswith("img/") or k in ("llm/embed", "llm/final_norm")
                   or (k.startswith("llm/") and (k.endswith("0") or k.endswith("n_attn") or k.endswith("n_ffw"))))

    def _lin32(self, x, wname, bname):
        """nnx.Linear in f32: y = x @ W^T + b (W stored [out][in])."""
        w = self.F(wname)
        out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        return hip.gemm_f32(x, w, out, M=x.shape[0], N=w.shape[0], K=w.shape[1], lda=x.stride(0), ldb=w.stride(0), ldc=w.shape[0],
                            bias=self.F(bname))

    def _lin32_bwd(self, x, dy, wname, bname, need_dx=True):
        """Gradients of _lin32: dW += dy^T x, db += colsum(dy), returns dx = dy @ W."""
        w = self.F(wname)
        hip.gemm_f32(dy, x, self.G(wname), M=w.shape[0], N=w.shape[1], K=x.shape[0], lda=dy.stride(0), ldb=x.stride(0), ldc=w.shape[1],
                     a_kc=False, b_kc=False, accum=True)
        hip.colsum(dy, self.G(bname))
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        return hip.gemm_f32(dy, w, dx, M=x.shape[0], N=w.shape[1], K=w.shape[0], lda=dy.stride(0), ldb=w.stride(0), ldc=w.shape[1],
                            a_kc=True, b_kc=False)

    # ================================================================== SigLIP
    def _siglip_fwd(self, images: torch.Tensor, save: bool, collect=None, x_in=None, blocks=None):
        """images f32 [N,H,W,3] -> tokens bf16 [N*T, Dv].  openpi siglip (missing) restated in
        siglip_gemma3.py:382-545 minus :432, plus head bias.
        Test hook (teacher-forced per-block parity): with `x_in` (bf16 [N*T, W]) the stem is skipped, only `blocks` run
        and the block output is returned instead of the projected tokens."""
        s, T = self.s, self.n_img_tok
        W = s.width
        hd = W // s.num_heads
        ctx = {"blocks": []} if save else None
        if x_in is not None:
            x, N = x_in, x_in.shape[0] // T
            for l in blocks:
                x = self._siglip_block(l, x, N, T, W, hd, None, False)
            return x, None
        N = images.shape[0]
        patches = hip.im2col_patch(images.contiguous(), s.patch)
        # f32 stem (siglip_gemma3.py:398-408) on the MFMA path: x = hi + lo (2 x bf16, 16 mantissa bits), products exact
        # in the f32 accumulator, the lo.lo term (2^-18 relative) dropped
        p_hi, p_lo = hip.split_f32_hilo(patches)
        w_hi, w_lo = hip.split_f32_hilo(self.F("img/stem_w"))
        del patches
        stem = torch.empty((p_hi.shape[0], W), dtype=torch.float32, device=p_hi.device)
        R, Kp = p_hi.shape
        hip.gemm(p_hi, w_hi, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, bias=self.F("img/stem_b"))
        hip.gemm(p_hi, w_lo, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, accum=True)
        hip.gemm(p_lo, w_hi, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, accum=True)
        x = hip.add_posemb_cast(stem, self.F("img/pos"), T)
        del stem
        if collect is not None:
            collect["img/stem"] = x
        if save:
            ctx["patches"] = (p_hi, p_lo)
        for l in range(s.depth):
            x = self._siglip_block(l, x, N, T, W, hd, ctx, save)
            if collect is not None:
                collect[f"img/block{l:02d}"] = x
        self.comm.wait_unit("img_head")
        enc, mean, rstd = hip.layernorm_fwd(x, self.F("img/norm_g"), self.F("img/norm_b"))
        tok = hip.linear_fwd(enc, self.W("img/head_w"), bias=self.F("img/head_b"))
        if save:
            ctx["final"] = (x, enc, mean, rstd)
        if collect is not None:
            collect["img/out"] = tok
        return tok, ctx

    def _siglip_fwd_serve(self, images: torch.Tensor):
        """The tower of `_siglip_fwd` for the serving prefill (nothing is kept for a backward): same operations and rounding
        points, fewer launches — GELU in fc1's epilogue (after the bf16 rounding of the Dense output), and fc2's split-K reduce,
        bias, residual add and the NEXT LayerNorm in one consumer pass (`lap_fused_reduce_norm`).  12 -> 10 launches per block,
        on block tiles sized for 512 rows (lap_gemm_bf16_ex serving rule)."""
        s, T = self.s, self.n_img_tok
        W = s.width
        hd = W // s.num_heads
        N = images.shape[0]
        patches = hip.im2col_patch(images.contiguous(), s.patch)
        p_hi, p_lo = hip.split_f32_hilo(patches)
        w_hi, w_lo = hip.split_f32_hilo(self.F("img/stem_w"))
        stem = torch.empty((p_hi.shape[0], W), dtype=torch.float32, device=p_hi.device)
        R, Kp = p_hi.shape
        hip.gemm(p_hi, w_hi, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, bias=self.F("img/stem_b"))
        hip.gemm(p_hi, w_lo, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, accum=True)
        hip.gemm(p_lo, w_hi, stem, M=R, N=W, K=Kp, lda=Kp, ldb=Kp, ldc=W, accum=True)
        x = hip.add_posemb_cast(stem, self.F("img/pos"), T)
        scratch = hip._gemm_scratch(self.device)
        self.comm.wait_unit("img0")
        y, _, _ = hip.layernorm_fwd(x, self.F("img/0/ln1_g"), self.F("img/0/ln1_b"))
        rows = x.shape[0]
        mlp = self.W("img/0/w1").shape[0]
        panel = (self.serve_panel and rows <= 640 and hip.panel_gemm_ok(rows, 3 * W, W) and hip.panel_gemm_ok(rows, mlp, W)
                 and hip.panel_gemm_ok(rows, W, mlp, 4))
        pf = (lambda name: self._pw(name)) if panel and self.serve_prefetch else (lambda name: None)   # the next launch's weights
        for l in range(s.depth):
            p = f"img/{l}/"
            if panel:    # us per launch at 512 rows, replayed graph (tools/probes/panel_bench.py): 11.6 -> 11.0, 9.0 -> 8.0, 14.5 -> 12.8, 14.9 -> 12.0
                qkv = hip.panel_linear(y, self._pw(p + "wqkv"), 3 * W, bias=self.F(p + "bqkv"), nt=self._panel_nt[0], prefetch=pf(p + "wo"))
            else:
                qkv = hip.linear_fwd(y, self.W(p + "wqkv"), bias=self.F(p + "bqkv"))
            (o, _), _ = hip.attention_fwd([qkv[:, :W]], [qkv[:, W:2 * W]], [qkv[:, 2 * W:]], [T], [T], N, s.num_heads, s.num_heads, hd,
                                          scale=hd ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0), need_lse=False)
            if panel:
                x1 = hip.panel_linear(o, self._pw(p + "wo"), W, bias=self.F(p + "bo"), residual=x, nt=self._panel_nt[1], prefetch=pf(p + "w1"))
            else:
                x1 = hip.linear_fwd(o, self.W(p + "wo"), bias=self.F(p + "bo"), residual=x)
            y2, _, _ = hip.layernorm_fwd(x1, self.F(p + "ln2_g"), self.F(p + "ln2_b"))
            if panel:
                a = hip.panel_linear(y2, self._pw(p + "w1"), mlp, bias=self.F(p + "b1"), gelu=self._panel_gelu, nt=self._panel_nt[2], prefetch=pf(p + "w2"))
                if l + 1 < s.depth:
                    self.comm.wait_unit(f"img{l + 1}")
                part, ks = hip.panel_partials(a, self._pw(p + "w2"), W, scratch, 4, nt=self._panel_nt[3], prefetch=pf(f"img/{l + 1}/wqkv") if l + 1 < s.depth else None)
            else:
                a = hip.linear_fwd(y2, self.W(p + "w1"), bias=self.F(p + "b1"), gelu="bf16")
                part, ks = hip.linear_partials(a, self.W(p + "w2"), scratch)
            if l + 1 < s.depth:
                self.comm.wait_unit(f"img{l + 1}")
                g, b = self.F(f"img/{l + 1}/ln1_g"), self.F(f"img/{l + 1}/ln1_b")
            else:
                self.comm.wait_unit("img_head")
                g, b = self.F("img/norm_g"), self.F("img/norm_b")
            x, y = hip.fused_reduce_norm(part, ks, x1.shape[0], W, bias=self.F(p + "b2"), residual=x1, norm=2, gamma=g, beta=b)
        return hip.linear_fwd(y, self.W("img/head_w"), bias=self.F("img/head_b"))

    def _siglip_block(self, l, x, N, T, W, hd, ctx, save):
        """One pre-LN encoder block (siglip_gemma3.py:59-167): x + MHA(LN(x)), then + MLP(LN(.))."""
        s = self.s
        self.comm.wait_unit(f"img{l}")
        p = f"img/{l}/"
        y, mean1, rstd1 = hip.layernorm_fwd(x, self.F(p + "ln1_g"), self.F(p + "ln1_b"))
        qkv = hip.linear_fwd(y, self.W(p + "wqkv"), bias=self.F(p + "bqkv"))
        (o, _), lse = hip.attention_fwd([qkv[:, :W]], [qkv[:, W:2 * W]], [qkv[:, 2 * W:]], [T], [T], N, s.num_heads, s.num_heads, hd,
                                        scale=hd ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0), need_lse=save)
        x1 = hip.linear_fwd(o, self.W(p + "wo"), bias=self.F(p + "bo"), residual=x)
        y2, mean2, rstd2 = hip.layernorm_fwd(x1, self.F(p + "ln2_g"), self.F(p + "ln2_b"))
        self.comm.pace(f"img{l}")
        if save:
            if self.fuse_gelu and hip.linear_bias_gelu_train_ok(y2, self.W(p + "w1"), self.F(p + "b1")):
                h, a = hip.linear_bias_gelu_train(y2, self.W(p + "w1"), self.F(p + "b1"))     # fc1 + bias with the GELU in its epilogue
            else:
                h = hip.linear_fwd(y2, self.W(p + "w1"), bias=self.F(p + "b1"))
                a = hip.gelu_fwd(h)
        else:   # nothing keeps the pre-activation: GELU in the GEMM epilogue, after the bf16 rounding of the Dense output (same bits)
            h, a = None, hip.linear_fwd(y2, self.W(p + "w1"), bias=self.F(p + "b1"), gelu="bf16")
        x2 = hip.linear_fwd(a, self.W(p + "w2"), bias=self.F(p + "b2"), residual=x1)
        if save:
            ctx["blocks"].append((x, y, mean1, rstd1, qkv, o, lse, x1, y2, mean2, rstd2, h, a))
        return x2

    def _siglip_bwd(self, ctx, dtok: torch.Tensor):
        s, T = self.s, self.n_img_tok
        W = s.width
        hd = W // s.num_heads
        x, enc, mean, rstd = ctx["final"]
        N = x.shape[0] // T
        self._bgrad(dtok, "img/head_b")
        self._wgrad(dtok, enc, "img/head_w")
        denc = hip.linear_dgrad(dtok, self.W("img/head_w"))
        self._unit_done("img_head")
        # Bias gradients = column sums of a dy.  Two of a block's four (fc2's and the out projection's) come out of the LayerNorm
        # backward that PRODUCES that dy instead of a pass of their own over 38 MB (+1.8 us in that kernel against a 15.5 us
        # column-sum launch; a GELU backward that sums its columns was 2.3 x slower than the two kernels it replaced).
        fuse_b = os.environ.get("LAP_FUSE_BGRAD", "1") != "0"
        last = s.depth - 1
        dx = hip.layernorm_bwd(x, denc, self.F("img/norm_g"), mean, rstd, self.G("img/norm_g"), self.G("img/norm_b"),
                               dxsum=self.G(f"img/{last}/b2") if fuse_b else None)
        for l in reversed(range(s.depth)):
            p = f"img/{l}/"
            x, y, mean1, rstd1, qkv, o, lse, x1, y2, mean2, rstd2, h, a = ctx["blocks"][l]
            if not fuse_b:
                self._bgrad(dx, p + "b2")
            self._wgrad(dx, a, p + "w2")
            if self.fuse_gelu and hip.dgrad_gelu_bwd_ok(dx, self.W(p + "w2"), h):
                dh = hip.linear_dgrad_gelu_bwd(dx, self.W(p + "w2"), h)      # fc2's data gradient with the GELU backward as its epilogue
            else:
                da = hip.linear_dgrad(dx, self.W(p + "w2"))
                dh = hip.gelu_bwd(h, da)
                del da
            self._bgrad(dh, p + "b1")
            self._wgrad(dh, y2, p + "w1")
            dy2 = hip.linear_dgrad(dh, self.W(p + "w1"))
            del dh
            self._wg_join(dx)    # (the fc2 weight / bias gradients read dx)
            hip.layernorm_bwd(x1, dy2, self.F(p + "ln2_g"), mean2, rstd2, self.G(p + "ln2_g"), self.G(p + "ln2_b"), dx=dx, accum_dx=True,
                              dxsum=self.G(p + "bo") if fuse_b else None)
            if not fuse_b:
                self._bgrad(dx, p + "bo")
            self._wgrad(dx, o, p + "wo")
            do = hip.linear_dgrad(dx, self.W(p + "wo"))
            dqkv = torch.empty_like(qkv)
            hip.attention_bwd([qkv[:, :W]], [qkv[:, W:2 * W]], [qkv[:, 2 * W:]], [o], [do], lse, [T], [T], N, s.num_heads, s.num_heads, hd,
                              scale=hd ** -0.5, q_rs=(3 * W, 0), kv_rs=(3 * W, 0),
                              dq_out=[dqkv[:, :W]], dk_out=[dqkv[:, W:2 * W]], dv_out=[dqkv[:, 2 * W:]])
            self._bgrad(dqkv, p + "bqkv")
            self._wgrad(dqkv, y, p + "wqkv")
            dy = hip.linear_dgrad(dqkv, self.W(p + "wqkv"))
            self._wg_join(dx)    # (the out-projection's read dx)
            hip.layernorm_bwd(x, dy, self.F(p + "ln1_g"), mean1, rstd1, self.G(p + "ln1_g"), self.G(p + "ln1_b"), dx=dx, accum_dx=True,
                              dxsum=self.G(f"img/{l - 1}/b2") if (fuse_b and l > 0) else None)
            ctx["blocks"][l] = None
            self._unit_done(f"img{l}")
        dstem = hip.add_posemb_cast_bwd(dx, self.G("img/pos"), T)     # f32 copy of a bf16 gradient: exact in bf16
        p_hi, p_lo = ctx["patches"]
        gw = self.G("img/stem_w")
        tmp = torch.empty((gw.shape[0], p_hi.shape[1]), dtype=torch.float32, device=gw.device)   # [W, 592]
        hip.linear_wgrad(dx, p_hi, tmp)
        hip.linear_wgrad(dx, p_lo, tmp, accum=True)
        gw.add_(tmp[:, :gw.shape[1]])
        hip.colsum(dstem, self.G("img/stem_b"))

    # ================================================================== token info words / positions
    def _prefix_masks(self, obs: CoTObservation):
        """lap.py:118-170: input mask and ar mask over [image tokens ..., prompt tokens]."""
        B = obs.tokenized_prompt.shape[0]
        T = self.n_img_tok
        im = [obs.image_masks[k][:, None].expand(B, T) for k in self.config.image_keys]
        prefix_mask = torch.cat(im + [obs.tokenized_prompt_mask], 1)
        zeros = torch.zeros(B, T * len(im), dtype=torch.bool, device=prefix_mask.device)
        la = obs.tokenized_langact_mask if obs.tokenized_langact_mask is not None else torch.zeros_like(obs.tokenized_prompt_mask)
        return prefix_mask, torch.cat([zeros, la], 1)

    def _suffix_idx(self, B, S, dev):
        """Block indices of the suffix tokens ([B, S] for pi05; [B, S + 1] with pi0's state token in front, one block earlier)."""
        idx = torch.full((B, S), SUFFIX_IDX_BASE + 1, dtype=torch.int32, device=dev)
        if S and not self.config.pi05:
            idx = torch.cat([idx[:, :1], idx + 1], 1)
        return idx

    def _train_infos(self, obs: CoTObservation, S: int):
        """Per-token info words equivalent to _build_combined_attention_mask / make_attn_mask (lap.py:303-364):
        class bit 1: valid prefix token (seen by prefix queries);  bit 2: in prefix_mask_action (seen by action
        queries);  bit 4: suffix token.  Low 24 bits: cumulative ar index (suffix indices offset above all prefix
        ones).  Positions per lap.py:366-377."""
        prefix_mask, prefix_ar = self._prefix_masks(obs)
        B, Pn = prefix_mask.shape
        dev = prefix_mask.device
        pma = prefix_mask & ~prefix_ar if obs.tokenized_langact_mask is not None else prefix_mask  # lap.py:303-325
        cs = torch.cumsum(prefix_ar.to(torch.int32), 1)
        kcls = prefix_mask.to(torch.int32) | (pma.to(torch.int32) << 1)
        kinfo_p = (kcls << 24) | cs
        qinfo_p = (prefix_mask.to(torch.int32) << 24) | cs
        # suffix: mask all ones, ar = [1, 0, ...] (embed_suffix pi05) -> one block; pi0: a state token in front as a block of its
        # own (ar = [1, 1, 0, ...]): the action tokens see it, it does not see them
        s_idx = self._suffix_idx(B, S, dev)
        S = s_idx.shape[1]
        kinfo = torch.cat([kinfo_p, (4 << 24) | s_idx], 1).to(torch.int32).contiguous()  # cumsum promoted to int64
        qinfo = torch.cat([qinfo_p, (6 << 24) | s_idx], 1).to(torch.int32).contiguous()
        ppos = torch.cumsum(prefix_mask.to(torch.int32), 1) - 1
        spos = pma.sum(-1, keepdim=True).to(torch.int32) + torch.arange(S, dtype=torch.int32, device=dev)[None]
        pos = torch.cat([ppos, spos], 1).to(torch.int32).contiguous()
        return qinfo, kinfo, pos

    def _serve_infos(self, obs: CoTObservation, S: int):
        """sample_actions masks (lap.py:624-654): prefix attends per make_attn_mask(prefix_mask, prefix_ar); suffix
        queries see every valid prefix token and all suffix tokens; suffix positions follow sum(prefix_mask)."""
        keys = self.config.image_keys
        if (self.config.pi05 and self.serve_fusions and obs.tokenized_prompt_mask.is_cuda and len(keys) <= 4
                and all(obs.image_masks[k].dtype == torch.bool for k in keys) and obs.tokenized_prompt_mask.dtype == torch.bool):
            la = obs.tokenized_langact_mask
            return hip.serve_infos([obs.image_masks[k].contiguous() for k in keys], self.n_img_tok, obs.tokenized_prompt_mask.contiguous(),
                                   None if la is None else la.to(torch.bool).contiguous(), S, SUFFIX_IDX_BASE + 1)
        prefix_mask, prefix_ar = self._prefix_masks(obs)
        B, Pn = prefix_mask.shape
        dev = prefix_mask.device
        cs = torch.cumsum(prefix_ar.to(torch.int32), 1)
        pm = prefix_mask.to(torch.int32)
        kinfo_p = ((pm | (pm << 1)) << 24) | cs
        qinfo_p = (pm << 24) | cs
        s_idx = self._suffix_idx(B, S, dev)
        S = s_idx.shape[1]
        kinfo_s, qinfo_s = (4 << 24) | s_idx, (6 << 24) | s_idx
        ppos = (torch.cumsum(pm, 1) - 1).to(torch.int32).contiguous()
        spos = (pm.sum(-1, keepdim=True) + torch.arange(S, dtype=torch.int32, device=dev)[None]).to(torch.int32)
        i32 = lambda t: t.to(torch.int32).contiguous()
        return (i32(qinfo_p), i32(kinfo_p), ppos, i32(qinfo_s), i32(torch.cat([kinfo_p, kinfo_s], 1)), i32(torch.cat([ppos, spos], 1)))

    # ================================================================== embedding of the two streams
    def _embed_prefix(self, obs: CoTObservation, save: bool, collect=None, serve: bool = False):
        """lap.py:118-170 -> x0 bf16 [B*Pn, Dv] with rows (b, [img0 | img1 | prompt])."""
        cfg = self.config
        keys = cfg.image_keys
        B = obs.tokenized_prompt.shape[0]
        T, Lt, Dv = self.n_img_tok, obs.tokenized_prompt.shape[1], self.v.width
        Pn = T * len(keys) + Lt
        images = torch.cat([obs.images[k] for k in keys], 0)
        if save or collect is not None or not (serve and self.serve_fusions):
            tok, ictx = self._siglip_fwd(images, save, collect)
        else:
            tok, ictx = self._siglip_fwd_serve(images), None
        x0 = torch.empty((B * Pn, Dv), dtype=torch.bfloat16, device=self.device)
        for i in range(len(keys)):
            hip.copy_rows_bf16(tok[i * B * T:(i + 1) * B * T], x0, B * T, T, Dv, T, 0, Pn, i * T)
        self.comm.wait_unit("embed")
        tokens = obs.tokenized_prompt.to(torch.int32).contiguous()
        rows, lo, hi = self.ps.embed_rows()
        if self.comm.world_size == 1:
            hip.embed_gather(rows, tokens, x0, B * Lt, Lt, Dv, Pn, T * len(keys), math.sqrt(Dv), lo, hi)
        else:
            self.comm.sharded_embed_gather(rows, lo, hi, tokens, x0, Lt, Dv, Pn, T * len(keys), math.sqrt(Dv))
        return x0, Pn, (ictx, tokens)

    def _embed_prefix_bwd(self, pctx, dx0, B, Pn):
        ictx, tokens = pctx
        keys = self.config.image_keys
        T, Lt, Dv = self.n_img_tok, tokens.shape[1], self.v.width
        hip.embed_scatter_add(self.G("llm/embed"), tokens, dx0, B * Lt, Lt, Dv, Pn, T * len(keys), math.sqrt(Dv))
        self.comm.grads_ready("embed")
        dtok = torch.empty((len(keys) * B * T, Dv), dtype=torch.bfloat16, device=self.device)
        for i in range(len(keys)):
            hip.copy_rows_bf16(dx0, dtok[i * B * T:(i + 1) * B * T], B * T, T, Dv, Pn, i * T, T, 0)
        self._siglip_bwd(ictx, dtok)

    def _time_mod(self, time: torch.Tensor, save: bool):
        """[UPSTREAM-RECALL] openpi Pi0.embed_suffix (pi05), time branch: posemb_sincos -> time_mlp_in -> swish ->
        time_mlp_out -> swish = adaRMS condition; then all 2L+1 adaRMS Dense layers (gemma.py:128) as ONE GEMM against the
        adaRMS bank.  time f32 [n] -> mod bf16 [n, nslots*3*We]."""
        We = self.e.width
        temb = hip.posemb_sincos(time.contiguous(), We, 4e-3, 4.0)
        h1 = self._lin32(temb, "act/time_in_w", "act/time_in_b")
        s1 = hip.swish_fwd(h1)
        h2 = self._lin32(s1, "act/time_out_w", "act/time_out_b")
        cond = hip.swish_fwd(h2)
        cond16 = hip.cast_f32_to_bf16(cond)
        mod = hip.linear_fwd(cond16, self.W("ada/w"), bias=self.F("ada/b"))
        return mod, ((temb, h1, s1, h2, cond16) if save else None)

    def _embed_actions(self, x_t: torch.Tensor):
        """action_in_proj (f32 nnx.Linear, lap.py:52) on the noisy actions -> bf16 suffix tokens."""
        B, S, ad = x_t.shape
        xt2 = x_t.reshape(B * S, ad).contiguous()
        return hip.cast_f32_to_bf16(self._lin32(xt2, "act/in_w", "act/in_b")), xt2

    def _embed_suffix_pi0(self, x_t: torch.Tensor, time: torch.Tensor, state: torch.Tensor, save: bool):
        """[UPSTREAM-RECALL] openpi Pi0.embed_suffix, pi0 branch (parameters: lap.py:56-61): a state token `state_proj(state)`, then the
        action tokens mixed with the time embedding: action_time_mlp_out(swish(action_time_mlp_in([action_in_proj(x_t) | posemb(t)]))).
        f32 nnx.Linear layers; -> bf16 suffix tokens [B * (S + 1), We], no adaRMS condition."""
        B, S, ad = x_t.shape
        We = self.e.width
        xt2 = x_t.reshape(B * S, ad).contiguous()
        a = self._lin32(xt2, "act/in_w", "act/in_b")
        temb = hip.posemb_sincos(time.contiguous(), We, 4e-3, 4.0)
        cat = torch.cat([a.view(B, S, We), temb[:, None, :].expand(B, S, We)], -1).reshape(B * S, 2 * We).contiguous()
        h1 = self._lin32(cat, "act/atime_in_w", "act/atime_in_b")
        s1 = hip.swish_fwd(h1)
        at = self._lin32(s1, "act/atime_out_w", "act/atime_out_b")
        st2 = state.to(self.device, torch.float32).reshape(B, ad).contiguous()
        st = self._lin32(st2, "act/state_w", "act/state_b")
        tok = torch.cat([st[:, None, :], at.view(B, S, We)], 1).reshape(B * (S + 1), We).contiguous()
        return hip.cast_f32_to_bf16(tok), ((xt2, cat, h1, s1, st2) if save else None)

    def _embed_suffix_pi0_bwd(self, sctx, dx1, B, S):
        xt2, cat, h1, s1, st2 = sctx
        We = self.e.width
        d = hip.cast_bf16_to_f32(dx1).view(B, S + 1, We)
        self._lin32_bwd(st2, d[:, 0].contiguous(), "act/state_w", "act/state_b", need_dx=False)
        ds1 = self._lin32_bwd(s1, d[:, 1:].reshape(B * S, We).contiguous(), "act/atime_out_w", "act/atime_out_b")
        dh1 = hip.swish_bwd(h1, ds1)
        dcat = self._lin32_bwd(cat, dh1, "act/atime_in_w", "act/atime_in_b")
        self._lin32_bwd(xt2, dcat[:, :We].contiguous(), "act/in_w", "act/in_b", need_dx=False)      # (the time half has no parameters behind it)

    def _embed_suffix(self, x_t: torch.Tensor, time: torch.Tensor, save: bool, overlap: bool = False):
        """overlap (train step): the dozen small kernels go to the second HIP stream and run under whatever the current stream was
        given before (the SigLIP tower); their results are next touched by `_llm_fwd`, which joins the streams."""
        sfx = self._suffix_stream(x_t, time) if overlap else None
        self.comm.wait_unit("ada", also=sfx)
        with (torch.cuda.stream(sfx) if sfx is not None else contextlib.nullcontext()):
            x1, xt2 = self._embed_actions(x_t)
            mod, tctx = self._time_mod(time, save)
        if sfx is not None:
            mod.record_stream(torch.cuda.current_stream()); x1.record_stream(torch.cuda.current_stream())
        return x1, mod, ((xt2, *tctx) if save else None)

    def _embed_suffix_bwd(self, sctx, dx1, dmod):
        """On the second HIP stream when there is one: it runs under the SigLIP backward that the caller issues next.  The
        caller joins the streams before it declares the "small" unit's gradients complete."""
        xt2, temb, h1, s1, h2, cond16 = sctx
        sfx = self._suffix_stream(dx1, dmod)
        with (torch.cuda.stream(sfx) if sfx is not None else contextlib.nullcontext()):
            dmod16 = hip.cast_f32_to_bf16(dmod)
            hip.colsum(dmod, self.G("ada/b"))
            self._wgrad(dmod16, cond16, "ada/w")
            dcond = hip.cast_bf16_to_f32(hip.linear_dgrad(dmod16, self.W("ada/w")))
            self.comm.grads_ready("ada")       # (its side stream starts behind the CURRENT stream: the one these gradients are on)
            dh2 = hip.swish_bwd(h2, dcond)
            ds1 = self._lin32_bwd(s1, dh2, "act/time_out_w", "act/time_out_b")
            dh1 = hip.swish_bwd(h1, ds1)
            self._lin32_bwd(temb, dh1, "act/time_in_w", "act/time_in_b", need_dx=False)
            self._lin32_bwd(xt2, hip.cast_bf16_to_f32(dx1), "act/in_w", "act/in_b", need_dx=False)
        return sfx

    # ================================================================== joint Gemma layers
    def _mod_slot(self, mod, slot):
        W3 = 3 * self.e.width
        return mod[:, slot * W3:(slot + 1) * W3]

    def _llm_fwd(self, x0, x1, mod, pos, qinfo, kinfo, B, n0, n1, save: bool, kv_cache=None, cache_out=None, collect=None,
                 mod_shared: bool = False, layers=None):
        """gemma.Module.__call__ layers (gemma.py:336-387,167-290).  x0 [B*n0, Dv] or None, x1 [B*n1, De] or None.
        kv_cache: per-layer (k, v) of the prefix used as key segment 0 when x0 is None (serving).
        Returns final pre-norm activations and the saved context."""
        v, e = self.v, self.e
        NH, HD, KV = v.num_heads, v.head_dim, v.num_kv_heads
        Ttot = pos.shape[1]
        ctx = [] if save else None
        mld = 0 if mod_shared else (mod.stride(0) if mod is not None else 0)  # 0: one modulation row for every sample
        sfx = self._suffix_stream(*([x1, mod] if mod is not None else [x1])) if (x0 is not None and x1 is not None) else None
        main = torch.cuda.current_stream() if sfx is not None else None
        on_sfx = (lambda: torch.cuda.stream(sfx)) if sfx is not None else contextlib.nullcontext
        for l in (range(v.depth) if layers is None else layers):    # `layers`: test hook (teacher-forced per-layer parity)
            self.comm.wait_unit(f"llm{l}", also=sfx)
            p = f"llm/{l}/"
            q = [None, None]; k = [None, None]; vv = [None, None]; h = [None, None]; rstd_a = [None, None]
            if x0 is not None:
                h[0], rstd_a[0] = hip.rmsnorm_fwd(x0, scale=self.F(p + "n_attn"), save_rstd=save)
                qkv = self._lin0(h[0], p + "wqkv0")
                q[0], k[0], vv[0] = hip.rope_split_fwd(qkv, pos, B, n0, Ttot, 0, NH, HD, HD ** -0.5)
                del qkv
            elif kv_cache is not None:
                k[0], vv[0] = kv_cache[l]
            if x1 is not None:
                with on_sfx():
                    if mod is None:     # pi0: plain RMSNorm in the expert (`use_adarms=[False, False]`, lap.py:51)
                        h[1], rstd_a[1] = hip.rmsnorm_fwd(x1, scale=self.F(p + "n_attn1"), save_rstd=save)
                    else:
                        h[1], rstd_a[1] = hip.rmsnorm_fwd(x1, mod=self._mod_slot(mod, 2 * l), rows_per_sample=n1, save_rstd=save, mod_ld=mld)
                    qkv = hip.linear_fwd(h[1], self.W(p + "wqkv1"))
                    q[1], k[1], vv[1] = hip.rope_split_fwd(qkv, pos, B, n1, Ttot, Ttot - n1, NH, HD, HD ** -0.5)
                    del qkv
                self._handoff(sfx, main, q[1], k[1], vv[1])
            if cache_out is not None:
                cache_out.append((k[0], vv[0]))
            qlen = [n0 if x0 is not None else 0, n1 if x1 is not None else 0]
            klen = [k[0].shape[0] // B if k[0] is not None else 0, n1 if x1 is not None else 0]
            o, lse = hip.attention_fwd(q, k, vv, qlen, klen, B, NH, KV, HD, qinfo, kinfo, need_lse=save)
            self._handoff(main, sfx, o[1])
            xa = [None, None]; y1 = None; hf = [None, None]; rstd_f = [None, None]; gu = [None, None]; act = [None, None]; y1f = None
            xn = [None, None]
            if x1 is not None and mod is None:       # pi0: plain residuals (gemma.py:577-583 with gate None)
                with on_sfx():
                    xa[1] = hip.linear_fwd(o[1], self.W(p + "wo1"), residual=x1)
                    hf[1], rstd_f[1] = hip.rmsnorm_fwd(xa[1], scale=self.F(p + "n_ffw1"), save_rstd=save)
                    gu[1] = hip.linear_fwd(hf[1], self.W(p + "wgu1"))
                    act[1] = hip.geglu_fwd(gu[1])
                    xn[1] = hip.linear_fwd(act[1], self.W(p + "wd1"), residual=xa[1])
            elif x1 is not None:     # (issued first: 8 short kernels that then run under the prefix stream's GEMMs)
                with on_sfx():
                    y1 = hip.linear_fwd(o[1], self.W(p + "wo1"))
                    xa[1] = hip.gated_residual_fwd(x1, y1, self._mod_slot(mod, 2 * l)[:, 2 * e.width:], n1, mld)
                    hf[1], rstd_f[1] = hip.rmsnorm_fwd(xa[1], mod=self._mod_slot(mod, 2 * l + 1), rows_per_sample=n1, save_rstd=save, mod_ld=mld)
                    gu[1] = hip.linear_fwd(hf[1], self.W(p + "wgu1"))
                    act[1] = hip.geglu_fwd(gu[1])
                    y1f = hip.linear_fwd(act[1], self.W(p + "wd1"))
                    xn[1] = hip.gated_residual_fwd(xa[1], y1f, self._mod_slot(mod, 2 * l + 1)[:, 2 * e.width:], n1, mld)
            if x0 is not None:
                xa[0] = self._lin0(o[0], p + "wo0", residual=x0)
                hf[0], rstd_f[0] = hip.rmsnorm_fwd(xa[0], scale=self.F(p + "n_ffw"), save_rstd=save)
                self.comm.pace(f"llm{l}")     # optimizer units released here start under the longest MFMA-bound GEMM of the layer
                if save and self.fuse_geglu_fwd and hip.linear_geglu_train_ok(hf[0], self.W(p + "wgu0")):
                    # gate | up projection with the GeGLU in its epilogue: gu (kept for the backward pass) and act leave one launch
                    gu[0], act[0] = hip.linear_geglu_train(hf[0], self.W(p + "wgu0"))
                else:
                    gu_out = None
                    if save and self.fuse_geglu_bwd:    # rows padded like d(gate | up): the fused backward kernel shares one row stride
                        gu_out = hip._padded_rows(hf[0].shape[0], 2 * v.mlp_dim, hf[0].device, hip._row_pad(2 * v.mlp_dim))
                    gu[0] = self._lin0(hf[0], p + "wgu0", out=gu_out)
                    act[0] = hip.geglu_fwd(gu[0], pad=self.gemm_dtype != "fp8")
                xn[0] = self._lin0(act[0], p + "wd0", residual=xa[0])
            if save:
                ctx.append(dict(x=[x0, x1], h=h, rstd_a=rstd_a, q=q, k=k, v=vv, o=o, lse=lse, xa=xa, y1=y1, hf=hf, rstd_f=rstd_f,
                                gu=gu, act=act, y1f=y1f))
            x0, x1 = xn
            if collect is not None:
                collect[f"llm/layer{l:02d}/x0"], collect[f"llm/layer{l:02d}/x1"] = x0, x1
        self._handoff(sfx, main, x1)
        return x0, x1, ctx

    def _llm_prefill(self, x0, pos, qinfo, kinfo, B, n0, cache_out, kv_events=None):
        """The prefix-only pass of `_llm_fwd` (x1 = None, nothing saved) for the serving prefill: K / V of every layer go to
        `cache_out`, the last layer's residual stream is returned.  Same operations and rounding points; the split-K projections
        leave f32 slabs and their consumers do the rest in one pass each — qkv: reduce + RoPE + head split (sin / cos of the
        prefix positions from one table for all layers), out / down: reduce + residual + the NEXT RMSNorm — 14 -> 10 launches
        per layer."""
        v = self.v
        NH, HD, KV = v.num_heads, v.head_dim, v.num_kv_heads
        Ttot = pos.shape[1]
        Dv = v.width
        scratch = hip._gemm_scratch(self.device)
        tab = hip.rope_table(pos, B, n0, Ttot, 0, HD)
        self.comm.wait_unit("llm0")
        h, _ = hip.rmsnorm_fwd(x0, scale=self.F("llm/0/n_attn"), save_rstd=False)
        rows = x0.shape[0]
        panel = (self.serve_panel and rows <= 640 and hip.panel_gemm_ok(rows, (NH + 2 * KV) * HD, Dv) and hip.panel_gemm_ok(rows, Dv, NH * HD))
        panel_q, panel_o = panel and "q" in self._panel_llm, panel and "o" in self._panel_llm
        pf = (lambda name: self._pw(name)) if panel_o and self.serve_prefetch else (lambda name: None)   # the next panel launch's weights
        for l in range(v.depth):
            p = f"llm/{l}/"
            if panel_q:   # one f32 slab, no K split (us, 560 rows: 23.5 -> 16.9 in front of the same consumer)
                part, ks = hip.panel_partials(h, self._pw(p + "wqkv0"), (NH + 2 * KV) * HD, scratch, 1, nt=2, prefetch=pf(p + "wo0"))
            else:
                part, ks = hip.linear_partials(h, self.W(p + "wqkv0"), scratch, ksplit=self._prefill_ks[0])
            q, k, vv = hip.fused_reduce_rope_split(part, ks, pos, B, n0, Ttot, 0, NH, HD, HD ** -0.5, table=tab)
            cache_out.append((k, vv))
            if kv_events is not None:     # layer l's K / V exist from here on: the first denoise step may use them (sample_actions)
                ev = torch.cuda.Event()
                ev.record()
                kv_events.append(ev)
            o, _ = hip.attention_fwd([q, None], [k, None], [vv, None], [n0, 0], [n0, 0], B, NH, KV, HD, qinfo, kinfo, need_lse=False)
            if panel_o:   # (16.2 -> 14.6)
                xa = hip.panel_linear(o[0], self._pw(p + "wo0"), Dv, residual=x0, nt=4)   # (the next panel launch is 200 MB of gate|up and down weights away)
                hf, _ = hip.rmsnorm_fwd(xa, scale=self.F(p + "n_ffw"), save_rstd=False)
            elif self._prefill_ks[1] > 1:
                part, ks = hip.linear_partials(o[0], self.W(p + "wo0"), scratch, ksplit=self._prefill_ks[1])
                xa, hf = hip.fused_reduce_norm(part, ks, rows, Dv, residual=x0, norm=1, gamma=self.F(p + "n_ffw"))
            else:   # (measured: the unsplit 64-row tile with the residual epilogue + a norm launch beats split + fused consumer here)
                xa = hip.linear_fwd(o[0], self.W(p + "wo0"), residual=x0)
                hf, _ = hip.rmsnorm_fwd(xa, scale=self.F(p + "n_ffw"), save_rstd=False)
            if rows <= 640 and (v.mlp_dim & 127) == 0:    # gate|up projection + GeGLU in one launch (the 320-row tile's paired epilogue)
                act = hip.linear_geglu(hf, self.W(p + "wgu0"), exp2=self._panel_gelu == "exp2")
            else:
                act = hip.geglu_fwd(hip.linear_fwd(hf, self.W(p + "wgu0")))
            part, ks = hip.linear_partials(act, self.W(p + "wd0"), scratch, ksplit=self._prefill_ks[2], tile=self._prefill_ks[3])
            if l + 1 < v.depth:
                self.comm.wait_unit(f"llm{l + 1}")
                x0, h = hip.fused_reduce_norm(part, ks, rows, Dv, residual=xa, norm=1, gamma=self.F(f"llm/{l + 1}/n_attn"))
            else:
                x0, h = hip.fused_reduce_norm(part, ks, rows, Dv, residual=xa, norm=0)
        return x0

    def _llm_bwd(self, ctx, dx0, dx1, mod, dmod, pos, qinfo, kinfo, B, n0, n1):
        v, e = self.v, self.e
        NH, HD, KV = v.num_heads, v.head_dim, v.num_kv_heads
        Ttot = pos.shape[1]
        has_sfx = dx1 is not None         # False: prefix-only backward (enable_action_training=False, lap.py:449-455)
        ada = has_sfx and mod is not None  # False with a suffix stream: pi0 (plain norms and residuals in the expert)
        ldm = mod.stride(0) if ada else 0
        zero_do0 = None
        sfx = self._suffix_stream(*([dx1, dmod, mod] if ada else [dx1]))
        main = torch.cuda.current_stream() if sfx is not None else None
        on_sfx = (lambda: torch.cuda.stream(sfx)) if sfx is not None else contextlib.nullcontext
        for l in reversed(range(v.depth)):
            p = f"llm/{l}/"
            c = ctx[l]
            d_o = [None, None]
            # ---- FFN + attention output, suffix stream: xn = xa + y1f * gate_f  (on the second HIP stream, see the module doc)
            slot_f, slot_a = 2 * l + 1, 2 * l
            if has_sfx and not ada:
                with on_sfx():      # xn = xa + act wd^T, xa = x + o wo^T: the residuals pass dx1 through, the norms add onto it in place
                    self._wgrad(dx1, c["act"][1], p + "wd1")
                    dact = hip.linear_dgrad(dx1, self.W(p + "wd1"))
                    dgu = hip.geglu_bwd(c["gu"][1], dact)
                    self._wgrad(dgu, c["hf"][1], p + "wgu1")
                    dhf = hip.linear_dgrad(dgu, self.W(p + "wgu1"))
                    hip.rmsnorm_bwd(c["xa"][1], dhf, c["rstd_f"][1], scale=self.F(p + "n_ffw1"), dx=dx1, dscale=self.G(p + "n_ffw1"), accum_dx=True)
                    self._wgrad(dx1, c["o"][1], p + "wo1")
                    d_o[1] = hip.linear_dgrad(dx1, self.W(p + "wo1"))
                    del dact, dgu, dhf
            elif has_sfx:
                with on_sfx():
                    gate_f = self._mod_slot(mod, slot_f)[:, 2 * e.width:]
                    dy1f = hip.gated_residual_bwd(dx1, c["y1f"], gate_f, n1, ldm, self._mod_slot(dmod, slot_f)[:, 2 * e.width:], dmod.stride(0))
                    self._wgrad(dy1f, c["act"][1], p + "wd1")
                    dact = hip.linear_dgrad(dy1f, self.W(p + "wd1"))
                    dgu = hip.geglu_bwd(c["gu"][1], dact)
                    self._wgrad(dgu, c["hf"][1], p + "wgu1")
                    dhf = hip.linear_dgrad(dgu, self.W(p + "wgu1"))
                    hip.rmsnorm_bwd(c["xa"][1], dhf, c["rstd_f"][1], mod=self._mod_slot(mod, slot_f), rows_per_sample=n1, dx=dx1,
                                    dmod=self._mod_slot(dmod, slot_f), accum_dx=True)
                    gate_a = self._mod_slot(mod, slot_a)[:, 2 * e.width:]
                    dy1 = hip.gated_residual_bwd(dx1, c["y1"], gate_a, n1, ldm, self._mod_slot(dmod, slot_a)[:, 2 * e.width:], dmod.stride(0))
                    self._wgrad(dy1, c["o"][1], p + "wo1")
                    d_o[1] = hip.linear_dgrad(dy1, self.W(p + "wo1"))
                    del dy1f, dact, dgu, dhf, dy1
            # ---- FFN, prefix stream: xn = xa + act @ wd^T   (dx0 is None: the whole prefix side is frozen)
            if dx0 is not None:
                self._wgrad(dx0, c["act"][0], p + "wd0")
                if self.fuse_geglu_bwd and hip.dgrad_geglu_bwd_ok(dx0, self.W(p + "wd0"), c["gu"][0]):
                    # the down projection's data gradient with the GeGLU backward as its epilogue: d(act) never reaches memory
                    dgu = hip.linear_dgrad_geglu_bwd(dx0, self.W(p + "wd0"), c["gu"][0])
                else:
                    dact = self._dgrad0(dx0, p + "wd0")
                    dgu = hip.geglu_bwd(c["gu"][0], dact, pad=self.gemm_dtype != "fp8")
                    del dact
                self._wgrad(dgu, c["hf"][0], p + "wgu0")
                dhf = self._dgrad0(dgu, p + "wgu0")
                del dgu
                self._wg_join(dx0)   # (the down projection's weight gradient reads dx0)
                hip.rmsnorm_bwd(c["xa"][0], dhf, c["rstd_f"][0], scale=self.F(p + "n_ffw"), dx=dx0, dscale=self.G(p + "n_ffw"), accum_dx=True)
                del dhf
                self._wgrad(dx0, c["o"][0], p + "wo0")
                d_o[0] = self._dgrad0(dx0, p + "wo0")
            else:   # the attention backward still needs a dO for the prefix queries: zero (their dq / dk / dv are discarded)
                if zero_do0 is None:
                    zero_do0 = torch.zeros_like(c["o"][0])
                d_o[0] = zero_do0
            # ---- attention
            self._handoff(sfx, main, d_o[1])
            dq, dk, dv = hip.attention_bwd(c["q"], c["k"], c["v"], c["o"], d_o, c["lse"], [n0, n1], [n0, n1], B, NH, KV, HD, qinfo, kinfo,
                                           stop_q1_to_k0=self.config.stop_action_to_vlm_grad)
            self._handoff(main, sfx, dq[1], dk[1], dv[1])
            if has_sfx:
                with on_sfx():
                    dqkv = hip.rope_split_bwd(dq[1], dk[1], dv[1], pos, B, n1, Ttot, Ttot - n1, NH, HD, HD ** -0.5)
                    self._wgrad(dqkv, c["h"][1], p + "wqkv1")
                    dh = hip.linear_dgrad(dqkv, self.W(p + "wqkv1"))
                    if ada:
                        hip.rmsnorm_bwd(c["x"][1], dh, c["rstd_a"][1], mod=self._mod_slot(mod, slot_a), rows_per_sample=n1, dx=dx1,
                                        dmod=self._mod_slot(dmod, slot_a), accum_dx=True)
                    else:
                        hip.rmsnorm_bwd(c["x"][1], dh, c["rstd_a"][1], scale=self.F(p + "n_attn1"), dx=dx1, dscale=self.G(p + "n_attn1"), accum_dx=True)
                    del dqkv, dh
            if dx0 is not None:
                dqkv = hip.rope_split_bwd(dq[0], dk[0], dv[0], pos, B, n0, Ttot, 0, NH, HD, HD ** -0.5)
                self._wgrad(dqkv, c["h"][0], p + "wqkv0")
                dh = self._dgrad0(dqkv, p + "wqkv0")
                self._wg_join(dx0)   # (the out projection's reads dx0)
                hip.rmsnorm_bwd(c["x"][0], dh, c["rstd_a"][0], scale=self.F(p + "n_attn"), dx=dx0, dscale=self.G(p + "n_attn"), accum_dx=True)
                del dqkv, dh
            ctx[l] = None
            self._unit_done(f"llm{l}", sfx)   # complete once both streams are through: the optimizer's stream waits for both, the
                                             # compute stream goes on (it meets the second stream again at the next attention)
        self._handoff(sfx, main)
        return dx0, dx1

    def _expert_denoise_fwd(self, x1, mod, pos, qinfo, kinfo, B, Pn, S, cache, rope_tab=None):
        """The 18 action-expert layers of one denoise step (gemma.py:336-387 with xs = [None, suffix], kv_cache given)
        on the fused serving kernels: every projection is a split-K GEMM that leaves f32 partials, and the reduction
        happens inside the consumer (RoPE+split / GeGLU / gated residual + next adaptive RMSNorm).  `mod` is the single
        modulation row of this step (shared by all samples).  Returns the final adaRMS-normed suffix activations."""
        v, e = self.v, self.e
        NH, HD, KV = v.num_heads, v.head_dim, v.num_kv_heads
        Ttot = pos.shape[1]
        We = e.width
        scratch = hip._gemm_scratch(self.device)
        slot = lambda j: self._mod_slot(mod, j)
        h, _ = hip.rmsnorm_fwd(x1, mod=slot(0), rows_per_sample=S, save_rstd=False, mod_ld=0)
        x = x1
        for l in range(v.depth):
            self.comm.wait_unit(f"llm{l}")
            p = f"llm/{l}/"
            part, ks = hip.linear_partials(h, self.W(p + "wqkv1"), scratch)
            q, k, vv = hip.fused_reduce_rope_split(part, ks, pos, B, S, Ttot, Ttot - S, NH, HD, HD ** -0.5, table=rope_tab)
            ck, cv = cache[l]
            o, _ = hip.attention_fwd([None, q], [ck, k], [cv, vv], [0, S], [Pn, S], B, NH, KV, HD, qinfo, kinfo, need_lse=False)
            part, ks = hip.linear_partials(o[1], self.W(p + "wo1"), scratch)
            xa, hf = hip.fused_reduce_residual_norm(part, ks, x, slot(2 * l)[:, 2 * We:], 0, slot(2 * l + 1), 0, S)
            part, ks = hip.linear_partials(hf, self.W(p + "wgu1"), scratch)
            act = hip.fused_reduce_geglu(part, ks, B * S, e.mlp_dim)
            part, ks = hip.linear_partials(act, self.W(p + "wd1"), scratch)
            x, h = hip.fused_reduce_residual_norm(part, ks, xa, slot(2 * l + 1)[:, 2 * We:], 0, slot(2 * l + 2), 0, S)
        return h   # slot 2L is final_norm_1: h == final adaRMS norm of the last layer's output

    def _expert_denoise_skinny(self, x1, mod, qinfo, kinfo, B, Pn, S, cache, rope_tab, kv_events=None):
        """The same 18 layers on the skinny-M fused projections (csrc/serve_skinny.hip): five launches per layer —
        [adaRMS + qkv + RoPE/split] -> attention -> [out-proj + gated residual] -> [adaRMS + gate|up + GeGLU] ->
        [down-proj + gated residual] — with no f32 partial slabs in between.  Returns the last layer's residual stream
        (the final adaRMS norm is part of lap_serve_final_euler)."""
        v, e = self.v, self.e
        NH, HD, KV = v.num_heads, v.head_dim, v.num_kv_heads
        We = e.width
        slot = lambda j: self._mod_slot(mod, j)
        x = x1
        for l in range(v.depth):
            self.comm.wait_unit(f"llm{l}")
            p = f"llm/{l}/"
            q, k, vv = hip.serve_qkv_rope(x, slot(2 * l), 0, S, self.W(p + "wqkv1"), rope_tab, NH, HD, HD ** -0.5)
            ck, cv = cache[l]
            if kv_events is not None:     # running beside the prefill (first denoise step): layer l's cache is ready at its event
                torch.cuda.current_stream().wait_event(kv_events[l])
                ck.record_stream(torch.cuda.current_stream()); cv.record_stream(torch.cuda.current_stream())
            o, _ = hip.attention_fwd([None, q], [ck, k], [cv, vv], [0, S], [Pn, S], B, NH, KV, HD, qinfo, kinfo, need_lse=False)
            xa = hip.serve_proj_residual(o[1], self.W(p + "wo1"), x, slot(2 * l)[:, 2 * We:], 0, S)
            act = hip.serve_gate_up(xa, slot(2 * l + 1), 0, S, self.W(p + "wgu1"))
            x = hip.serve_proj_residual(act, self.W(p + "wd1"), xa, slot(2 * l + 1)[:, 2 * We:], 0, S)
        return x

    # ================================================================== training forward (+ backward)
    def _loss_impl(self, rng, observation: CoTObservation, actions: torch.Tensor, *, train: bool, noise=None, time=None,
                   backward: bool, collect: dict | None = None):
        cfg = self.config
        # lap.py:426-462,557-596: three branches — both losses (LAP-3B); enable_action_training=False: `llm([prefix])` and the
        # cross entropy only (VLA-0 style configs); enable_langact_training=False: both streams, flow matching only (pi0 style)
        act_on, lang_on = cfg.enable_action_training, cfg.enable_langact_training
        if not (act_on or lang_on):
            raise ValueError("LAPConfig with neither enable_action_training nor enable_langact_training has no loss")
        dev = self.device
        self.comm.wait_unit("small")
        g = _gen(rng, dev)
        obs = preprocess_observation(observation, train=train, image_keys=cfg.image_keys, image_resolution=cfg.image_resolution,
                                     enable_image_augmentation=cfg.enable_image_augmentation, rng=g)
        actions = actions.to(dev, torch.float32).contiguous()
        B, S, ad = actions.shape
        if S != self.action_horizon:
            raise ValueError(f"actions horizon {S} != action_horizon {self.action_horizon}")
        x1 = mod = sctx = u_t = None
        if act_on:
            # lap.py:185-207 prepare_suffix — noise ~ N(0,1), time ~ Beta(1.5, 1) * 0.999 + 0.001
            if noise is None:
                noise = torch.randn(actions.shape, generator=g, device=dev, dtype=torch.float32)
            if time is None:
                u1 = torch.rand(B, generator=g, device=dev, dtype=torch.float32)
                time = u1.pow(1.0 / 1.5) * 0.999 + 0.001  # Beta(a, 1) by inverse CDF
            noise = noise.to(dev, torch.float32).contiguous(); time = time.to(dev, torch.float32).contiguous()
            x_t, u_t = hip.fm_mix(noise, actions, time)
            if cfg.pi05:
                # suffix first: its small kernels go to the second HIP stream and run under the SigLIP tower issued next
                x1, mod, sctx = self._embed_suffix(x_t, time, backward, overlap=True)
            else:
                if obs.state is None:
                    raise ValueError("pi05=False feeds the continuous state through state_proj: the observation has no `state`")
                x1, sctx = self._embed_suffix_pi0(x_t, time, obs.state, backward)
        # suffix rows in the joint sequence: none without the action expert (lap.py:449-455); pi0 has its state token in front
        Sx = (S + (0 if cfg.pi05 else 1)) if act_on else 0
        x0, Pn, pctx = self._embed_prefix(obs, backward, collect)
        qinfo, kinfo, pos = self._train_infos(obs, S if act_on else 0)
        if collect is not None:
            collect["x0_in"], collect["x1_in"], collect["pos"], collect["mod"] = x0, x1, pos, mod
        xf0, xf1, lctx = self._llm_fwd(x0, x1, mod, pos, qinfo, kinfo, B, Pn, Sx, backward, collect=collect)
        if collect is not None:
            collect["x0_out"], collect["x1_out"] = xf0, xf1

        fb = lambda t: t.to(torch.float32)
        sm = obs.sample_mask if obs.sample_mask is not None else torch.ones(B, dtype=torch.bool, device=dev)
        lang_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        sel = pl = None
        Dv, V = self.v.width, cfg.vocab_size
        if lang_on:
            # ---- language loss (lap.py:209-289): rows Pn-Lt .. Pn-2 predict tokens 1 .. Lt-1
            Lt = obs.tokenized_prompt.shape[1]
            loss_mask = obs.tokenized_langact_mask[:, 1:] & obs.tokenized_prompt_mask[:, 1:]
            if obs.token_loss_mask is not None:
                loss_mask = loss_mask & obs.token_loss_mask[:, 1:]
            lm_bool = loss_mask if obs.sample_mask is None else loss_mask & obs.sample_mask[:, None]
            lm = lm_bool.to(torch.float32)
            cnt = torch.clamp(lm.sum(-1), min=1.0)
            # Only rows whose loss mask is set matter (the reference multiplies the other rows' cross entropy by 0): with the host
            # hint `loss_rows_max` the head runs on that many rows per sample — the masked ones first (stable order), padded with
            # rows of weight 0 — instead of all Lt - 1 (BASELINE shapes: 16 of 47).  A hint smaller than a sample's count would drop
            # tokens silently, so the device-side check turns the loss into NaN instead (no host sync).
            n_sel = observation.loss_rows_max if observation.loss_rows_max is not None else obs.loss_rows_max
            sel = None
            if n_sel is not None and 0 < n_sel < Lt - 1 and os.environ.get("LAP_LM_ALL_ROWS", "0") != "1":
                sel = torch.sort((~lm_bool).to(torch.uint8), dim=1, stable=True).indices[:, :n_sel]          # [B, n_sel] in 0 .. Lt-2
                hint_too_small = (lm.sum(-1) > n_sel).any()
                Ls = n_sel
                rowid = (torch.arange(B, device=dev) * Pn + (Pn - Lt))[:, None] + sel
                rows = xf0.index_select(0, rowid.view(-1))
                targets = obs.tokenized_prompt[:, 1:].gather(1, sel).to(torch.int32).contiguous().view(-1)
                lm_s = lm.gather(1, sel)
            else:
                Ls = Lt - 1
                rows = torch.empty((B * Ls, Dv), dtype=torch.bfloat16, device=dev)
                hip.copy_rows_bf16(xf0, rows, B * Ls, Ls, Dv, Pn, Pn - Lt, Ls, 0)
                targets = obs.tokenized_prompt[:, 1:].to(torch.int32).contiguous().view(-1)
                lm_s = lm
            R = B * Ls
            pl, rstd_pl = hip.rmsnorm_fwd(rows, scale=self.F("llm/final_norm"), save_rstd=backward)
            # Embedder.decode (gemma.py:153-154) multiplies the bf16 pre-logits by the F32 table: table = hi + lo, two bf16 planes
            # (16 mantissa bits; the products are exact in the f32 accumulator) -> logits to ~2^-17 of the f32 product
            table16, table_lo = self.W("llm/embed"), self.ps.w16lo("llm/embed")
            if os.environ.get("LAP_LM_NO_LO", "0") == "1":       # A/B switch: the bf16 mirror alone (the pre-round-3 dtype flow)
                table_lo = None
            # vocab chunks: one when [R, V] bf16 stays below the 2 GiB buffer-descriptor range of the GEMM (B <= 32 here)
            vc_max = max(1024, (int(1.5e9) // (2 * R)) // 1024 * 1024)
            chunks = [(v0, min(vc_max, V - v0)) for v0 in range(0, V, vc_max)]
            m = torch.full((R,), -3.0e38, dtype=torch.float32, device=dev)
            lsum = torch.zeros(R, dtype=torch.float32, device=dev); tl = torch.zeros(R, dtype=torch.float32, device=dev)
            logit_chunks = []
            for v0, vc in chunks:
                lg = torch.empty((R, vc), dtype=torch.float32, device=dev)
                hip.gemm(pl, table16[v0:v0 + vc], lg, M=R, N=vc, K=Dv, lda=Dv, ldb=Dv, ldc=vc)
                if table_lo is not None:
                    hip.gemm(pl, table_lo[v0:v0 + vc], lg, M=R, N=vc, K=Dv, lda=Dv, ldb=Dv, ldc=vc, accum=True)
                hip.ce_chunk_update(lg, targets, m, lsum, tl, v0)
                logit_chunks.append(lg if backward else None)
            nll = (m + torch.log(lsum) - tl).view(B, Ls)
            lang_loss = (nll * lm_s).sum(-1) / cnt
            if sel is not None:
                lang_loss = torch.where(hint_too_small, torch.full_like(lang_loss, float("nan")), lang_loss)

        # ---- action loss (lap.py:291-301)
        pre1 = v_t = None
        if act_on:
            if cfg.pi05:
                pre1, rstd_p1 = hip.rmsnorm_fwd(xf1, mod=self._mod_slot(mod, 2 * self.v.depth), rows_per_sample=S, save_rstd=backward)
            else:       # plain final norm; the action head reads the last S rows of each sample (`suffix_out[:, -ah:]`, lap.py:298)
                pre1_all, rstd_p1 = hip.rmsnorm_fwd(xf1, scale=self.F("llm/final_norm1"), save_rstd=backward)
                pre1 = pre1_all.view(B, Sx, self.e.width)[:, 1:].reshape(B * S, self.e.width).contiguous()
            pre1f = hip.cast_bf16_to_f32(pre1)
            v_t = self._lin32(pre1f, "act/out_w", "act/out_b")  # [B*S, ad]
        # ---- combination (lap.py:472-596).  Per-sample weights: language loss x {language, VQA (optionally per dataset),
        # prediction} weight by sample kind; action loss only on samples that are neither VQA nor prediction samples.
        mixing = lang_on and (cfg.enable_vqa_training or cfg.enable_prediction_training)
        vqa = obs.is_vqa_sample.to(dev, torch.bool) if (cfg.enable_vqa_training and obs.is_vqa_sample is not None) else None
        pred = obs.is_prediction_sample.to(dev, torch.bool) if (cfg.enable_prediction_training and obs.is_prediction_sample is not None) else None
        extra_metrics = {}
        if mixing:
            vqa_m = (vqa if vqa is not None else torch.zeros(B, dtype=torch.bool, device=dev)) & sm      # lap.py:480-486
            pred_m = (pred if pred is not None else torch.zeros(B, dtype=torch.bool, device=dev)) & sm
            lang_m = ~((vqa if vqa is not None else vqa_m) | (pred if pred is not None else pred_m)) & sm
            vqa_w = torch.full((B,), cfg.vqa_loss_weight, dtype=torch.float32, device=dev)               # lap.py:526-543
            if cfg.enable_vqa_training and cfg.vqa_loss_weights and obs.vqa_dataset_id is not None:
                from lap_amd.config import VQA_DATASET_ID_MAP

                ids = obs.vqa_dataset_id.to(dev)
                for name, wgt in cfg.vqa_loss_weights.items():
                    if name in VQA_DATASET_ID_MAP:
                        vqa_w = torch.where(ids == VQA_DATASET_ID_MAP[name], torch.full_like(vqa_w, float(wgt)), vqa_w)
            wl = vqa_w * fb(vqa_m) + cfg.prediction_loss_weight * fb(pred_m) + cfg.language_loss_weight * fb(lang_m)
            act_mask = ~vqa_m & ~pred_m          # the masks were AND-ed with the sample mask before this point (lap.py:484-485,562-566)
            n_act_loc = fb(sm).sum()
            for pfx, msk in (("vqa_", vqa_m), ("pred_", pred_m), ("langact_", lang_m)):   # metrics.py:49-56 (per-rank values)
                if (pfx == "vqa_" and not cfg.enable_vqa_training) or (pfx == "pred_" and not cfg.enable_prediction_training):
                    continue
                extra_metrics[pfx + "loss"] = (lang_loss * fb(msk)).sum() / torch.clamp(fb(msk).sum(), min=1.0)
                extra_metrics[pfx + "num_samples"] = fb(msk).sum()
                extra_metrics[pfx + "sample_portion"] = fb(msk).sum() / torch.clamp(n_act_loc, min=1.0)
            extra_metrics["active_num_samples"] = n_act_loc
            extra_metrics["active_sample_portion"] = n_act_loc / max(B, 1)
        else:   # (also the langact-off branch: the VQA / prediction masks reach the action loss as they came, lap.py:557-566)
            wl = torch.full((B,), cfg.language_loss_weight if lang_on else 0.0, dtype=torch.float32, device=dev)
            act_mask = torch.ones(B, dtype=torch.bool, device=dev)
            if vqa is not None:
                act_mask = act_mask & ~vqa
            if pred is not None:
                act_mask = act_mask & ~pred
        n_active = torch.clamp(self.comm.all_reduce_sum(fb(sm).sum().view(1)), min=1.0) if obs.sample_mask is not None else \
            self.comm.all_reduce_sum(torch.tensor([float(B)], device=dev))
        # lang_term: sum / active samples, or the batch mean without a sample mask (lap.py:579-596; the action-off branch's
        # `final_loss` is the same expression)
        lang_term = (wl * lang_loss).sum() / n_active
        act_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        action_term = 0.0
        dv = None
        if act_on:
            n_action = torch.clamp(self.comm.all_reduce_sum(fb(act_mask).sum().view(1)), min=1.0)
            coef = cfg.action_loss_weight * fb(act_mask) / n_action
            act_loss, dv = hip.mse_fwd_bwd(v_t.view(B, S * ad), u_t.view(B, S * ad), coef, need_grad=backward)
            action_term = (cfg.action_loss_weight * act_loss * fb(act_mask)).sum() / n_action
        loss = self.comm.all_reduce_sum((lang_term + action_term).view(1)).view(())
        metrics = {"lang_loss": lang_loss.mean(), "action_loss": (act_loss * fb(act_mask)).sum() / torch.clamp(fb(act_mask).sum(), min=1.0),
                   "langact_loss": (lang_loss * fb(sm)).sum() / torch.clamp(fb(sm).sum(), min=1.0) if not mixing else extra_metrics["langact_loss"],
                   **{k: v for k, v in extra_metrics.items() if k != "langact_loss"}}
        if collect is not None:
            collect.update(pl=pl, pre1=pre1, v_t=v_t.view(B, S, ad) if v_t is not None else None, u_t=u_t, per_sample_lang=lang_loss,
                           per_sample_action=act_loss)
        if not backward:
            return loss, metrics

        # =============================== backward ===============================
        self.comm.before_backward()
        self._wg_begin()
        We = self.e.width
        dmod = dx1 = None
        if act_on:      # action head
            dpre1f = self._lin32_bwd(pre1f, dv.view(B * S, ad), "act/out_w", "act/out_b")
            if cfg.pi05:
                dmod = torch.zeros(mod.shape, dtype=torch.float32, device=dev)
                dx1 = hip.rmsnorm_bwd(xf1, hip.cast_f32_to_bf16(dpre1f), rstd_p1, mod=self._mod_slot(mod, 2 * self.v.depth), rows_per_sample=S,
                                      dmod=self._mod_slot(dmod, 2 * self.v.depth))
            else:
                dall = torch.zeros((B, Sx, We), dtype=torch.bfloat16, device=dev)      # (the state token's row of the final norm has no consumer)
                dall[:, 1:] = hip.cast_f32_to_bf16(dpre1f).view(B, S, We)
                dx1 = hip.rmsnorm_bwd(xf1, dall.view(B * Sx, We), rstd_p1, scale=self.F("llm/final_norm1"), dscale=self.G("llm/final_norm1"))
        skip_prefix = self._prefix_frozen()
        dx0 = None
        if not skip_prefix and not lang_on:
            # no language loss: the prefix stream's only cotangents are those of its keys / values under the action queries
            dx0 = torch.zeros((B * Pn, Dv), dtype=torch.bfloat16, device=dev)
            # the LM-head weight-gradient product is what overwrites (beta = 0) the embedding table's f32 gradient buffer; without it the
            # scatter-add of _embed_prefix_bwd would accumulate onto the previous step's values
            if self.ps.is_trainable("llm/embed"):
                self.G("llm/embed").zero_()
        elif not skip_prefix:
            # language head: dlogits = w * (softmax - onehot); w = d loss / d nll.  The cotangent of the f32 logits stays f32 in the
            # reference (d pre_logits = dlogits . table, d table = dlogits^T . pre_logits in f32): dlogits = dh + dl (two bf16
            # planes), table = hi + lo -> dh.hi + dl.hi + dh.lo (dl.lo is 2^-16 of the sum) and (dh + dl)^T . pre_logits
            w = (wl[:, None] * lm_s / cnt[:, None] / n_active).contiguous().view(-1)
            hilo = table_lo is not None
            # hi / lo planes stacked along the rows, [dh; dl]: ONE weight-gradient product over 2R rows against [pl; pl] (the f32
            # [V, D] output is written once instead of accumulated onto), ONE data-gradient product [dh; dl] . hi (the table plane
            # is read once), plus dh . lo onto its first half
            RR = 2 * R if hilo else R
            pl2 = torch.cat([pl, pl], 0) if hilo else pl
            dpl32 = torch.empty((RR, Dv), dtype=torch.float32, device=dev) if (len(chunks) > 1 or hilo) else None
            gE = self.G("llm/embed")
            for ci, (v0, vc) in enumerate(chunks):
                dlogits = torch.empty((RR, vc), dtype=torch.bfloat16, device=dev)
                hip.ce_chunk_grad(logit_chunks[ci], targets, m, lsum, w, dlogits[:R], v0, dlogits_lo=dlogits[R:] if hilo else None)
                logit_chunks[ci] = None
                if self.ps.is_trainable("llm/embed"):
                    hip.linear_wgrad(dlogits, pl2, gE[v0:v0 + vc])
                if dpl32 is None:
                    dpl = hip.linear_dgrad(dlogits, table16[v0:v0 + vc])
                else:
                    hip.linear_dgrad(dlogits, table16[v0:v0 + vc], out=dpl32, accum=ci > 0)
                    if hilo:
                        hip.linear_dgrad(dlogits[:R], table_lo[v0:v0 + vc], out=dpl32[:R], accum=True)
                del dlogits
            if dpl32 is not None:
                dpl = hip.cast_f32_to_bf16(dpl32[:R] + dpl32[R:] if hilo else dpl32)
            drows = hip.rmsnorm_bwd(rows, dpl, rstd_pl, scale=self.F("llm/final_norm"), dscale=self.G("llm/final_norm"))
            dx0 = torch.zeros((B * Pn, Dv), dtype=torch.bfloat16, device=dev)
            if sel is not None:
                dx0.index_copy_(0, rowid.view(-1), drows)
            else:
                hip.copy_rows_bf16(drows, dx0, R, Lt - 1, Dv, Lt - 1, 0, Pn, Pn - Lt)
        dx0, dx1 = self._llm_bwd(lctx, dx0, dx1, mod, dmod, pos, qinfo, kinfo, B, Pn, Sx)
        sfx = None
        if act_on and cfg.pi05:
            sfx = self._embed_suffix_bwd(sctx, dx1, dmod)
        elif act_on:
            self._embed_suffix_pi0_bwd(sctx, dx1, B, S)
        elif "ada" in self.ps.unit_by_name:   # the action expert's units saw no gradient (zeros): still declared complete for the optimizer's pipeline
            self.comm.grads_ready("ada")
        if not skip_prefix:
            self._embed_prefix_bwd(pctx, dx0, B, Pn)
        self._handoff(sfx, torch.cuda.current_stream() if sfx is not None else None)
        self._wg_end()
        self.comm.grads_ready("small")
        return loss, metrics

    def compute_loss(self, rng, observation, actions, *, train: bool = False, stage_config=None, verbose_mode=None,
                     return_augmented_images: bool = False, noise=None, time=None, collect=None):
        """lap.py:380-602.  rng: int seed or torch.Generator.  `noise` / `time` may be given explicitly (parity tests)."""
        return self._loss_impl(rng, observation, actions, train=train, noise=noise, time=time, backward=False, collect=collect)

    def loss_and_grad(self, rng, observation, actions, *, train: bool = True, noise=None, time=None, collect=None):
        """Forward + backward; gradients land in self.ps.grad (engine layout; bf16 for the GEMM-weight units, f32 for the embedding table and
        the small unit: ParamStore.grad_dtype).  The caller zeroes the accumulated (f32) units first."""
        return self._loss_impl(rng, observation, actions, train=train, noise=noise, time=time, backward=True, collect=collect)

    # ================================================================== serving
    @torch.no_grad()
    def sample_actions(self, rng, observation, *, num_steps: int = 10, noise=None, collect=None, fused=True):
        """lap.py:605-675: prefix prefill once -> per-layer K/V kept in HBM -> `num_steps` Euler steps of the action
        expert attending to [cached prefix | fresh suffix] as two key segments (the reference concatenates, gemma.py:228-230).
        `fused`: True = the fastest denoise-step kernels the shapes allow ("skinny" fused projections for the LAP-3B action
        expert, else "partials" = split-K partial slabs + fused consumers); False = the generic layer path (A/B tests)."""
        cfg = self.config
        dev = self.device
        self.comm.wait_unit("small")
        obs = preprocess_observation(observation, train=False, image_keys=cfg.image_keys, image_resolution=cfg.image_resolution)
        B = obs.tokenized_prompt.shape[0]
        S, ad = self.action_horizon, self.action_dim
        if fused is True:
            fused = "skinny" if hip.serve_supported(self.e.width, self.v.head_dim, self.e.mlp_dim, self.v.num_heads) else "partials"
        if fused not in (False, "skinny", "partials"):
            raise ValueError(f"fused={fused!r}")
        if noise is None:
            noise = torch.randn((B, S, ad), generator=_gen(rng, dev), device=dev, dtype=torch.float32)
        x_t = noise.to(dev, torch.float32).contiguous().clone()
        x0, Pn, _ = self._embed_prefix(obs, False, serve=True)
        qinfo_p, kinfo_p, ppos, qinfo_s, kinfo_all, pos_all = self._serve_infos(obs, S)
        if not cfg.pi05:
            return self._sample_actions_pi0(obs, x_t, x0, Pn, (qinfo_p, kinfo_p, ppos, qinfo_s, kinfo_all, pos_all), num_steps, collect)
        dt = -1.0 / num_steps
        times, t = [], 1.0
        while t >= -dt / 2:  # lap.py:669-674 loop condition, unrolled on the host (the time grid is data independent)
            times.append(t)
            t += dt
        # the adaRMS condition depends on the denoise time only: all steps' modulations in one pass over the adaRMS bank —
        # and on nothing else, so they are computed ONCE per (step count, parameter version) and kept: a captured sampler
        # (serve.GraphedSampler warms up before it captures) holds no time-MLP kernels at all (-0.25 ms per chunk)
        mods = self._serve_mods(len(times), dt)
        # ... and so are the action tokens' positions: one sin / cos table serves the 10 x 18 fused RoPE kernels
        rope_tab = hip.rope_table(pos_all, B, S, pos_all.shape[1], pos_all.shape[1] - S, self.v.head_dim) if fused else None
        cache = []
        fast_prefill = self.serve_fusions and collect is None and self.gemm_dtype == "bf16"
        # The first denoise step needs layer l's K / V only when it reaches layer l: it is issued on a second stream and runs
        # beside the prefill's layers l+1 .. (whose big GEMMs leave CUs idle at their tails), joined before step 1.
        overlap = fast_prefill and fused == "skinny" and self.serve_overlap and dev.type == "cuda" and len(times) > 1
        kv_events = [] if overlap else None
        if overlap:
            if self._den is None:
                self._den = torch.cuda.Stream(dev)
            start = torch.cuda.Event()
            start.record()
        if fast_prefill:
            self._llm_prefill(x0, ppos, qinfo_p, kinfo_p, B, Pn, cache, kv_events=kv_events)
        else:
            self._llm_fwd(x0, None, None, ppos, qinfo_p, kinfo_p, B, Pn, 0, False, cache_out=cache)
        nslot = 2 * self.v.depth
        chain = (fused == "skinny" and self.serve_chain and not overlap and dev.type == "cuda" and self.v.depth <= hip.CHAIN_MAX_DEPTH
                 and hip.serve_chain_ok(B, S, self.e.width, self.e.mlp_dim, self.v.num_heads, self.v.head_dim, self.v.num_kv_heads, Pn))
        if chain:
            if self._chain_ctr is None:
                self._chain_ctr = hip.serve_chain_counters(dev)
            for l in range(self.v.depth):
                self.comm.wait_unit(f"llm{l}")
            tp = (self.serve_packed and self.serve_tp
                  and hip.serve_chain_tp_ok(B, S, self.e.width, self.e.mlp_dim, self.v.num_heads, self.v.head_dim, self.v.num_kv_heads, Pn))
            if self.serve_packed:
                chain_w = self._serve_packed_weights()
                if self._chain_scratch is None or (tp and "tp_slabs" not in self._chain_scratch):
                    self._chain_scratch = hip.serve_chain_scratch(dev, self.e.width, self.e.mlp_dim, self.v.num_heads, self.v.head_dim, tp=tp)
            else:
                chain_w = [tuple(self.W(f"llm/{l}/{n}") for n in ("wqkv1", "wo1", "wgu1", "wd1")) for l in range(self.v.depth)]
        for step in range(len(times)):
            mod = mods[step:step + 1]
            if chain:
                fuse_tail = self.serve_euler_embed and ad in (7, 8) and self.e.width == 1024     # the step's tail embeds the next step's tokens in the same launch
                if not fuse_tail or step == 0:
                    x1 = hip.serve_embed_actions(x_t.view(B * S, ad), self.F("act/in_w"), self.F("act/in_b"))
                xf1 = hip.serve_chain(x1, mod, 3 * self.e.width, chain_w, cache, rope_tab, qinfo_s, kinfo_all, B, S, self.v.num_heads,
                                      self.v.head_dim, self.e.mlp_dim, Pn, self.v.head_dim ** -0.5, self._chain_ctr,
                                      packed_scratch=self._chain_scratch if self.serve_packed else None, tp=tp)
                v_t = torch.empty((B * S, ad), dtype=torch.float32, device=dev) if collect is not None else None
                if fuse_tail:
                    x1 = torch.empty((B * S, self.e.width), dtype=torch.bfloat16, device=dev) if step + 1 < len(times) else None
                    hip.serve_final_euler_embed(xf1, self._mod_slot(mod, nslot), 0, S, self.F("act/out_w"), self.F("act/out_b"), x_t.view(B * S, ad), dt, v_t,
                                                w_in=self.F("act/in_w"), b_in=self.F("act/in_b"), tokens=x1)
                else:
                    hip.serve_final_euler(xf1, self._mod_slot(mod, nslot), 0, S, self.F("act/out_w"), self.F("act/out_b"), x_t.view(B * S, ad), dt, v_t)
                if collect is not None:
                    collect[f"v_t/{step}"] = v_t.view(B, S, ad)
                continue
            if fused == "skinny":
                side = overlap and step == 0
                main = torch.cuda.current_stream() if side else None
                if side:
                    self._den.wait_event(start)
                    for tns in (x_t, mods, rope_tab, qinfo_s, kinfo_all):
                        tns.record_stream(self._den)
                with (torch.cuda.stream(self._den) if side else contextlib.nullcontext()):
                    x1 = hip.serve_embed_actions(x_t.view(B * S, ad), self.F("act/in_w"), self.F("act/in_b"))
                    xf1 = self._expert_denoise_skinny(x1, mod, qinfo_s, kinfo_all, B, Pn, S, cache, rope_tab, kv_events=kv_events if side else None)
                    v_t = torch.empty((B * S, ad), dtype=torch.float32, device=dev) if collect is not None else None
                    hip.serve_final_euler(xf1, self._mod_slot(mod, nslot), 0, S, self.F("act/out_w"), self.F("act/out_b"), x_t.view(B * S, ad), dt, v_t)
                if side:
                    main.wait_stream(self._den)
                if collect is not None:
                    collect[f"v_t/{step}"] = v_t.view(B, S, ad)
                continue
            x1, _ = self._embed_actions(x_t)
            if fused == "partials":
                pre1 = self._expert_denoise_fwd(x1, mod, pos_all, qinfo_s, kinfo_all, B, Pn, S, cache, rope_tab)
            else:  # generic path (same numerics; kept for A/B tests)
                _, xf1, _ = self._llm_fwd(None, x1, mod, pos_all, qinfo_s, kinfo_all, B, Pn, S, False, kv_cache=cache, mod_shared=True)
                pre1, _ = hip.rmsnorm_fwd(xf1, mod=self._mod_slot(mod, nslot), rows_per_sample=S, save_rstd=False, mod_ld=0)
            v_t = self._lin32(hip.cast_bf16_to_f32(pre1), "act/out_w", "act/out_b")
            if collect is not None:
                collect[f"v_t/{step}"] = v_t.view(B, S, ad).clone()
            hip.axpy_f32(x_t, v_t, dt)
        return x_t

    def _sample_actions_pi0(self, obs, x_t, x0, Pn, infos, num_steps, collect):
        """`sample_actions` for pi0 (`pi05=False`): the suffix is [state token | action tokens mixed with the time embedding], embedded anew
        at every Euler step; the expert's layers run on the generic layer loop (plain norms and residuals) against the prefix cache.
        Functional path: no fused serving kernels, no hipGraph chain (no registry config of the reference is pi0)."""
        if obs.state is None:
            raise ValueError("pi05=False feeds the continuous state through state_proj: the observation has no `state`")
        qinfo_p, kinfo_p, ppos, qinfo_s, kinfo_all, pos_all = infos
        B, S, ad = x_t.shape
        Sx, We = S + 1, self.e.width
        cache = []
        self._llm_fwd(x0, None, None, ppos, qinfo_p, kinfo_p, B, Pn, 0, False, cache_out=cache)
        dt = -1.0 / num_steps
        t, step = 1.0, 0
        while t >= -dt / 2:      # lap.py:669-674
            x1, _ = self._embed_suffix_pi0(x_t, torch.full((B,), t, dtype=torch.float32, device=self.device), obs.state, False)
            _, xf1, _ = self._llm_fwd(None, x1, None, pos_all, qinfo_s, kinfo_all, B, Pn, Sx, False, kv_cache=cache)
            pre_all, _ = hip.rmsnorm_fwd(xf1, scale=self.F("llm/final_norm1"), save_rstd=False)
            pre1 = pre_all.view(B, Sx, We)[:, 1:].reshape(B * S, We).contiguous()
            v_t = self._lin32(hip.cast_bf16_to_f32(pre1), "act/out_w", "act/out_b")
            if collect is not None:
                collect[f"v_t/{step}"] = v_t.view(B, S, ad).clone()
            hip.axpy_f32(x_t, v_t, dt)
            t += dt
            step += 1
        return x_t

    def serve_chain_failed(self) -> bool:
        """True if a launch of the one-launch denoise step gave up at a grid barrier since the last check (synchronises)."""
        return self._chain_ctr is not None and hip.serve_chain_failed(self._chain_ctr)

    def disable_serve_chain(self):
        """After a launch that gave up at a barrier: one step back — the tensor-parallel chain (which needs 32 blocks on each of 8
        XCDs) falls back to the flat chain, the flat chain to the separate launches."""
        if self.serve_chain and self.serve_packed and self.serve_tp:
            self.serve_tp = False
        else:
            self.serve_chain = False
        if self._chain_ctr is not None:
            torch.cuda.synchronize(self.device)
            self._chain_ctr.zero_()

    def _serve_mods(self, nsteps: int, dt: float):
        """adaRMS modulations of the denoise time grid t_k = 1 + k dt (lap.py:655-660 through `_time_mod`), one PERSISTENT tensor
        per (nsteps, dt), refreshed IN PLACE when the parameters change: a captured sampler holds its address (ADVICE r3 — a
        cache that dropped the tensor on a version change left a replayed graph reading freed memory and stale modulations).
        `GraphedSampler.__call__` calls `refresh_serve_caches` before every replay; never refreshed during stream capture."""
        key = (nsteps, float(dt))
        ent = self.__dict__.setdefault("_mods_cache", {})
        rec = ent.get(key)
        if rec is None or rec[0] != self.ps.version:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("serving caches are stale inside a stream capture: call refresh_serve_caches() first")
            tvec = 1.0 + dt * torch.arange(nsteps, dtype=torch.float32, device=self.device)
            self.comm.wait_unit("ada")
            new = self._time_mod(tvec, False)[0]
            if rec is None:
                ent[key] = rec = [self.ps.version, new]
            else:
                rec[1].copy_(new)
                rec[0] = self.ps.version
        return rec[1]

    def _serve_packed_weights(self):
        """Fragment-packed images of the action expert's projections for the packed chain (csrc/serve_skinny_body.hpp PK),
        persistent like `_serve_mods` and re-packed in place per parameter version (+0.62 GB for LAP-3B)."""
        rec = self._packed_w
        if rec is not None and rec[0] == self.ps.version:
            return rec[1]
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("serving caches are stale inside a stream capture: call refresh_serve_caches() first")
        kinds = (("wqkv1", hip.PACK_QKV), ("wo1", hip.PACK_PLAIN), ("wgu1", hip.PACK_GATE_UP), ("wd1", hip.PACK_PLAIN))
        old = rec[1] if rec is not None else None
        out = []
        for l in range(self.v.depth):
            self.comm.wait_unit(f"llm{l}")
            out.append(tuple(hip.serve_pack_weight(self.W(f"llm/{l}/{n}"), kind, self.v.head_dim, out=None if old is None else old[l][j])
                             for j, (n, kind) in enumerate(kinds)))
        self._packed_w = [self.ps.version, out]
        return out

    def _pw(self, name):
        """Fragment-packed image (lap_serve_pack_weight kind 3) of a prefill projection for the row-panel kernel, persistent like
        `_serve_packed_weights` and re-packed in place per parameter version."""
        rec = self._prefill_pw.get(name)
        if rec is None or rec[0] != self.ps.version:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("serving caches are stale inside a stream capture: call refresh_serve_caches() first")
            img = hip.serve_pack_weight(self.W(name), hip.PACK_PLAIN, out=None if rec is None else rec[1])
            self._prefill_pw[name] = rec = [self.ps.version, img]
        return rec[1]

    def _time_grid(self, num_steps: int):
        dt = -1.0 / num_steps
        n, t = 0, 1.0
        while t >= -dt / 2:  # lap.py:669-674 loop condition
            n += 1
            t += dt
        return n, dt

    def refresh_serve_caches(self, num_steps: int = 10):
        """Bring the sampler's parameter-derived caches (adaRMS modulations, packed expert weights) up to the current parameter
        version, in place.  Eager `sample_actions` does this itself; a captured graph cannot — call it before a replay."""
        n, dt = self._time_grid(num_steps)
        for name in list(self._prefill_pw):
            kind, l = name.split("/")[:2]
            self.comm.wait_unit(f"{kind}{l}")
            self._pw(name)
        if not self.config.pi05:
            return          # (pi0 has no adaRMS bank and runs on the generic layer loop: nothing else is cached)
        self._serve_mods(n, dt)
        if self._packed_w is not None:
            self._serve_packed_weights()

    EOS_TOKEN = 1   # PaliGemma <eos> (lap.py: self.EOS_TOKEN)

    def _vlm_decode_step(self, token, pos, step, cache, gen, qinfo_d, kinfo_prefix, B, Pn):
        """One expert-0 decode step (lap.py:734-752): embed the sampled token, run the 18 VLM layers on that single row
        per sample with keys = [prefilled prefix cache | generated tokens incl. this one], return f32 logits [B, V].
        The reference appends into a fixed-size cache by index (gemma.py:597-605); here the generated keys are a second
        key segment that grows by one row per step."""
        v = self.v
        NH, HD, KV, Dv = v.num_heads, v.head_dim, v.num_kv_heads, v.width
        dev = self.device
        x = torch.empty((B, Dv), dtype=torch.bfloat16, device=dev)
        rows, lo, hi = self.ps.embed_rows()
        hip.embed_gather(rows, token.view(B, 1).contiguous(), x, B, 1, Dv, 1, 0, math.sqrt(Dv), lo, hi)
        kinfo = torch.cat([kinfo_prefix, torch.full((B, step + 1), 1 << 24, dtype=torch.int32, device=dev)], 1).contiguous()
        for l in range(v.depth):
            p = f"llm/{l}/"
            h, _ = hip.rmsnorm_fwd(x, scale=self.F(p + "n_attn"), save_rstd=False)
            qkv = hip.linear_fwd(h, self.W(p + "wqkv0"))
            q, k, vv = hip.rope_split_fwd(qkv, pos, B, 1, 1, 0, NH, HD, HD ** -0.5)
            gk, gv = gen[l]
            gk = k.view(B, 1, -1) if gk is None else torch.cat([gk, k.view(B, 1, -1)], 1)
            gv = vv.view(B, 1, -1) if gv is None else torch.cat([gv, vv.view(B, 1, -1)], 1)
            gen[l] = (gk, gv)
            ck, cv = cache[l]
            o, _ = hip.attention_fwd([None, q], [ck, gk.view(B * (step + 1), -1)], [cv, gv.view(B * (step + 1), -1)], [0, 1],
                                     [Pn, step + 1], B, NH, KV, HD, qinfo_d, kinfo, need_lse=False)
            xa = hip.linear_fwd(o[1], self.W(p + "wo0"), residual=x)
            hf, _ = hip.rmsnorm_fwd(xa, scale=self.F(p + "n_ffw"), save_rstd=False)
            act = hip.geglu_fwd(hip.linear_fwd(hf, self.W(p + "wgu0")))
            x = hip.linear_fwd(act, self.W(p + "wd0"), residual=xa)
        return self._lm_logits(x)

    def _lm_logits(self, rows):
        """final norm + Embedder.decode (gemma.py:153-154, 525-527): f32 logits [R, V]."""
        pl, _ = hip.rmsnorm_fwd(rows, scale=self.F("llm/final_norm"), save_rstd=False)
        V, Dv = self.config.vocab_size, self.v.width
        lg = torch.empty((rows.shape[0], V), dtype=torch.float32, device=self.device)
        hip.gemm(pl, self.W("llm/embed"), lg, M=rows.shape[0], N=V, K=Dv, lda=Dv, ldb=Dv, ldc=V)
        lo = self.ps.w16lo("llm/embed")      # the f32 table as hi + lo (see _loss_impl)
        if lo is not None:
            hip.gemm(pl, lo, lg, M=rows.shape[0], N=V, K=Dv, lda=Dv, ldb=Dv, ldc=V, accum=True)
        return lg

    def sample_tokens(self, rng, observation, *, max_decoding_steps: int = 390, temperature: float = 0.0, collect=None):
        """lap.py:678-766 (LAP_AR serving mode): VLM-only prefill, then single-token decode until every sample has emitted
        EOS or `max_decoding_steps` tokens; returns int32 [B, max_decoding_steps] (zeros after the stop).

        The reference right-aligns the prefix (`left_to_right_align`) so that a fixed-size cache can be addressed by
        `prefix_start`; rolling changes neither the attention pattern nor `cumsum(mask) - 1` positions of valid tokens,
        so the engine keeps the tokens in place and expresses the decode mask `[prefix_start, prefill_size + step]` in
        un-rolled coordinates: prefix keys `seqlen - prefill_len <= j < seqlen` (seqlen = last valid index + 1) plus every
        generated key.  Known deviation: with a hole inside the prefix (a masked-out image followed by valid tokens) the
        reference's range mask covers the hole's tokens, whose prefill activations above layer 0 are softmax outputs of
        fully masked rows (uniform averages, "never consumed" elsewhere); the engine's attention writes zeros for such
        rows, so decode logits differ from the reference in that case only (prefill logits still agree; tested).
        temperature > 0 samples with the Gumbel-max trick from a torch generator seeded by `rng` (the JAX PRNG stream of
        `jax.random.categorical` cannot be reproduced)."""
        cfg = self.config
        dev = self.device
        if self.comm.world_size != 1:
            raise NotImplementedError("sample_tokens is a serving path: replicas only (SURVEY.md §8e)")
        self.comm.wait_unit("small")
        obs = preprocess_observation(observation, train=False, image_keys=cfg.image_keys, image_resolution=cfg.image_resolution)
        B = obs.tokenized_prompt.shape[0]
        x0, Pn, _ = self._embed_prefix(obs, False, serve=True)
        qinfo_p, kinfo_p, ppos = self._serve_infos(obs, 1)[:3]
        prefix_mask, _ = self._prefix_masks(obs)
        ar = torch.arange(Pn, device=dev)
        seqlen = (prefix_mask.to(torch.int64) * ar).max(-1).values + 1          # left_to_right_align's roll amount
        plen = prefix_mask.sum(-1)                                               # prefill_len
        in_range = (ar[None] >= (seqlen - plen)[:, None]) & (ar[None] < seqlen[:, None])
        kinfo_prefix = (in_range.to(torch.int32) << 24).contiguous()
        qinfo_d = torch.full((B, 1), (1 << 24) | 0xFFFFFF, dtype=torch.int32, device=dev)
        cache = []
        if self.serve_fusions and self.gemm_dtype == "bf16":
            xf0 = self._llm_prefill(x0, ppos, qinfo_p, kinfo_p, B, Pn, cache)
        else:
            xf0, _, _ = self._llm_fwd(x0, None, None, ppos, qinfo_p, kinfo_p, B, Pn, 0, False, cache_out=cache)
        last = (torch.arange(B, device=dev) * Pn + seqlen - 1)
        logits = self._lm_logits(xf0.index_select(0, last).contiguous())        # decodes the first token (lap.py:716)
        out = torch.zeros((B, max_decoding_steps), dtype=torch.int32, device=dev)
        eos = torch.zeros((B,), dtype=torch.bool, device=dev)
        gen = [(None, None)] * self.v.depth
        g = _gen(rng, dev) if temperature > 0.0 else None
        step = 0
        while step < max_decoding_steps:
            if temperature > 0.0:
                u = torch.rand(logits.shape, generator=g, device=dev, dtype=torch.float32).clamp_(1e-20, 1.0)
                logits = logits / temperature - torch.log(-torch.log(u))
            token = hip.argmax_rows(logits)
            if collect is not None:
                collect[f"logit/{step}"] = logits.clone()
            out[:, step] = token
            eos |= token == self.EOS_TOKEN
            step += 1
            if step >= max_decoding_steps or bool(eos.all()):   # lap.py:754-756 loop condition (the unused last decode is skipped)
                break
            pos = (plen + (step - 1)).to(torch.int32).view(B, 1).contiguous()
            logits = self._vlm_decode_step(token, pos, step - 1, cache, gen, qinfo_d, kinfo_prefix, B, Pn)
        return out
