"""bench.py — LAP-3B bf16 train-step throughput on MI355X (BASELINE.json metric, config[1] / config[2]) and, at N = 1,
the batch-1 action-chunk latency (config[3]) in the same JSON line.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run on 127.0.0.1, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full train step (forward + hand-written backward + clip + AdamW + EMA, scripts/train.py:329-419) of
LAP-3B (SigLIP So400m/14 + Gemma-2B + Gemma-300M action expert, random-init weights) on a synthetic batch of 32
samples per GPU: 2 x 224x224 images, 48-token prompt (last 16 = language-action tokens), 50-step action chunk
(SURVEY.md §8d).  Weak scaling: per-GPU batch fixed, parameters/optimizer FSDP-sharded over RCCL.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (the bf16 MFMA GEMM family, csrc/gemm.hip: gemm_sp_kernel + gemm_pq_kernel + gemm_kernel):
achieved = sum of 2*M*N*K over the GEMM launches of the timed steps / their summed durations, measured with HIP
events on the launch stream; `cpu_baseline` times the CPU oracle (oracle/lap_oracle.py, kind "port") on the host
cores on a bounded slice of the same workload.  `serve` is the second half of the metric: prefix prefill + 10 denoise
steps of `sample_actions` at batch 1, hipGraph-replayed, against the 13.4 GB of algorithmic HBM bytes.
`roofline.traffic` is the HBM-side byte count of ONE launch of the dominant kernel on its largest shape from the
committed rocprofv3 --pmc passes (profiles/, file named in `traffic_source`); rocprofv3 cannot run inside this process.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

SERVE_BYTES = 4.79e9 + 10 * 0.86e9 + 10 * 10.3e6   # prefix weights once + 10 x (expert weights + KV)
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy)
def gemm_traffic():
    """`roofline.traffic`: L2-to-fabric bytes of ONE launch of the step's largest GEMM (gate|up forward, M = 17920, N = 32768,
    K = 2048) from the NEWEST committed rocprofv3 --pmc pass (profiles/rNN_gemm_pmc_counters.txt, written by tools/pmc_traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, raw counter unit KB): FETCH_SIZE x 2 (gfx950 half-count of wide coalesced reads,
    MI355X_MICROARCH.md section HBM) + WRITE_SIZE.  rocprofv3 cannot run inside this process: the figure is static per commit
    (`traffic_measured_in_run` false) and names its source file.  These are requests the L2s send to the fabric; the Infinity
    Cache serves part of them (the guide: hits are counted, not excluded), so it is an upper bound of the HBM bytes."""
    import glob
    import re

    alg = {"plain": 73.4e6 + 134.2e6 + 1174.4e6,           # A 17920 x 2048 + B 32768 x 2048 read, C 17920 x 32768 written (bf16)
           "geglu": 73.4e6 + 134.2e6 + 1174.4e6 + 587.2e6}  # ... + act = gelu(gate) * up, 17920 x 16384, written by the same launch
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_pmc_counters.txt")):
        m = re.match(r"r(\d+)([a-z]?)_", os.path.basename(f))
        txt = open(f).read()
        fe = re.findall(r"FETCH_SIZE per launch \(raw counter units, KB\): ([0-9.eE+]+)", txt)
        wr = re.findall(r"WRITE_SIZE per launch \(raw counter units, KB\): ([0-9.eE+]+)", txt)
        if m and fe and wr:
            key = (int(m.group(1)), m.group(2))
            if best is None or key > best[0]:
                kind = "geglu" if "traffic kernel: lap_gemm_asm_nt_geglu" in txt else "plain"
                best = (key, float(fe[0]) * 1e3, float(wr[0]) * 1e3, os.path.relpath(f, ROOT), kind)
    if best is None:
        return {"bytes_per_launch": None, "algorithmic_bytes_per_launch": alg["plain"], "shape": "gate-up fwd M=17920 N=32768 K=2048", "source": None}
    _, fetch, write, src, kind = best
    return {"bytes_per_launch": 2 * fetch + write, "algorithmic_bytes_per_launch": alg[kind], "source": src,
            "shape": "gate-up fwd M=17920 N=32768 K=2048 (" + ("lap_gemm_asm_nt_geglu: gate|up + GeGLU, writes gu and act" if kind == "geglu" else "lap_gemm_asm_nt") + ")"}


TRAIN_FLOP_PER_SAMPLE = 8.375e12  # 3 x forward (2.792 TFLOP), recompute not credited
MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA


def synthetic_batch(cfg, B, device, seed):
    from lap_amd.observation import CoTObservation

    g = torch.Generator(device="cpu").manual_seed(seed)
    L, H = cfg.max_token_len, cfg.image_size
    images = {k: (torch.rand(B, H, H, 3, generator=g) * 2 - 1).to(device) for k in cfg.image_keys}
    la = torch.zeros(B, L, dtype=torch.bool)
    la[:, L - 16:] = True
    obs = CoTObservation(
        images=images, image_masks={k: torch.ones(B, dtype=torch.bool, device=device) for k in cfg.image_keys},
        state=(torch.rand(B, cfg.action_dim, generator=g) * 2 - 1).to(device),
        tokenized_prompt=torch.randint(0, cfg.vocab_size, (B, L), generator=g, dtype=torch.int32).to(device),
        tokenized_prompt_mask=torch.ones(B, L, dtype=torch.bool, device=device),
        tokenized_langact_mask=la.to(device), token_loss_mask=torch.ones(B, L, dtype=torch.bool, device=device),
        sample_mask=torch.ones(B, dtype=torch.bool, device=device),
        loss_rows_max=int(la[:, 1:].sum(-1).max()))     # host-side hint the data loaders attach: 16 loss-carrying tokens per sample
    actions = torch.randn(B, cfg.action_horizon, cfg.action_dim, generator=g).to(device)
    return obs, actions


class GemmMeter:
    """Wraps lap_amd.hip.gemm: HIP event pair around every launch of the dominant kernel (on torch's current stream,
    which is the stream the C ABI launches on).  The model issues the action expert's GEMMs (1,600 rows, 1.7 % of the step's
    GEMM FLOPs) on a second stream under the prefix stream's: an event pair there mostly times the wait for free CUs, so those
    launches are counted (`second_stream`) but kept out of the achieved-rate sum — the compute stream's launches are timed
    with whatever the co-running kernels cost them included."""

    def __init__(self, hip):
        self.hip = hip
        self.orig = hip.gemm
        self.records = []
        self.enabled = False

    def install(self):
        def gemm(a, b, out, *, M, N, K, **kw):
            if not self.enabled:
                return self.orig(a, b, out, M=M, N=N, K=K, **kw)
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(a, b, out, M=M, N=N, K=K, **kw)
            e.record()
            if torch.cuda.current_stream().cuda_stream != self.main_stream:
                self.second.append(2.0 * M * N * K)
                self.second_ev.append((s, e))
            else:
                self.records.append((s, e, 2.0 * M * N * K))
                sig = (M, N, K, kw.get("lda"), kw.get("ldb"), kw.get("ldc"), bool(kw.get("a_kc", True)), bool(kw.get("b_kc", True)),
                       None if kw.get("bias") is None else kw["bias"].dtype, kw.get("residual") is not None, kw.get("ldr", 0),
                       kw.get("gelu", False), bool(kw.get("accum", False)), out.dtype, kw.get("tile", -1), kw.get("ksplit", 0))
                self.shapes[sig] = self.shapes.get(sig, 0) + 1
            return r
        self.hip.gemm = gemm
        # the GEMMs with a fused elementwise epilogue have entry points of their own (lap_gemm_asm_geglu_fwd / _geglu_bwd / _bias_gelu /
        # _gelu_bwd): same 2MNK, timed and re-run in isolation like the others (their epilogue work is inside the measured time)
        self.fused_orig = {}
        def wrap(name, dims):
            orig = getattr(self.hip, name)
            self.fused_orig[name] = orig
            def fn(*args, **kw):
                if not self.enabled:
                    return orig(*args, **kw)
                M, N, K = dims(*args)
                s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig(*args, **kw)
                e.record()
                if torch.cuda.current_stream().cuda_stream != self.main_stream:
                    self.second.append(2.0 * M * N * K); self.second_ev.append((s, e))
                else:
                    self.records.append((s, e, 2.0 * M * N * K))
                    sig = ("fused", name, M, N, K, args[0].stride(0), args[2].dtype if name == "linear_wgrad_sumsq" else None)
                    self.shapes[sig] = self.shapes.get(sig, 0) + 1
                return r
            setattr(self.hip, name, fn)
        wrap("linear_wgrad_sumsq", lambda dy, x, out, ss: (dy.shape[1], x.shape[1], dy.shape[0]))
        wrap("linear_geglu_train", lambda x, w, *a: (x.shape[0], w.shape[0], x.shape[1]))
        wrap("linear_dgrad_geglu_bwd", lambda dy, w, gu: (dy.shape[0], w.shape[1], w.shape[0]))
        wrap("linear_bias_gelu_train", lambda x, w, b: (x.shape[0], w.shape[0], x.shape[1]))
        wrap("linear_dgrad_gelu_bwd", lambda dy, w, h: (dy.shape[0], w.shape[1], w.shape[0]))
        self.main_stream = torch.cuda.current_stream().cuda_stream
        self.second, self.second_ev = [], []
        self.base = None
        self.shapes = {}

    def start(self):
        self.base = torch.cuda.Event(enable_timing=True)
        self.base.record()
        self.enabled = True

    def summary(self):
        t = sum(s.elapsed_time(e) for s, e, _ in self.records) * 1e-3
        fl = sum(f for _, _, f in self.records)
        return len(self.records), t, fl

    def isolated(self, dev, reps: int = 4):
        """The contention-free rate of the family: every distinct call signature the compute stream issued during the timed steps
        is re-run ALONE (fresh random operands, `reps` timed launches after two warm-up launches, HIP events on the launch
        stream, nothing else on the device) and the step's mix is priced with those durations:
            achieved = sum_shapes count x 2MNK / sum_shapes count x t_isolated.
        In the step itself the launches share the chip with the optimizer stream, the action expert's stream and the third
        (weight-gradient) stream — overlap that shortens the step but lengthens every event-timed launch; that figure is
        reported next to this one as `in_situ_event_timed`."""
        rnd = lambda *sh: (torch.rand(*sh, device=dev) * 2 - 1).to(torch.bfloat16)
        tot_t = tot_f = 0.0
        rows = []
        for sig, cnt in self.shapes.items():
            if sig[0] == "fused":
                _, name, M, N, K, lda, odt = sig
                fn = self.fused_orig[name]
                if name == "linear_wgrad_sumsq":     # (lda = the row stride of dy: the engine pads d(gate|up); odt: f32 or bf16 gradient buffer)
                    dy, x = rnd(K, lda)[:, :M], rnd(K, N)
                    o, acc = torch.empty(M, N, device=dev, dtype=odt), torch.zeros(1, device=dev)
                    call, label = (lambda: fn(dy, x, o, acc)), f"TN {M}x{N}x{K}"
                elif name == "linear_geglu_train":
                    x, w = rnd(M, lda)[:, :K], rnd(N, K) * 0.05
                    call, label = (lambda: fn(x, w)), f"NT+GeGLU {M}x{N}x{K}"
                elif name == "linear_dgrad_geglu_bwd":
                    dy, w = rnd(M, lda)[:, :K], rnd(K, N)
                    gu = (rnd(M, 2 * N + 64) * 4)[:, :2 * N]
                    call, label = (lambda: fn(dy, w, gu)), f"NN+dGeGLU {M}x{N}x{K}"
                elif name == "linear_bias_gelu_train":
                    x, w, b = rnd(M, lda)[:, :K], rnd(N, K) * 0.05, torch.randn(N, device=dev)
                    call, label = (lambda: fn(x, w, b)), f"NT+bias+GELU {M}x{N}x{K}"
                else:
                    dy, w, h = rnd(M, lda)[:, :K], rnd(K, N), rnd(M, N) * 4
                    call, label = (lambda: fn(dy, w, h)), f"NN+dGELU {M}x{N}x{K}"
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e-3 / reps
                f = 2.0 * M * N * K
                tot_t += cnt * t
                tot_f += cnt * f
                rows.append((cnt * t, label, cnt, t, f / t / 1e12))
                continue
            M, N, K, lda, ldb, ldc, a_kc, b_kc, bias_dt, has_res, ldr, gelu, accum, odt, tile, ksplit = sig
            a = rnd(M if a_kc else K, lda)
            b = rnd(N if b_kc else K, ldb)
            out = torch.zeros(M, ldc, dtype=odt, device=dev)
            bias = None if bias_dt is None else torch.randn(N, device=dev).to(bias_dt)
            res = rnd(M, ldr) if has_res else None
            call = lambda: self.orig(a, b, out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, a_kc=a_kc, b_kc=b_kc, bias=bias, residual=res,
                                     ldr=ldr, gelu=gelu, accum=accum, tile=tile, ksplit=ksplit)
            for _ in range(2):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / reps
            f = 2.0 * M * N * K
            tot_t += cnt * t
            tot_f += cnt * f
            rows.append((cnt * t, f"{'NT'[0] if a_kc else 'T'}{'T' if b_kc else 'N'} {M}x{N}x{K}", cnt, t, f / t / 1e12))
            del a, b, out, bias, res
        rows.sort(reverse=True)
        return tot_f, tot_t, rows

    def union_seconds(self):
        """Length of the union of all GEMM launch intervals, both streams: the time during which the family had a launch in
        flight.  Equal to the sum of the durations when nothing overlaps; the informative figure when launches do."""
        iv = sorted((self.base.elapsed_time(s), self.base.elapsed_time(e)) for s, e in
                    [(s, e) for s, e, _ in self.records] + self.second_ev)
        tot, cur_a, cur_b = 0.0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    tot += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            tot += cur_b - cur_a
        return tot * 1e-3


def cpu_baseline(cores: int):
    """CPU oracle (oracle/lap_oracle.py: f32, torch CPU autograd; kind "port" — the reference's JAX-CPU path cannot be installed
    offline) on the host cores: ONE full train-step forward + backward of the real LAP-3B (27 SigLIP blocks, 18 joint Gemma
    layers, 257,152-word vocabulary) at batch 1 on the benchmark's shapes — the smallest whole unit of the metric, no
    extrapolation.  Hosts with too little memory for the f32 parameters + gradients (27 GB + activations)
    fall back to a 6-layer slice extrapolated by the forward-FLOP ratio, and say so."""
    from oracle import lap_oracle as O

    torch.set_num_threads(cores)
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:   # noqa: BLE001
        free_gb = 0.0
    Tp, S, L = 560, 50, 48
    w_sig_l, w_vlm_l, w_exp_l = 412.4e6 / 27, 1.982e9 / 18, 0.311e9 / 18
    f_full = 2.792e12

    def run(oc, B, V):
        P = O.init_params(oc, 0)
        g = torch.Generator().manual_seed(0)
        la = torch.zeros(B, L, dtype=torch.bool); la[:, L - 16:] = True
        obs = dict(images={k: torch.rand(B, 224, 224, 3, generator=g) * 2 - 1 for k in oc.image_keys},
                   image_masks={k: torch.ones(B, dtype=torch.bool) for k in oc.image_keys},
                   tokenized_prompt=torch.randint(0, V, (B, L), generator=g), tokenized_prompt_mask=torch.ones(B, L, dtype=torch.bool),
                   tokenized_langact_mask=la, token_loss_mask=torch.ones(B, L, dtype=torch.bool), sample_mask=torch.ones(B, dtype=torch.bool))
        actions = torch.randn(B, 50, 7, generator=g); noise = torch.randn(B, 50, 7, generator=g); t = torch.rand(B, generator=g) * 0.999 + 0.001
        Pg = {k: v.requires_grad_(True) for k, v in P.items()}
        t0 = time.perf_counter()
        loss, _ = O.compute_loss(Pg, oc, obs, actions, noise, t)
        loss.backward()
        dt = time.perf_counter() - t0
        # the serving half of the metric on the same host cores: one batch-1 action chunk (prefix prefill + KV cache + 10 Euler
        # steps of the action expert, lap.py:605-675) on the oracle, f32, no autograd
        for v in Pg.values():
            v.grad = None
        so = {k: ({kk: vv[:1] for kk, vv in v.items()} if isinstance(v, dict) else v[:1]) for k, v in obs.items() if k != "tokenized_langact_mask"}
        with torch.no_grad():
            t0 = time.perf_counter()
            O.sample_actions(P, oc, so, noise[:1], num_steps=10)
            dts = time.perf_counter() - t0
        return dt, dts

    if free_gb >= 96:
        oc = O.OracleCfg(paligemma_variant="gemma_2b", action_expert_variant="gemma_300m", siglip_variant="So400m/14",
                         action_horizon=50, max_token_len=L, vocab_size=257152, language_loss_weight=0.4)
        dt, dts = run(oc, 1, 257152)
        return {"value": round(1.0 / dt, 6), "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"oracle f32 forward + backward (torch CPU autograd) of the full LAP-3B at batch 1, benchmark shapes: one sample in {dt:.1f} s on "
                          f"{cores} threads (optimizer update not included: < 1 % of the step's FLOPs); stand-in for the reference's JAX-CPU path, "
                          f"which cannot be installed offline"}, \
               {"value": round(dts * 1e3, 1), "unit": "ms per batch-1 action chunk", "cores": cores, "kind": "port",
                "sample": f"oracle f32 sample_actions of the full LAP-3B at batch 1 (SigLIP + prefix prefill + 10 denoise steps against the cached "
                          f"prefix keys / values): one chunk in {dts:.2f} s on {cores} threads; stand-in for the reference's JAX-CPU path"}
    NL, B, V = 6, 2, 16384
    O.GEMMA["gemma_2b_slice"] = O.GemmaCfg(2048, NL, 16384, 8, 1, 256)
    O.GEMMA["gemma_300m_slice"] = O.GemmaCfg(1024, NL, 4096, 8, 1, 256)
    O.SIGLIP["So400m/14_slice"] = O.SiglipCfg(1152, NL, 4304, 16)
    oc = O.OracleCfg(paligemma_variant="gemma_2b_slice", action_expert_variant="gemma_300m_slice", siglip_variant="So400m/14_slice",
                     action_horizon=50, max_token_len=L, vocab_size=V, language_loss_weight=0.4)
    dt, dts = run(oc, B, V)
    f_slice = 2 * (2 * 256 * NL * w_sig_l + NL * 4 * 256 ** 2 * 1152) + 2 * Tp * NL * w_vlm_l + NL * 4 * Tp ** 2 * 2048 \
        + 2 * S * NL * w_exp_l + NL * 4 * S * (Tp + S) * 2048 + 2 * 47 * 2048 * V
    return {"value": round(B / (dt * f_full / f_slice), 6), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"host has {free_gb:.0f} GB free (< 96): oracle f32 fwd+bwd, batch {B}, LAP-3B widths, {NL}/18 Gemma layers + {NL}/27 SigLIP blocks "
                      f"+ 16k-row vocab slice ({dt:.1f} s), extrapolated by forward-FLOP ratio {f_full / f_slice:.1f}x; stand-in for the reference's "
                      f"JAX-CPU path, which cannot be installed offline"}, \
           {"value": round(dts * f_full / f_slice * 1e3, 1), "unit": "ms per batch-1 action chunk", "cores": cores, "kind": "port",
            "sample": f"host has {free_gb:.0f} GB free (< 96): oracle f32 sample_actions at batch 1 on the {NL}-layer slice ({dts:.2f} s), extrapolated by the "
                      f"forward-FLOP ratio {f_full / f_slice:.1f}x"}


def serve_latency(cfg, dev, reps: int = 20):
    """BASELINE.json config[3]: LAP-3B batch-1 action chunk = SigLIP + VLM prefix prefill (KV cache kept in HBM) + 10
    flow-matching denoise steps of the action expert, captured once into a HIP graph and replayed (serve.GraphedSampler).
    Times `reps` replays bracketed by device synchronisation; also checks the replay against the eager sampler."""
    from lap_amd.model import LAP
    from lap_amd.serve import GraphedSampler

    model = LAP(cfg, seed=0, device=dev, with_grads=False)
    gs = GraphedSampler(model, 1, 10)
    gen = torch.Generator(device="cpu").manual_seed(0)
    for k in gs.obs.images:
        gs.obs.images[k].copy_(torch.rand(1, cfg.image_size, cfg.image_size, 3, generator=gen) * 2 - 1)
    gs.obs.tokenized_prompt.copy_(torch.randint(0, cfg.vocab_size, gs.obs.tokenized_prompt.shape, generator=gen, dtype=torch.int32))
    gs.noise.copy_(torch.randn(1, cfg.action_horizon, cfg.action_dim, generator=gen))
    eager = model.sample_actions(0, gs.obs, num_steps=10, noise=gs.noise).clone()
    gs.capture()
    gs.graph.replay()
    torch.cuda.synchronize()
    if model.serve_chain_failed():     # a block of the one-launch denoise step was not scheduled in time (shared GPU): separate launches
        model.disable_serve_chain()
        eager = model.sample_actions(0, gs.obs, num_steps=10, noise=gs.noise).clone()
        gs.capture()
    for _ in range(3):
        gs.graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gs.graph.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # one request at a time (what a control loop sees): every replay waits for the previous chunk, so the graph launch's host
    # side is not hidden behind the previous replay's kernels
    t0 = time.perf_counter()
    for _ in range(reps):
        gs.graph.replay()
        torch.cuda.synchronize()
    ms_one = (time.perf_counter() - t0) / reps * 1e3
    same = bool(torch.equal(eager, gs.out))
    model_chain = model.serve_chain and model._chain_ctr is not None and not model.serve_chain_failed()
    del gs, model
    torch.cuda.empty_cache()
    gbps = SERVE_BYTES / (ms * 1e-3) / 1e9
    return {"metric": "batch-1 action-chunk ms LAP-3B bf16 (prefix prefill + 10 denoise steps)", "ms_per_chunk": round(ms, 3),
            "ms_per_chunk_one_at_a_time": round(ms_one, 3), "denoise_layers_in_one_launch": bool(model_chain),
            "budget_ms": 10.0, "hbm_floor_ms": round(SERVE_BYTES / (HBM_PEAK_GBPS * 1e9) * 1e3, 2),
            "hbm_floor_ms_at_measured_copy_bw": round(SERVE_BYTES / 6.29e12 * 1e3, 2), "algorithmic_GB": round(SERVE_BYTES / 1e9, 2),
            "achieved_GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4), "graph": True,
            "graph_equals_eager": same, "replays": reps, "prompt_len": cfg.max_token_len, "action_horizon": cfg.action_horizon}


def _self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one rank per GPU under
    torch.distributed.run (rendezvous on 127.0.0.1, a free port) and pass the JSON line through."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    # Kernel arguments in device memory: ~1 ms per train step less launch latency on MI300-class parts (interleaved A/B: 310.4 ->
    # 309.3 ms; the replayed serving graph is unaffected).  A HIP runtime setting, read when the runtime starts (nothing has
    # touched the device yet); an explicit value in the environment wins.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--config", default="lap_bench")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serve", action="store_true", help="skip the batch-1 action-chunk latency leg (N = 1 only)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="bf16 = the headline metric; fp8 = BASELINE.json config 5 (fp8 forward / data-gradient GEMMs of the VLM "
                         "expert, everything else bf16) — reported with dtype 'fp8', a NON-headline line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for flow tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = local % torch.cuda.device_count()   # (flow tests run several ranks on one GPU with --backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import dataclasses

    from lap_amd import hip
    from lap_amd.config import get_config
    from lap_amd.train import TrainingStepRunner, init_train_state

    comm = None
    ranks_seen = devices_seen = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        # What the communicator itself saw, not the launcher's WORLD_SIZE: every rank contributes a 1 and a one-hot of its
        # device's PCI bus id slot through the backend's own all-reduce (RCCL for "nccl": the tensors live on the device).
        one = torch.ones(1, dtype=torch.float32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        uuid = torch.zeros(world, dtype=torch.int64, device=one.device)
        try:
            import zlib
            uuid[rank] = zlib.crc32(str(torch.cuda.get_device_properties(dev).uuid).encode()) + 1     # (str hashes are salted per process)
        except Exception:   # noqa: BLE001
            uuid[rank] = dev.index if hasattr(dev, "index") and dev.index is not None else rank
        dist.all_reduce(uuid)
        devices_seen = len(set(uuid.tolist()))
    tc = dataclasses.replace(get_config(args.config), batch_size=args.batch * world, fsdp_devices=world, gemm_dtype=args.dtype)
    state = init_train_state(tc, device=dev, world_size=world, rank=rank, use_fsdp=world > 1)
    runner = TrainingStepRunner(tc)
    batches = [synthetic_batch(tc.model, args.batch, dev, seed=1000 * rank + i) for i in range(2)]
    meter = GemmMeter(hip)
    meter.install()

    def sync():   # barrier + device synchronize (all streams: compute, optimizer / collective side stream)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        state, info = runner(0, state, batches[i % 2], state.step)
    sync()
    # The timed steps run WITHOUT the meter (772 event pairs per step are measurement overhead inside the
    # headline number): `enabled` is False, the wrappers pass straight through.
    t0 = time.perf_counter()
    for i in range(args.steps):
        state, info = runner(0, state, batches[i % 2], state.step)
    sync()
    dt = time.perf_counter() - t0
    loss_t = info["loss"]
    # ... and the in-situ figure and the step's call signatures come from a separate, short, event-timed pass behind them
    METER_STEPS = 2
    meter.start()
    for i in range(METER_STEPS):
        state, info = runner(0, state, batches[i % 2], state.step)
    sync()
    meter.enabled = False
    info = {"loss": loss_t}
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    loss = info["loss"].item()
    if rank == 0:
        n_launch, t_gemm, fl_gemm = meter.summary()
        GEMM_TRAFFIC = gemm_traffic()
        samples = args.batch * world * args.steps
        value = samples / dt
        in_situ = fl_gemm / t_gemm / 1e12 if t_gemm > 0 else 0.0
        union_s = meter.union_seconds()
        second_fl = sum(meter.second)
        # the family's rate by isolated per-shape timing (see GemmMeter.isolated): the train state stays resident, operands are fresh
        iso_f, iso_t, iso_rows = meter.isolated(dev)
        achieved = iso_f / iso_t / 1e12 if iso_t > 0 else 0.0
        msteps = METER_STEPS        # the metered pass, not the timed steps, is what the per-step figures below are divided by
        if os.environ.get("LAP_BENCH_SHAPES"):      # the whole table, for tools / profiles (stderr: stdout carries the one JSON line)
            for tt, nm, c, t, r in iso_rows:
                print(f"  {nm:28s} x{c // msteps:3d}  {t * 1e6:8.1f} us  {r:7.0f} TF/s  {tt / msteps * 1e3:7.2f} ms/step", file=sys.stderr)
        out = {
            "metric": "train-step samples/sec LAP-3B bf16", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("LAP-3B (SigLIP So400m/14 + Gemma-2B + Gemma-300M expert) full train step fwd+bwd+AdamW+EMA, "
                                    "2x224x224 images + 48-token prompt + 50-step action chunk, random-init weights")
                       if args.config == "lap_bench" and args.dtype == "bf16" else
                       ("NON-HEADLINE: BASELINE.json config 5 on one GPU — the same train step with fp8 (e4m3, per-tensor scaling) forward / "
                        "data-gradient GEMMs in the VLM expert, bf16 weight gradients and everything else" if args.config == "lap_bench"
                        else f"NON-HEADLINE flow test: config {args.config}"),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": 610,
                       "parallelism": f"fsdp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "kernel": "lap_gemm_asm_* (csrc/gemm_asm_kernels.s) / gemm_pq_kernel / gemm_sp_kernel / gemm_kernel (bf16 MFMA GEMM family, csrc/gemm.hip)", "achieved": round(in_situ, 1),
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(in_situ / MFMA_PEAK_TFLOPS, 4),
                         "definition": "IN THE STEP: sum of 2MNK over the compute stream's GEMM launches / sum of their durations, HIP events on the "
                                       "launch stream around every launch of a separate 2-step pass behind the timed steps (the timed steps carry no "
                                       "events); includes what the co-running optimizer / action-expert / weight-gradient streams cost each launch; "
                                       "GEMMs with a fused GeGLU / GELU epilogue count 2MNK and carry their epilogue in the duration.  This is the "
                                       "figure profiles/*_per_queue_step_breakdown.txt reproduces.",
                         "gemm_ms_per_step": round(t_gemm / msteps * 1e3, 2),
                         "isolated": {"achieved": round(achieved, 1), "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                                      "gemm_ms_per_step": round(iso_t / msteps * 1e3, 2),
                                      "note": "side figure: every distinct call signature of the step re-run ALONE in this process (fresh operands, HIP events), "
                                              "the step's mix priced with those durations: sum count x 2MNK / sum count x t_isolated"},
                         "distinct_shapes": len(iso_rows),
                         "plain_signatures_only": (lambda pl: {"achieved": round(sum(r * tt_ for tt_, _, _, _, r in pl) / max(sum(tt_ for tt_, *_ in pl), 1e-12), 1),
                                                               "ms_per_step": round(sum(tt_ for tt_, *_ in pl) / msteps * 1e3, 2),
                                                               "note": "the ISOLATED sum without the four fused-epilogue signatures (gate|up + GeGLU, down dgrad + GeGLU "
                                                                       "backward, SigLIP fc1 + GELU, fc2 dgrad + GELU backward), whose durations contain elementwise work "
                                                                       "that used to be separate HBM-bound kernels"})([x for x in iso_rows if "+" not in x[1]]),
                         "top_shapes_isolated": [{"shape": nm, "launches_per_step": c // msteps, "us": round(t * 1e6, 1), "TFLOPs": round(r, 0)}
                                        for _, nm, c, t, r in iso_rows[:8]],
                         "traffic": GEMM_TRAFFIC["bytes_per_launch"], "traffic_unit": "L2-to-fabric bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits included)",
                         "traffic_shape": GEMM_TRAFFIC["shape"], "traffic_algorithmic_bytes": GEMM_TRAFFIC["algorithmic_bytes_per_launch"],
                         "traffic_source": GEMM_TRAFFIC["source"], "traffic_measured_in_run": False,
                         "launches_per_step": n_launch // msteps,
                         "second_stream": {"launches_per_step": len(meter.second) // msteps,
                                           "flop_frac": round(second_fl / max(fl_gemm + second_fl, 1.0), 4),
                                           "note": "action-expert GEMMs co-running on a second HIP stream: counted, not in the timed sums"},
                         "union_of_launch_intervals": {"achieved": round((fl_gemm + second_fl) / max(union_s, 1e-9) / 1e12, 1),
                                                       "note": "all GEMM FLOPs of both streams / time during which any GEMM launch was in flight (informative)"},
                         "avg_launch_us": round(iso_t / max(sum(c for _, _, c, _, _ in iso_rows), 1) * 1e6, 2),
                         "gemm_time_frac_of_step": round(t_gemm / msteps / (dt / args.steps), 4),
                         "step_mfu": round(value / world * TRAIN_FLOP_PER_SAMPLE / 1e12 / MFMA_PEAK_TFLOPS, 4)},
            "final_loss": round(loss, 5),
        }
        if world == 1 and not args.no_serve and args.config == "lap_bench" and args.dtype == "bf16":
            del state, runner, batches
            torch.cuda.empty_cache()
            out["serve"] = serve_latency(tc.model, dev)
        if world == 1 and not args.no_cpu_baseline:
            # 32 host threads: torch-CPU matmuls of this size stop scaling (and get slower) well beyond that (256 threads on
            # the GPU box took 166 s for a 6-layer slice that 16 threads finish in 9 s)
            out["cpu_baseline"], serve_cpu = cpu_baseline(min(os.cpu_count() or 1, 32))
            if "serve" in out:
                out["serve"]["cpu_baseline"] = serve_cpu
        if world > 1:
            try:
                ver = torch.cuda.nccl.version()
            except Exception:   # noqa: BLE001
                ver = None
            out["rccl"] = {"ranks": ranks_seen, "ranks_source": "all_reduce(SUM) of one 1 per rank on the process group's backend, device tensor",
                           "devices_seen": devices_seen, "group_world_size": torch.distributed.get_world_size(),
                           "backend": torch.distributed.get_backend(), "version": ".".join(str(v) for v in ver) if isinstance(ver, tuple) else ver}
        # every LAP_* / RCCL / NCCL switch that steers the step and is set in this process's environment (a bench line is
        # only comparable with another one taken under the same switches)
        out["env_switches"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("LAP_", "NCCL_", "RCCL_", "HSA_", "HIP_", "GPU_MAX_HW"))}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
