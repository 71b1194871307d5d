"""Batch-1 action-chunk serving: hipGraph-captured `sample_actions` and a minimal `Policy.infer` surface.

Reference: scripts/serve_policy.py:69-107 -> policy_config_adapter.create_trained_policy (85-154) ->
openpi Policy.infer [UPSTREAM-RECALL]: add batch dim -> Observation.from_dict -> jit(sample_actions) -> strip batch
-> {"actions", "policy_timing": {"infer_ms"}}.  The request-level transforms (CoTInputs, Normalize, tokenizer,
Unnormalize, CoTOutputs) are SURVEY.md §8(f) rank-1 "next" items; this module takes model-level inputs.

The reference's jit turns the whole sampler (prefix prefill + lax.while_loop of 10 denoise steps, lap.py:605-675)
into one XLA executable.  Here the same region is captured once into a HIP graph (torch.cuda.CUDAGraph is the
capture plumbing; every node is one of our C-ABI kernel launches) and replayed per request: ~2,800 kernel launches
cost one graph launch on the host.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from lap_amd.model import LAP
from lap_amd.observation import CoTObservation


class GraphedSampler:
    def __init__(self, model: LAP, batch_size: int = 1, num_steps: int = 10, prompt_len: int | None = None):
        self.model, self.B, self.num_steps = model, batch_size, num_steps
        cfg = model.config
        dev = model.device
        L = prompt_len or cfg.max_token_len
        H = cfg.image_size
        self.obs = CoTObservation(
            images={k: torch.zeros(batch_size, H, H, 3, device=dev) for k in cfg.image_keys},
            image_masks={k: torch.ones(batch_size, dtype=torch.bool, device=dev) for k in cfg.image_keys},
            state=torch.zeros(batch_size, cfg.action_dim, device=dev),
            tokenized_prompt=torch.zeros(batch_size, L, dtype=torch.int32, device=dev),
            tokenized_prompt_mask=torch.ones(batch_size, L, dtype=torch.bool, device=dev))
        self.noise = torch.zeros(batch_size, cfg.action_horizon, cfg.action_dim, device=dev)
        self.graph = None
        self.out = None

    def _load(self, obs: CoTObservation, noise: torch.Tensor):
        for k in self.obs.images:
            self.obs.images[k].copy_(obs.images[k])
            if k in obs.image_masks and obs.image_masks[k] is not None:
                self.obs.image_masks[k].copy_(obs.image_masks[k])
        self.obs.tokenized_prompt.copy_(obs.tokenized_prompt)
        self.obs.tokenized_prompt_mask.copy_(obs.tokenized_prompt_mask)
        self.noise.copy_(noise)

    def capture(self):
        side = torch.cuda.Stream(device=self.model.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture (kernel attribute setup, allocator pools)
            for _ in range(2):
                self.model.sample_actions(0, self.obs, num_steps=self.num_steps, noise=self.noise)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.model.sample_actions(0, self.obs, num_steps=self.num_steps, noise=self.noise)
        return self

    def __call__(self, obs: CoTObservation, noise: torch.Tensor) -> torch.Tensor:
        if self.graph is None:
            self.capture()
        # parameter-derived caches the graph holds addresses of (adaRMS modulations, packed expert weights) follow the
        # parameters in place; a no-op unless a train step / restore / reload happened since the last call
        self.model.refresh_serve_caches(self.num_steps)
        self._load(obs, noise)
        self.graph.replay()
        return self.out


class Policy:
    """openpi's Policy surface: `infer(obs)` -> {"actions", ..., "policy_timing"}.

    With `transforms` / `output_transforms` (lap_amd/policy_io.py, assembled by `create_trained_policy`) `obs` is the raw
    client request ({"observation": {image keys, "state"}, "prompt", ...}) and the result carries un-normalised actions, as
    in policies/policy_config_adapter.py:139-153.  Without them `obs` is already model-level (un-batched numpy arrays)."""

    def __init__(self, model: LAP, *, num_steps: int = 10, use_graph: bool = True, seed: int = 0, metadata: dict | None = None,
                 transforms=(), output_transforms=(), sample_kwargs: dict | None = None):
        from lap_amd.policy_io import compose

        self.model = model
        self.metadata = metadata or {}
        self._sample_kwargs = dict(sample_kwargs or {})
        self.num_steps = self._sample_kwargs.pop("num_steps", num_steps)
        # (pi0, `pi05=False`, runs on the generic layer loop with its suffix re-embedded by torch ops at every step: served eagerly)
        self._sampler = GraphedSampler(model, 1, self.num_steps) if (use_graph and model.config.pi05) else None
        self._gen = torch.Generator(device=model.device).manual_seed(seed)
        self._has_transforms = bool(transforms) or bool(output_transforms)
        self._transforms = list(transforms)
        self._input_transform = compose(self._transforms)
        self._output_transform = compose(list(output_transforms))

    def _to_observation(self, inputs: dict) -> tuple[CoTObservation, dict]:
        batched = {k: ({kk: np.asarray(vv)[None] for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)[None])
                   for k, v in inputs.items() if v is not None and not isinstance(v, str)}
        return CoTObservation.from_dict(batched, device=self.model.device), batched

    def infer(self, obs: dict, *, noise=None) -> dict:
        t0 = time.perf_counter()
        dev = self.model.device
        inputs = self._input_transform(dict(obs))
        o, batched = self._to_observation(inputs)
        cfg = self.model.config
        if noise is None:
            noise = torch.randn((1, cfg.action_horizon, cfg.action_dim), generator=self._gen, device=dev)
        else:
            noise = torch.as_tensor(np.asarray(noise), dtype=torch.float32, device=dev).reshape(1, cfg.action_horizon, cfg.action_dim)
        if self._sampler is not None and self._graph_compatible(o):
            a = self._sampler(o, noise)
        else:
            a = self.model.sample_actions(0, o, num_steps=self.num_steps, noise=noise)
        actions = a[0].cpu().numpy()  # device sync
        if self.model.serve_chain_failed():
            # A block of the one-launch denoise step (csrc/serve_chain.hpp) was not scheduled within its bounded wait — the GPU is
            # shared with work that holds compute units — so this chunk is invalid: go back to the separate launches for good.
            self.model.disable_serve_chain()
            if self._sampler is not None:
                self._sampler.graph = None
            return self.infer(obs, noise=noise.cpu().numpy())
        model_ms = (time.perf_counter() - t0) * 1e3
        if not self._has_transforms:
            return {"actions": actions, "policy_timing": {"infer_ms": model_ms}}
        outputs = self._output_transform({"state": batched["state"][0], "actions": actions})
        outputs["policy_timing"] = {"infer_ms": model_ms}
        return outputs

    def _graph_compatible(self, o: CoTObservation) -> bool:
        """The captured graph is tied to the model's native image resolution and prompt length."""
        g = self._sampler.obs
        return (all(tuple(o.images[k].shape) == tuple(g.images[k].shape) for k in g.images)
                and tuple(o.tokenized_prompt.shape) == tuple(g.tokenized_prompt.shape))


def create_trained_policy(train_config, checkpoint_dir, *, tokenizer_model_path=None, tokenizer=None, repack_transforms=(),
                          sample_kwargs: dict | None = None, default_prompt: str | None = None, norm_stats: dict | None = None,
                          device="cuda", use_graph: bool = True, deterministic: bool = False) -> Policy:
    """policies/policy_config_adapter.py:85-154: model from `<checkpoint_dir>/params`, norm stats from
    `<checkpoint_dir>/assets/<asset_id>/norm_stats.json`, and the standard transform stack
        [repack.inputs, InjectDefaultPrompt, CoTInputs (the data config's), Normalize, InjectDefaultPrompt, Tokenize, PadStatesAndActions]
        -> model -> [DetokenizeReasoning, Unnormalize, CoTOutputs, repack.outputs]  (VLA-0: [DetokenizeReasoning, CoTOutputs(norm stats)]).
    `repack_transforms` = (inputs, outputs) lists.  The PaliGemma SentencePiece model cannot be downloaded here: pass
    `tokenizer_model_path` (or a ready `tokenizer`).
    `deterministic` (not in the reference, which has no serving override): True switches off the training-time randomness the data /
    model config carries into the serving stack — wrist-image dropout, random un-masking of missing cameras, state dropout — so that
    equal requests give equal prompts and masks.  False keeps the reference's wiring and warns when any of the three is non-zero."""
    import pathlib

    from lap_amd import checkpoints as _ckpt
    from lap_amd import policy_io as pio

    checkpoint_dir = pathlib.Path(checkpoint_dir)
    mc = train_config.model
    model = mc.load(_ckpt.restore_params(checkpoint_dir), device=device, with_grads=False)
    if norm_stats is None:
        if getattr(train_config.data, "asset_id", None) is None:
            raise ValueError("Asset id is required to load norm stats.")
        norm_stats = _ckpt.load_norm_stats(checkpoint_dir / "assets")
    ntype = getattr(train_config.data, "action_proprio_normalization_type", "bounds_q99")
    if tokenizer is None:
        tokenizer = pio.PaligemmaTokenizer(tokenizer_model_path, mc.max_token_len, prompt_format=mc.prompt_format,
                                           reasoning_mask_prob=0.0)
    rin, rout = (list(repack_transforms[0]), list(repack_transforms[1])) if repack_transforms else ([], [])
    import dataclasses as _dc
    import warnings

    from lap_amd.data import data_transform_inputs
    dc = train_config.data
    state_dropout = getattr(mc, "state_dropout", 0.0)
    noisy = {k: v for k, v in (("data.wrist_image_dropout_prob", getattr(dc, "wrist_image_dropout_prob", 0.0)),
                               ("data.random_mask_prob", getattr(dc, "random_mask_prob", 0.0)), ("model.state_dropout", state_dropout)) if v}
    if deterministic:
        if _dc.is_dataclass(dc):
            dc = _dc.replace(dc, **{k: 0.0 for k in ("wrist_image_dropout_prob", "random_mask_prob") if hasattr(dc, k)})
        state_dropout = 0.0
    elif noisy:
        warnings.warn("create_trained_policy: the serving stack keeps the config's training-time randomness as the reference does "
                      f"({', '.join(f'{k}={v}' for k, v in noisy.items())}): images are blanked / un-masked and the state dropped at random "
                      "per request.  Pass deterministic=True (or zero these fields) for a deterministic server.", stacklevel=2)
    strategy = getattr(dc, "transform_strategy", "standard")
    fmt_name = getattr(dc, "language_action_format_name", "verbose_eef_with_rotation")
    # policy_config_adapter.py:137-150: the data config's OWN data-transform group serves too — the same `CoTInputs` as in training,
    # training-time image randomness included (wrist dropout / random un-masking are whatever the data config says: the reference has no
    # serving override; set them to 0 in the config a server is started with) — then Normalize, then the model transforms
    # (training/config.py:195-207: default prompt, tokenizer with the model's state dropout, padding).
    transforms = [*rin, pio.InjectDefaultPrompt(default_prompt), data_transform_inputs(dc, mc),
                  pio.Normalize(norm_stats, normalization_type=ntype), pio.InjectDefaultPrompt(None),
                  pio.TokenizePromptAndReasoning(tokenizer, discrete_state_input=mc.discrete_state_input, verbose_mode=mc.verbose_mode,
                                                 state_dropout=state_dropout),
                  pio.PadStatesAndActions(mc.action_dim)]
    detok = pio.DetokenizeReasoning(tokenizer)      # model_transforms.outputs (include_outputs, training/config.py:192-194)
    if strategy == "vla0":      # OutputTransformAssembler._build_vla0_outputs (:44-66): the decoder un-normalises, no Unnormalize
        outputs = [detok, pio.CoTOutputs(language_action_format=fmt_name, norm_stats=norm_stats, normalization_type=ntype, transform_strategy="vla0"), *rout]
    else:                       # _build_standard_outputs (:68-82)
        outputs = [detok, pio.Unnormalize(norm_stats, normalization_type=ntype),
                   pio.CoTOutputs(language_action_format=fmt_name, transform_strategy=strategy), *rout]
    return Policy(model, transforms=transforms, output_transforms=outputs, sample_kwargs=sample_kwargs, use_graph=use_graph,
                  metadata=getattr(train_config, "policy_metadata", None))


class ARPolicy:
    """policies/policy_adapter.py:13-61: the autoregressive serving mode (LAP_AR).  `infer` returns the generated token
    ids of `LAP.sample_tokens` (+ the model-level state) instead of an action chunk; decoding them to text / actions is
    the job of the output transforms, which take exactly this dict."""

    def __init__(self, base: Policy, *, sample_kwargs: dict | None = None):
        if not hasattr(base.model, "sample_tokens"):
            raise AssertionError("Model must have a sample_tokens method")
        self._base = base
        self._sample_kwargs = dict(sample_kwargs or {})
        self._calls = 0

    def __getattr__(self, name):
        return getattr(self._base, name)

    def infer_reasoning(self, obs: dict) -> dict:
        t0 = time.perf_counter()
        base = self._base
        if base._has_transforms:   # raw client request (policy_adapter.py:25-31)
            raw_state = np.array(obs["observation"]["state"], copy=True)
            inputs = base._input_transform(dict(obs))
        else:
            raw_state = np.array(obs["state"], copy=True) if obs.get("state") is not None else None
            inputs = obs
        o, batched = base._to_observation(inputs)
        self._calls += 1
        tokens = base.model.sample_tokens(self._calls, o, **self._sample_kwargs)
        out = {"state": batched.get("state"), "tokens": tokens.cpu().numpy(), "raw_state": raw_state}
        model_ms = (time.perf_counter() - t0) * 1e3
        if base._has_transforms:
            out = base._output_transform(out)
        out["policy_timing"] = {"infer_ms": model_ms}
        return out

    def infer(self, obs: dict, *, noise=None) -> dict:
        return self.infer_reasoning(obs)

    def vqa_infer(self, obs: dict) -> dict:
        return self.infer_reasoning(obs)


def create_trained_policy_ar(*args, sample_kwargs: dict | None = None, language_action_format="verbose_eef_with_rotation", **kwargs) -> ARPolicy:
    """policy_config_adapter.py:157-160 with the AR output stack [DetokenizeReasoning, CoTOutputs(language_action_format)]:
    generated ids -> text -> [dx, dy, dz, droll, dpitch, dyaw, gripper] (the language action describes the whole chunk as
    one delta, output_transforms.py:75-104; `Unnormalize` has no `actions` statistics to apply to such deltas and is left out
    as in the reference's standard stack it would act on the normalised `state` only)."""
    from lap_amd import policy_io as pio

    base = create_trained_policy(*args, use_graph=False, **kwargs)
    tok = next(t.tokenizer for t in base._transforms if isinstance(t, pio.TokenizePromptAndReasoning))
    base._output_transform = pio.compose([pio.DetokenizeReasoning(tok), pio.CoTOutputs(language_action_format=language_action_format)])
    return ARPolicy(base, sample_kwargs=sample_kwargs)
