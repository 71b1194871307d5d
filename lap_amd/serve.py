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
        self._load(obs, noise)
        self.graph.replay()
        return self.out


class Policy:
    """Model-level stand-in for openpi's Policy: `infer(obs)` with un-batched numpy inputs -> {"actions", "policy_timing"}."""

    def __init__(self, model: LAP, *, num_steps: int = 10, use_graph: bool = True, seed: int = 0, metadata: dict | None = None):
        self.model = model
        self.metadata = metadata or {}
        self.num_steps = num_steps
        self._sampler = GraphedSampler(model, 1, num_steps) if use_graph else None
        self._gen = torch.Generator(device=model.device).manual_seed(seed)

    def infer(self, obs: dict) -> dict:
        t0 = time.perf_counter()
        dev = self.model.device
        batched = {k: ({kk: np.asarray(vv)[None] for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)[None])
                   for k, v in obs.items() if v is not None}
        o = CoTObservation.from_dict(batched, device=dev)
        cfg = self.model.config
        noise = torch.randn((1, cfg.action_horizon, cfg.action_dim), generator=self._gen, device=dev)
        if self._sampler is not None:
            a = self._sampler(o, noise)
        else:
            a = self.model.sample_actions(0, o, num_steps=self.num_steps, noise=noise)
        actions = a[0].cpu().numpy()  # device sync
        return {"actions": actions, "policy_timing": {"infer_ms": (time.perf_counter() - t0) * 1e3}}


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
        dev = self._base.model.device
        raw_state = np.array(obs["state"], copy=True) if obs.get("state") is not None else None
        batched = {k: ({kk: np.asarray(vv)[None] for kk, vv in v.items()} if isinstance(v, dict) else np.asarray(v)[None])
                   for k, v in obs.items() if v is not None}
        o = CoTObservation.from_dict(batched, device=dev)
        self._calls += 1
        tokens = self._base.model.sample_tokens(self._calls, o, **self._sample_kwargs)
        out = {"state": batched.get("state"), "tokens": tokens.cpu().numpy(), "raw_state": raw_state}
        out["policy_timing"] = {"infer_ms": (time.perf_counter() - t0) * 1e3}
        return out

    def infer(self, obs: dict, *, noise=None) -> dict:
        return self.infer_reasoning(obs)

    def vqa_infer(self, obs: dict) -> dict:
        return self.infer_reasoning(obs)
