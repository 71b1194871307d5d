"""Request-level transforms of the LAP policies: client observation dict -> model inputs, model outputs -> client actions
(SURVEY.md §8f rank 1).  Host-side numpy code; nothing here touches the GPU.

Reference call chain (policies/policy_config_adapter.py:85-154):
    repack -> InjectDefaultPrompt -> CoTInputs -> Normalize(norm_stats) -> [InjectDefaultPrompt,
    TokenizePromptAndReasoning, PadStatesAndActions]  ==> model ==>  [DetokenizeReasoning] -> Unnormalize -> CoTOutputs
Restated from: src/lap/transforms.py:27-275 (tokenize / normalize / unnormalize / pad), policies/transforms/
{input_transforms,image_handler,image_utils,text_utils,output_transforms}.py, models/tokenizer.py:105-331 and the
openpi pieces they lean on (InjectDefaultPrompt, pad_to_dim, apply_tree, NormStats, msgpack_numpy — absent submodule,
[UPSTREAM-RECALL] in SURVEY.md §8c).  Parity status: PINNED by outputs of the reference's own classes, executed in the
build container with stand-ins only for the openpi base classes they subclass (never for arithmetic; each generator's
docstring says what it stands in for): tests/golden/{tokenize_v1, cot_inputs_v1, cot_outputs_v1, normalize_v1,
prompt_format_v1, lang_action_v1, question_v1}.json, replayed by tests/test_policy_io_cpu.py / test_questions_cpu.py
(ids and masks exact, actions to 1e-12, the training-side normalisation bit for bit in float32).  Still recalled from
openpi, not pinned: InjectDefaultPrompt, pad_to_dim, the msgpack-numpy wire format.

Scope: the inference path of robot samples (what a LIBERO / DROID client sends) and the training path of CoTInputs —
language-action labels through `lang_actions.ActionProcessor`, wrist-image dropout / random un-masking, and the VQA /
prediction sample handlers (`questions.py`); text <-> delta conversion lives in lap_amd/lang_actions.py.
"""
from __future__ import annotations

import dataclasses
import pathlib
from typing import Any, Callable, Sequence

import numpy as np

from lap_amd import prompt as _prompt

IMAGE_KEYS = ("base_0_rgb", "left_wrist_0_rgb")   # models/model_adapter.py:17-21


# ------------------------------------------------------------------------------------------------ small helpers
def pad_to_dim(x, target_dim: int, axis: int = -1, value: float = 0.0) -> np.ndarray:
    """openpi.transforms.pad_to_dim: right-pads `axis` with `value` up to target_dim (longer inputs pass unchanged)."""
    x = np.asarray(x)
    cur = x.shape[axis]
    if cur >= target_dim:
        return x
    width = [(0, 0)] * x.ndim
    width[axis] = (0, target_dim - cur)
    return np.pad(x, width, constant_values=value)


def flatten_dict(tree: dict, sep: str = "/", _pre: str = "") -> dict:
    out = {}
    for k, v in tree.items():
        key = f"{_pre}{sep}{k}" if _pre else str(k)
        if isinstance(v, dict):
            out.update(flatten_dict(v, sep, key))
        else:
            out[key] = v
    return out


def unflatten_dict(flat: dict, sep: str = "/") -> dict:
    out: dict = {}
    for key, v in flat.items():
        node = out
        parts = key.split(sep)
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return out


@dataclasses.dataclass(frozen=True)
class NormStats:
    """openpi.shared.normalize.NormStats (norm_stats.json entries)."""
    mean: np.ndarray
    std: np.ndarray
    q01: np.ndarray | None = None
    q99: np.ndarray | None = None
    min: np.ndarray | None = None
    max: np.ndarray | None = None

    @classmethod
    def from_dict(cls, d) -> "NormStats":
        if isinstance(d, NormStats):
            return d
        arr = lambda k: None if d.get(k) is None else np.asarray(d[k], dtype=np.float64)
        return cls(mean=arr("mean"), std=arr("std"), q01=arr("q01"), q99=arr("q99"), min=arr("min"), max=arr("max"))


def as_norm_stats(tree) -> dict[str, NormStats] | None:
    """Accepts what `checkpoints.load_norm_stats` returns (nested dict of lists) or ready NormStats objects."""
    if tree is None:
        return None
    flat = {}
    for k, v in tree.items():
        if isinstance(v, NormStats) or (isinstance(v, dict) and "mean" in v and "std" in v):
            flat[k] = NormStats.from_dict(v)
        elif isinstance(v, dict):
            for kk, vv in as_norm_stats(v).items():
                flat[f"{k}/{kk}"] = vv
        elif hasattr(v, "mean") and hasattr(v, "std"):     # any record with the fields as attributes (openpi's NormStats, ExtendedNormStats)
            flat[k] = NormStats.from_dict({f: getattr(v, f, None) for f in ("mean", "std", "q01", "q99", "min", "max")})
        else:
            raise TypeError(f"norm stats entry {k!r} is neither a stats record nor a sub-tree")
    return flat


def _apply_tree(data: dict, stats: dict[str, NormStats], fn: Callable, strict: bool) -> dict:
    """openpi.transforms.apply_tree: `fn` on every leaf of `data` that has statistics; strict -> every statistic needs a leaf."""
    flat = flatten_dict(data)
    if strict:
        for k in stats:
            if k not in flat:
                raise ValueError(f"Selector key {k} not found in tree")
    return unflatten_dict({k: (fn(np.asarray(v), stats[k]) if k in stats else v) for k, v in flat.items()})


_NORM_TYPES = ("normal", "bounds", "bounds_q99")   # datasets/utils/helpers.py:32-37


def _norm_type(t) -> str:
    t = getattr(t, "value", t)
    if t not in _NORM_TYPES:
        raise ValueError(f"Unknown normalization type: {t}")
    return t


def _need_quantiles(stats: dict[str, NormStats]):
    for k, s in stats.items():
        if s.q01 is None or s.q99 is None:
            raise ValueError(f"quantile stats must be provided if use_quantile_norm is True. Key {k} is missing q01 or q99.")


# ------------------------------------------------------------------------------------------------ normalisation
@dataclasses.dataclass(frozen=True)
class Normalize:
    """transforms.py:151-217.  normal: (x - mean) / (std + 1e-6); bounds: [min, max] -> [-1, 1], clipped, constant
    dimensions -> 0; bounds_q99: [q01, q99] -> [-1, 1] (NOT clipped), constant dimensions -> 0.  Statistics longer than the
    data are cut to the data's last dimension."""
    norm_stats: Any
    normalization_type: Any = "normal"
    strict: bool = False

    def __post_init__(self):
        object.__setattr__(self, "norm_stats", as_norm_stats(self.norm_stats))
        object.__setattr__(self, "normalization_type", _norm_type(self.normalization_type))
        if self.norm_stats is not None and self.normalization_type == "bounds_q99":
            _need_quantiles(self.norm_stats)

    def __call__(self, data: dict) -> dict:
        if self.norm_stats is None:
            return data
        fn = {"normal": self._z, "bounds": self._bounds, "bounds_q99": self._quantile}[self.normalization_type]
        return _apply_tree(data, self.norm_stats, fn, self.strict)

    @staticmethod
    def _z(x, s: NormStats):
        d = x.shape[-1]
        return (x - s.mean[..., :d]) / (s.std[..., :d] + 1e-6)

    @staticmethod
    def _bounds(x, s: NormStats):
        assert s.min is not None and s.max is not None
        d = x.shape[-1]
        lo, hi = s.min[..., :d], s.max[..., :d]
        y = np.clip(2.0 * (x - lo) / (hi - lo + 1e-8) - 1.0, -1.0, 1.0)
        return np.where(np.equal(lo, hi), 0.0, y)

    @staticmethod
    def _quantile(x, s: NormStats):
        assert s.q01 is not None and s.q99 is not None
        d = x.shape[-1]
        lo, hi = s.q01[..., :d], s.q99[..., :d]
        y = (x - lo) / (hi - lo + 1e-6) * 2.0 - 1.0
        return np.where(np.equal(lo, hi), 0.0, y)


@dataclasses.dataclass(frozen=True)
class Unnormalize:
    """transforms.py:220-275.  Model outputs are wider than the statistics (action_dim 32 vs 7 recorded dimensions): normal
    pads mean with 0 / std with 1, bounds pads min with -1 / max with +1, bounds_q99 leaves the extra dimensions untouched."""
    norm_stats: Any
    normalization_type: Any = "normal"

    def __post_init__(self):
        object.__setattr__(self, "norm_stats", as_norm_stats(self.norm_stats))
        object.__setattr__(self, "normalization_type", _norm_type(self.normalization_type))
        if self.norm_stats is not None and self.normalization_type == "bounds_q99":
            _need_quantiles(self.norm_stats)

    def __call__(self, data: dict) -> dict:
        if self.norm_stats is None:
            return data
        fn = {"normal": self._z, "bounds": self._bounds, "bounds_q99": self._quantile}[self.normalization_type]
        return _apply_tree(data, self.norm_stats, fn, strict=False)

    @staticmethod
    def _z(x, s: NormStats):
        d = x.shape[-1]
        return x * (pad_to_dim(s.std, d, value=1.0) + 1e-6) + pad_to_dim(s.mean, d, value=0.0)

    @staticmethod
    def _bounds(x, s: NormStats):
        assert s.min is not None and s.max is not None
        d = x.shape[-1]
        lo, hi = pad_to_dim(s.min, d, value=-1.0), pad_to_dim(s.max, d, value=1.0)
        return (x + 1.0) / 2.0 * (hi - lo + 1e-8) + lo

    @staticmethod
    def _quantile(x, s: NormStats):
        assert s.q01 is not None and s.q99 is not None
        d = s.q01.shape[-1]
        if d < x.shape[-1]:
            head = (x[..., :d] + 1.0) / 2.0 * (s.q99 - s.q01 + 1e-6) + s.q01
            return np.concatenate([head, x[..., d:]], axis=-1)
        return (x + 1.0) / 2.0 * (s.q99 - s.q01 + 1e-6) + s.q01


@dataclasses.dataclass(frozen=True)
class NormalizeActionAndProprio:
    """transforms.py:292-444, numpy branch: the normalisation of the TRAINING data path — the mixer maps it over every robot
    dataset with the mixture's global statistics (dataset_mixer.py:334-359; VQA sets are skipped), and the train-time transform
    group has no `Normalize` (training/config.py:195-207).  It is NOT `Normalize`: float32 arithmetic; `bounds` and `bounds_q99`
    share one formula, 2 (x - lo) / (hi - lo + 1e-8) - 1 CLIPPED to [-1, 1] (q01 / q99 or min / max), constant dimensions -> 0,
    where the policy-side `Normalize` leaves quantile-normalised values unclipped with 1e-6.  Statistics: `{"actions" | "action":
    {...}, "state": {...}}` of dicts or objects; a missing group or field leaves that entry as it is (cast to float32).
    Here the sample is flat (`data[action_key]`, `data[state_key]`); the reference's trajectory keeps the state under
    `observation`."""
    norm_stats: Any
    normalization_type: Any = "normal"
    action_key: str = "actions"
    state_key: str = "state"

    def __post_init__(self):
        object.__setattr__(self, "normalization_type", _norm_type(self.normalization_type))

    @staticmethod
    def _group(root, name):
        if not isinstance(root, dict):
            return None
        g = root.get(name)
        return root.get(name[:-1]) if g is None and name.endswith("s") else g

    @staticmethod
    def _value(group, key):
        if group is None:
            return None
        v = group.get(key) if isinstance(group, dict) else getattr(group, key, None)
        return None if v is None else np.asarray(v, dtype=np.float32)

    def _one(self, x, group):
        x = np.asarray(x, dtype=np.float32)
        if group is not None and not isinstance(group, dict):
            group = {k: getattr(group, k, None) for k in ("mean", "std", "q01", "q99", "min", "max")}
        if group is not None:     # statistics wider than the data (a mixture's pooled state width) are cut like `Normalize` cuts them
            group = {k: (None if v is None else np.asarray(v, dtype=np.float32)[..., :x.shape[-1]]) for k, v in group.items()
                     if k in ("mean", "std", "q01", "q99", "min", "max")}
        if self.normalization_type == "normal":
            mean, std = self._value(group, "mean"), self._value(group, "std")
            return x if mean is None or std is None else (x - mean) / (std + 1e-6)
        lo_k, hi_k = ("min", "max") if self.normalization_type == "bounds" else ("q01", "q99")
        lo, hi = self._value(group, lo_k), self._value(group, hi_k)
        if lo is None or hi is None:
            return x
        y = np.clip(2.0 * (x - lo) / (hi - lo + 1e-8) - 1.0, -1.0, 1.0)
        return np.where(np.equal(lo, hi), 0.0, y)

    def __call__(self, data: dict) -> dict:
        if self.norm_stats is None:
            return data
        out = dict(data)
        out[self.action_key] = self._one(data[self.action_key], self._group(self.norm_stats, "actions"))
        if data.get(self.state_key) is not None:
            out[self.state_key] = self._one(data[self.state_key], self._group(self.norm_stats, "state"))
        return out


# ------------------------------------------------------------------------------------------------ inputs
@dataclasses.dataclass(frozen=True)
class InjectDefaultPrompt:
    """openpi.transforms.InjectDefaultPrompt: fills `prompt` when the request has none."""
    prompt: str | None

    def __call__(self, data: dict) -> dict:
        if self.prompt is not None and "prompt" not in data:
            data = {**data, "prompt": np.asarray(self.prompt)}
        return data


@dataclasses.dataclass(frozen=True)
class PadStatesAndActions:
    """openpi.transforms.PadStatesAndActions: zero-pads `state` (and `actions` when present) to the model's action_dim."""
    model_action_dim: int

    def __call__(self, data: dict) -> dict:
        data = dict(data)
        data["state"] = pad_to_dim(data["state"], self.model_action_dim, axis=-1)
        if "actions" in data:
            data["actions"] = pad_to_dim(data["actions"], self.model_action_dim, axis=-1)
        return data


def dataset_resize_with_pad(image: np.ndarray, height: int, width: int) -> np.ndarray:
    """datasets/utils/image_utils.py:192-228 (`make_decode_images_fn(resize_to=(224, 224))`): what the reference's DATA pipeline does
    to every decoded frame so that a mixture of datasets with different camera resolutions batches — aspect-preserving bilinear
    resize to floor(h / r) x floor(w / r), r = max(w / width, h / height) (tf.image.resize: half-pixel centres, NO anti-aliasing,
    unlike the model-side `observation.resize_with_pad`), rounded to uint8, centred zero padding (the odd pixel goes after)."""
    import torch

    img = np.asarray(image)
    h, w = img.shape[:2]
    if (h, w) == (height, width):
        return img
    ratio = np.float32(max(np.float32(w) / np.float32(width), np.float32(h) / np.float32(height)))
    rh, rw = int(np.floor(np.float32(h) / ratio)), int(np.floor(np.float32(w) / ratio))
    x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].to(torch.float32)
    x = torch.nn.functional.interpolate(x, size=(rh, rw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0)
    if img.dtype == np.uint8:
        res, fill = x.round().clamp(0, 255).to(torch.uint8).numpy(), 0
    else:
        res, fill = x.clamp(-1.0, 1.0).numpy().astype(img.dtype), -1.0
    out = np.full((height, width, img.shape[2]), fill, dtype=res.dtype)
    ph0, pw0 = (height - rh) // 2, (width - rw) // 2
    out[ph0:ph0 + rh, pw0:pw0 + rw] = res
    return out


@dataclasses.dataclass(frozen=True)
class ResizeImages:
    """The loader-side use of `dataset_resize_with_pad`: every image of the sample to (height, width); a no-op for frames that have
    the size already (stores exported at the model's resolution)."""
    height: int
    width: int

    def __call__(self, data: dict) -> dict:
        if "image" not in data:
            return data
        data = dict(data)
        data["image"] = {k: dataset_resize_with_pad(v, self.height, self.width) for k, v in data["image"].items()}
        return data


def parse_image(image):
    """image_utils.py:7-17: float images are [0, 1] -> uint8 (truncating), CHW / TCHW -> channels last."""
    if image is None:
        return None
    image = np.asarray(image)
    if np.issubdtype(image.dtype, np.floating):
        image = (255 * image).astype(np.uint8)
    if image.ndim == 3 and image.shape[0] == 3:
        image = np.transpose(image, (1, 2, 0))
    if image.ndim == 4 and image.shape[1] == 3:
        image = np.transpose(image, (0, 2, 3, 1))
    return image


def _text(value, default: str = "") -> str:
    """text_utils.py:7-22 (np.str_ is a str; 0-d string arrays are unwrapped first)."""
    if isinstance(value, np.ndarray) and value.ndim == 0:
        value = value.item()
    if isinstance(value, bytes):
        return value.decode("utf-8")
    return value if isinstance(value, str) else default


@dataclasses.dataclass(frozen=True)
class CoTInputs:
    """policies/transforms/input_transforms.py:22-249 for robot samples: base image (+ wrist images, zeros and mask False
    when absent or all-zero; image_handler.py:22-166 incl. the training-time wrist dropout / random un-masking), state, prompt
    (text_utils.py:37-63, incl. the r1_lite `@` rule), frame description, actions padded to action_dim.  With raw
    `language_actions` in the sample (training) the label text is produced by `lang_actions.ActionProcessor`, the frame
    description follows the frame actually used, and idle labels clear `sample_mask` (sample_handlers.py:372-431).
    VQA samples (`is_vqa_sample`: caption = label text, no wrist dropout / random un-masking, always active) and prediction
    samples (`is_prediction_sample`: both frames taken as they are; default prompt, or — with `enable_diverse_questions` — a
    question / answer pair drawn by `questions.PredictionSampleHandler`) follow input_transforms.py:214-249 and
    image_handler.py:45-107."""
    action_dim: int
    language_action_format: Any = "verbose_eef_with_rotation"
    wrist_image_dropout_prob: float = 0.0
    enable_langact_training: bool = True
    use_rough_scale: bool = False
    random_base_prob: float = 0.0
    random_mask_prob: float = 0.0
    enable_diverse_questions: bool = False
    question_config: Any = None
    image_keys: tuple[str, ...] = IMAGE_KEYS
    transform_strategy: str = "standard"     # "vla0" (sample_handlers.py:434-457): the label is the grid of the (normalised) action chunk

    def __post_init__(self):
        from lap_amd import lang_actions as la
        f = self.language_action_format
        if f is not None and not isinstance(f, la.LanguageActionFormat):
            object.__setattr__(self, "language_action_format", la.get_language_action_format(f))
        if self.enable_diverse_questions and self.question_config is None:
            from lap_amd.questions import QuestionConfig
            object.__setattr__(self, "question_config", QuestionConfig())

    def _mask(self, image, random_mask_prob: float = 0.0):
        if np.all(image == 0.0):
            return np.True_ if (random_mask_prob > 0.0 and np.random.rand() < random_mask_prob) else np.False_
        return np.True_

    def _wrist(self, obs: dict, key: str, base, is_vqa: bool = False):
        if key not in obs:
            return np.zeros_like(base)
        image = parse_image(obs[key])
        if not is_vqa and self.wrist_image_dropout_prob > 0.0 and np.random.rand() < float(self.wrist_image_dropout_prob):
            return np.zeros_like(base)
        return image

    def __call__(self, data: dict) -> dict:
        from lap_amd import lang_actions as la
        assert "observation" in data
        is_vqa, is_pred = bool(data.get("is_vqa_sample", False)), bool(data.get("is_prediction_sample", False))
        obs = data["observation"]
        raw = obs.get(self.image_keys[0])
        base = None if (isinstance(raw, (str, bytes)) and len(raw) == 0) else parse_image(raw)
        if base is None:
            base = np.zeros((224, 224, 3), dtype=np.uint8)   # masked out below: an all-zero image
        if not is_pred:
            images = [base] + [self._wrist(obs, k, base, is_vqa) for k in self.image_keys[1:]]
            masks = [self._mask(base)] + [self._mask(im, 0.0 if is_vqa else self.random_mask_prob) for im in images[1:]]
        else:   # prediction samples: the two frames as they are (image_handler.py:88-105), no dropout, no random un-masking
            first = base if data.get("pred_use_primary", False) else (parse_image(obs[self.image_keys[0]]) if self.image_keys[0] in obs
                                                                      else np.zeros_like(base))
            images = [first] + [parse_image(obs[k]) if k in obs else np.zeros_like(base) for k in self.image_keys[1:]]
            masks = [self._mask(im) for im in images]
        dataset_name = _text(data.get("dataset_name"))
        prompt = data.get("prompt")
        assert prompt is not None, "Prompt missing from data"
        prompt = _text(prompt)
        if "r1_lite" in dataset_name:
            prompt = prompt.split("@")[-1]
        out = {
            "state": obs["state"],
            "image": dict(zip(self.image_keys, images)),
            "image_mask": dict(zip(self.image_keys, masks)),
            "prompt": prompt,
            "is_prediction_sample": is_pred,
        }
        if dataset_name:
            out["dataset_name"] = dataset_name
        if "frame_description" in data:
            out["frame_description"] = _text(data["frame_description"], default="robot base frame")
        if "actions" in data:
            out["actions"] = np.array(pad_to_dim(data["actions"], self.action_dim))
        out["is_vqa_sample"] = is_vqa
        out["time_horizon_seconds"] = data.get("time_horizon_seconds")
        out["vqa_dataset_id"] = data.get("vqa_dataset_id", 0)
        if is_vqa:
            from lap_amd.questions import VQASampleHandler
            return VQASampleHandler(enable_diverse_questions=self.enable_diverse_questions).process(data, out)
        if is_pred:
            out["prompt"] = "predict the robot's action between two images in the prediction"
            if self.enable_diverse_questions and self.question_config is not None:
                from lap_amd.questions import PredictionSampleHandler
                proc = la.ActionProcessor(language_action_format=self.language_action_format, random_base_prob=self.random_base_prob)
                return PredictionSampleHandler(self.question_config, proc).process(data, out, dataset_name, data.get("rotation_applied", False))
        fmt = self.language_action_format
        if self.transform_strategy == "vla0":     # VLA-0: the text IS the normalised, padded action chunk as integers; never masked out
            out["language_actions"] = fmt.summarize_actions(out["actions"]) if "actions" in out else ""
            out["frame_description"] = "normalized"
            out["sample_mask"] = True
            return out
        if "language_actions" in data and self.enable_langact_training:
            proc = la.ActionProcessor(language_action_format=fmt, random_base_prob=self.random_base_prob)
            text, frame = proc.summarize_language_actions(data, "language_actions", np.asarray(data["raw_state"]), dataset_name,
                                                          data.get("rotation_applied", False))
            out["frame_description"] = frame
            out["language_actions"] = la.describe_language_action_scale(text) if self.use_rough_scale else text
            out["sample_mask"] = True if self.use_rough_scale else \
                not la.is_idle_language_action(out["language_actions"], fmt.get_sum_decimal(), fmt.include_rotation)
        else:
            out["sample_mask"] = True
        return out


# ------------------------------------------------------------------------------------------------ tokenizer
class PaligemmaTokenizer:
    """models/tokenizer.py:221-331.  The reference downloads gs://big_vision/paligemma_tokenizer.model; there is no network
    here, so the SentencePiece model file is supplied by the caller (`model_path` or raw `model_proto` bytes)."""

    def __init__(self, model_path: str | pathlib.Path | None = None, max_len: int = 48, prompt_format="lap",
                 prediction_format="default", reasoning_mask_prob: float = 0.0, *, model_proto: bytes | None = None):
        import sentencepiece

        if model_proto is None:
            if model_path is None:
                raise ValueError("PaligemmaTokenizer needs the PaliGemma SentencePiece model: pass model_path=... "
                                 "(paligemma_tokenizer.model) or model_proto=bytes")
            model_proto = pathlib.Path(model_path).read_bytes()
        self._tokenizer = sentencepiece.SentencePieceProcessor(model_proto=model_proto)
        self._max_len = max_len
        self.reasoning_mask_prob = reasoning_mask_prob
        self._prompt_format = _prompt.resolve_prompt_format(prompt_format)
        self._prediction_format = _prompt.resolve_prompt_format(prediction_format, _prompt.PREDICTION_PROMPT_FORMAT_REGISTRY)
        self._vqa_format = _prompt.DEFAULT_VQA_PROMPT_FORMAT

    def tokenize(self, prompt: str, reasoning: str | None = None, state=None, state_type: str | None = None, *,
                 is_vqa_sample: bool = False, is_prediction_sample: bool = False, time_horizon_seconds: float | None = None,
                 frame_description: str = "robot base frame", state_dropout: float = 0.0):
        """-> (tokens i32 [max_len], attn_mask, reasoning_mask | None, number_mask | None, direction_mask | None,
        token_loss_mask): BOS + prompt pieces, then (training) the cleaned language action + EOS; truncated, right padded."""
        fmt = self._prediction_format if is_prediction_sample else (self._vqa_format if is_vqa_sample else self._prompt_format)
        text = fmt.format_prompt(prompt, state, state_type, time_horizon_seconds=None if is_vqa_sample else time_horizon_seconds,
                                 frame_description=frame_description, state_dropout=state_dropout)
        sp = self._tokenizer
        tokens = sp.encode(text, add_bos=True, add_eos=False)
        start = len(tokens)
        if reasoning is not None:
            tokens += sp.encode(reasoning.strip().replace("_", " ").replace("\n", " "), add_bos=False, add_eos=True)
        end = len(tokens)
        L = self._max_len
        if len(tokens) > L:
            tokens, end = tokens[:L], min(end, L)
        attn = np.zeros(L, dtype=bool); attn[:len(tokens)] = True
        loss = np.ones(L, dtype=bool)
        reason = number = direction = None
        if reasoning is not None:
            reason = np.zeros(L, dtype=bool)
            a, b = max(0, min(L, start)), max(0, min(L, end))
            if b > a:
                reason[a:b] = True
            if not 0.0 <= self.reasoning_mask_prob <= 1.0:
                raise ValueError(f"reasoning_mask_prob must be between 0.0 and 1.0, got {self.reasoning_mask_prob}")
            idx = np.where(reason)[0]
            if self.reasoning_mask_prob > 0.0 and not is_vqa_sample and len(idx):
                loss[idx[np.random.rand(len(idx)) < self.reasoning_mask_prob]] = False
            number, direction = np.zeros(L, dtype=bool), np.zeros(L, dtype=bool)
            if not is_vqa_sample:
                for i in idx:
                    piece = sp.id_to_piece(int(tokens[i]))
                    if piece:
                        number[i] = _prompt.is_number(piece)
                        direction[i] = bool(fmt.direction_token_checker(piece))
        tokens = tokens + [sp.pad_id()] * (L - len(tokens))
        return np.asarray(tokens, dtype=np.int32), attn, reason, number, direction, loss

    def decode(self, tokens) -> str:
        ids = tokens.tolist() if not isinstance(tokens, list) else tokens
        n = self._tokenizer.vocab_size()
        return self._tokenizer.decode([int(t) for t in ids if 0 <= t < n]).strip()

    def encode(self, text: str, add_bos: bool = False, add_eos: bool = False):
        return self._tokenizer.encode(text, add_bos=add_bos, add_eos=add_eos)


@dataclasses.dataclass(frozen=True)
class TokenizePromptAndReasoning:
    """transforms.py:27-113: consumes prompt / language_actions / dataset_name / frame_description / time horizon, adds the
    tokenized fields of CoTObservation (+ the left-padded dataset-name ids)."""
    tokenizer: PaligemmaTokenizer
    discrete_state_input: bool = False
    dataset_name_pad_len: int = 100
    verbose_mode: bool = False
    state_dropout: float = 0.0

    def __call__(self, data: dict) -> dict:
        data = dict(data)
        prompt = data.pop("prompt", None)
        if prompt is None:
            raise ValueError("Prompt is required")
        if not isinstance(prompt, str):
            prompt = prompt.item()
        state = None
        if self.discrete_state_input:
            state = data.get("state")
            if state is None:
                raise ValueError("State is required.")
        language_actions = data.pop("language_actions", None)
        dataset_name = data.pop("dataset_name", None)
        frame_description = data.pop("frame_description", "robot base frame")
        sp = self.tokenizer._tokenizer
        name_ids = sp.encode(dataset_name) if dataset_name is not None else []
        name_ids = [sp.pad_id()] * (self.dataset_name_pad_len - len(name_ids)) + name_ids
        horizon = data.pop("time_horizon_seconds", None)
        tokens, pad_mask, reason, number, direction, loss = self.tokenizer.tokenize(
            prompt, language_actions, state, is_vqa_sample=data["is_vqa_sample"], is_prediction_sample=data["is_prediction_sample"],
            time_horizon_seconds=horizon, frame_description=frame_description, state_dropout=self.state_dropout)
        out = {**data, "tokenized_prompt": tokens, "tokenized_prompt_mask": pad_mask, "tokenized_langact_mask": reason,
               "token_loss_mask": loss, "tokenized_dataset_name": np.asarray(name_ids, dtype=np.int32)}
        if self.verbose_mode:
            out.update({"critical_token_mask": np.logical_or(number, direction), "number_token_mask": number,
                        "direction_token_mask": direction})
        return out


@dataclasses.dataclass(frozen=True)
class DetokenizeReasoning:
    """transforms.py:116-124."""
    tokenizer: PaligemmaTokenizer

    def __call__(self, data: dict) -> dict:
        if "tokens" in data:
            return {**data, "reasoning": self.tokenizer.decode(np.asarray(data["tokens"]).squeeze().astype(np.int32))}
        return data


# ------------------------------------------------------------------------------------------------ outputs
@dataclasses.dataclass(frozen=True)
class CoTOutputs:
    """policies/transforms/output_transforms.py:20-214.  Flow-matching policies: actions pass through, `reasoning` None.
    LAP_AR: the decoded text is parsed back into [dx, dy, dz, droll, dpitch, dyaw (, gripper)] with the configured language
    action format (end-effector-frame formats are rotated into the base frame with the request's `raw_state`); the VLA-0
    strategy returns the full [horizon, dim] grid, un-normalised with `norm_stats["actions"]`."""
    language_action_format: Any = None
    norm_stats: Any = None
    normalization_type: str = "bounds_q99"
    transform_strategy: str = "standard"

    def __post_init__(self):
        from lap_amd import lang_actions as la
        f = self.language_action_format
        if f is not None and not isinstance(f, la.LanguageActionFormat):
            object.__setattr__(self, "language_action_format", la.get_language_action_format(f))
        object.__setattr__(self, "norm_stats", as_norm_stats(self.norm_stats))

    def __call__(self, data: dict) -> dict:
        from lap_amd import lang_actions as la
        if "reasoning" not in data:
            return {"actions": np.asarray(data["actions"]), "reasoning": None}
        reasoning, fmt = data.get("reasoning"), self.language_action_format
        assert fmt is not None
        assert reasoning is not None
        if self.transform_strategy == "vla0" and isinstance(fmt, la.VLA0ActionFormat):
            return {"actions": self._unnormalize_vla0(fmt.parse_to_full_actions(reasoning)), "reasoning": reasoning}
        state = np.asarray(data["raw_state"]) if (self.transform_strategy != "vla0" and fmt.use_eef_frame and "raw_state" in data) else None
        movement, gripper = fmt.parse_language_to_deltas(reasoning, initial_state=state) if self.transform_strategy != "vla0" \
            else fmt.parse_language_to_deltas(reasoning)
        return {"actions": movement if gripper is None else np.concatenate([movement, [gripper]]), "reasoning": reasoning}

    def _unnormalize_vla0(self, actions: np.ndarray) -> np.ndarray:
        """output_transforms.py:106-190: only the recorded leading dimensions are mapped back; unknown types pass through."""
        st = None if self.norm_stats is None else self.norm_stats.get("actions")
        if st is None:
            return actions
        lo, hi, eps = {"bounds_q99": (st.q01, st.q99, 1e-6), "bounds": (st.min, st.max, 1e-8), "normal": (st.mean, st.std, None)}.get(
            self.normalization_type, (None, None, None))
        if lo is None or hi is None:
            return actions
        d = min(lo.shape[-1], actions.shape[-1])
        head = actions[..., :d]
        head = head * (hi[..., :d] + 1e-6) + lo[..., :d] if eps is None else (head + 1.0) / 2.0 * (hi[..., :d] - lo[..., :d] + eps) + lo[..., :d]
        return head if actions.shape[-1] <= d else np.concatenate([head, actions[..., d:]], axis=-1)


def compose(transforms: Sequence[Callable[[dict], dict]]) -> Callable[[dict], dict]:
    def run(data: dict) -> dict:
        for t in transforms:
            data = t(data)
        return data
    return run


# ------------------------------------------------------------------------------------------------ wire format
def _pack_array(obj):
    """openpi_client.msgpack_numpy.pack_array: ndarrays / numpy scalars as tagged maps (byte keys), everything else as is."""
    if isinstance(obj, (np.ndarray, np.generic)) and obj.dtype.kind in ("V", "O", "c"):
        raise ValueError(f"Unsupported dtype: {obj.dtype}")
    if isinstance(obj, np.ndarray):
        return {b"__ndarray__": True, b"data": obj.tobytes(), b"dtype": obj.dtype.str, b"shape": obj.shape}
    if isinstance(obj, np.generic):
        return {b"__npgeneric__": True, b"data": obj.item(), b"dtype": obj.dtype.str}
    return obj


def _unpack_array(obj):
    if b"__ndarray__" in obj:
        return np.ndarray(buffer=obj[b"data"], dtype=np.dtype(obj[b"dtype"]), shape=obj[b"shape"])
    if b"__npgeneric__" in obj:
        return np.dtype(obj[b"dtype"]).type(obj[b"data"])
    return obj


def packb(obj) -> bytes:
    import msgpack
    return msgpack.packb(obj, default=_pack_array)


def unpackb(data: bytes):
    import msgpack
    return msgpack.unpackb(data, object_hook=_unpack_array)
