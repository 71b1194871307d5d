"""Configuration surface of the reference, kept so that its callers stay drop-in.

Mirrors (names, defaults, argument meaning):
  * gemma variants            src/lap/models/backbones/gemma.py:43-109
  * LAPConfig                 src/lap/models/lap_config.py:22-130
  * TrainConfig / EMA / registry / get_config
                              src/lap/training/config.py:372-603, 607-862
  * AdamW / CosineDecaySchedule defaults: openpi.training.optimizer [UPSTREAM-RECALL], with the LAP
    overrides of training/config.py:69-82,516-519.
Only the fields the hot path reads are functional; data / wandb / weight-loader fields are carried as
plain values so existing config code keeps constructing.
"""
from __future__ import annotations

import dataclasses
import difflib
import math
import pathlib
from typing import Literal

PALIGEMMA_VOCAB_SIZE = 257_152
IMAGE_RESOLUTION = (224, 224)
# VQA dataset ids as the reference's registry assigns them (datasets/registry.py:243-304: 1, 2, ... in registration order =
# import order of datasets/vqa/__init__.py:10-18; 0 = not a VQA sample).  `LAPConfig.vqa_loss_weights` is keyed by these names
# (lap.py:107-115) and `CoTObservation.vqa_dataset_id` carries the ids.  Extend the dict for additional VQA sets.
VQA_DATASET_ID_MAP: dict[str, int] = {"coco_captions": 1, "lvis": 2, "paco_lvis": 3, "paco_ego4d": 4, "pixmo_cap": 5,
                                      "pixmo_point": 6, "vqa": 7}


# ------------------------------------------------------------------------------ backbones
@dataclasses.dataclass(frozen=True)
class GemmaConfig:
    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    num_kv_heads: int
    head_dim: int


_GEMMA = {
    "dummy": GemmaConfig(64, 4, 128, 8, 1, 16),
    "gemma_300m": GemmaConfig(1024, 18, 4096, 8, 1, 256),
    "gemma_2b": GemmaConfig(2048, 18, 16384, 8, 1, 256),
}


def get_gemma_config(variant: str) -> GemmaConfig:
    if variant not in _GEMMA:
        raise ValueError(f"Unknown variant: {variant}")  # gemma.py:109 (LoRA / gemma3 variants are out of scope)
    return _GEMMA[variant]


@dataclasses.dataclass(frozen=True)
class SiglipConfig:
    width: int
    depth: int
    mlp_dim: int
    num_heads: int
    patch: int = 14


_SIGLIP = {"So400m/14": SiglipConfig(1152, 27, 4304, 16), "mu/14": SiglipConfig(32, 1, 128, 2)}


def get_siglip_config(variant: str) -> SiglipConfig:
    if variant not in _SIGLIP:
        raise ValueError(f"Unknown SigLIP variant: {variant}")
    return _SIGLIP[variant]


# ------------------------------------------------------------------------------ parameter-path filters
@dataclasses.dataclass(frozen=True)
class PathFilter:
    """Engine counterpart of the nnx filters the reference builds from `nnx_utils.PathRegex` (full-match regexes over
    the '/'-joined parameter path), `nnx.All`, `nnx.Any` and `nnx.Not`: a path passes when it matches every entry of
    `all_of`, at least one of `any_of` (if given) and none of `none_of`.  Entries are regex strings or nested filters."""
    all_of: tuple = ()
    any_of: tuple = ()
    none_of: tuple = ()

    @staticmethod
    def _hit(entry, path: str) -> bool:
        import re

        return entry(path) if callable(entry) else re.fullmatch(entry, path) is not None

    def __call__(self, path: str) -> bool:
        if not all(self._hit(e, path) for e in self.all_of):
            return False
        if self.any_of and not any(self._hit(e, path) for e in self.any_of):
            return False
        return not any(self._hit(e, path) for e in self.none_of)


# ------------------------------------------------------------------------------ model config
@dataclasses.dataclass(frozen=True)
class LAPConfig:
    """lap_config.py:22-75.  `siglip_variant`, `image_size` and `vocab_size` are additions used only by the
    small test configurations; their defaults are the reference's constants."""

    dtype: str = "bfloat16"
    paligemma_variant: str = "gemma_2b"
    action_expert_variant: str = "gemma_300m"
    action_dim: int = 7
    action_horizon: int = 16
    max_token_len: int = 220
    verbose_mode: bool = False
    pi05: bool = True
    discrete_state_input: bool | None = True      # lap_config.py:37,79-80: None -> pi05
    prompt_format: str = "lap"
    prediction_format: str = "default"
    use_fast: bool = False
    aug_wrist_image: bool = True
    enable_image_augmentation: bool = True
    use_bimanual: bool = False
    enable_action_training: bool = False
    enable_langact_training: bool = True
    enable_prediction_training: bool = False
    enable_vqa_training: bool = False
    language_loss_weight: float = 1.0
    action_loss_weight: float = 1.0
    prediction_loss_weight: float = 1.0
    vqa_loss_weight: float = 0.1
    vqa_loss_weights: dict | None = None
    state_dropout: float = 0.0
    reasoning_mask_prob: float = 0.0
    stop_action_to_vlm_grad: bool = False
    # --- additions (test-size models) ---
    siglip_variant: str = "So400m/14"
    image_size: int = 224
    vocab_size: int = PALIGEMMA_VOCAB_SIZE

    def __post_init__(self):
        if self.max_token_len is None:
            object.__setattr__(self, "max_token_len", 200 if self.pi05 else 48)
        if self.discrete_state_input is None:
            object.__setattr__(self, "discrete_state_input", self.pi05)
        if self.dtype != "bfloat16":
            raise ValueError("lap_amd computes in bfloat16 (the reference's LAPConfig.dtype default)")
        if "gemma3" in self.paligemma_variant:
            raise NotImplementedError("Gemma3 LAP variants are out of scope (SURVEY.md §2)")

    @property
    def image_keys(self) -> tuple[str, ...]:
        if self.use_bimanual:
            return ("base_0_rgb", "left_wrist_0_rgb", "right_wrist_0_rgb")
        return ("base_0_rgb", "left_wrist_0_rgb")

    @property
    def image_resolution(self) -> tuple[int, int]:
        return (self.image_size, self.image_size)

    @property
    def model_type(self) -> str:
        return "lap_fast" if self.use_fast else "lap"

    def get_freeze_filter(self):
        """lap_config.py:132-169: only the LoRA variants (out of scope here) freeze anything -> nnx.Nothing."""
        return None

    def get_vlm_freeze_filter(self):
        """lap_config.py:171-191: freeze the VLM (every `llm` parameter that is not on the `_1` action-expert branch)
        and the image encoder; the action expert and the action heads stay trainable.  Returned as a predicate over
        the reference's '/'-joined parameter paths (the engine's counterpart of an nnx filter)."""
        return PathFilter(any_of=(PathFilter(all_of=(".*llm.*",), none_of=(".*llm.*_1.*",)), ".*img.*"))

    def create(self, rng=0, **kw):
        """lap_config.py:102-111: build a randomly initialised model (rng = integer seed here)."""
        from lap_amd.model import LAP

        return LAP(self, seed=int(rng), **kw)

    def load(self, params: dict, **kw):
        """openpi BaseModelConfig.load: build a model from a reference-layout parameter tree."""
        from lap_amd.model import LAP

        return LAP(self, params=params, **kw)

    def inputs_spec(self, *, batch_size: int = 1):
        """lap_config.py:113-130: (observation spec, action spec) as {name: (shape, dtype)}."""
        img = ((batch_size, *self.image_resolution, 3), "float32")
        obs = {
            "images": dict.fromkeys(self.image_keys, img),
            "image_masks": dict.fromkeys(self.image_keys, ((batch_size,), "bool")),
            "state": ((batch_size, self.action_dim), "float32"),
            "tokenized_prompt": ((batch_size, self.max_token_len), "int32"),
            "tokenized_prompt_mask": ((batch_size, self.max_token_len), "bool"),
            "tokenized_langact_mask": ((batch_size, self.max_token_len), "bool"),
            "critical_token_mask": ((batch_size, self.max_token_len), "bool"),
        }
        return obs, ((batch_size, self.action_horizon, self.action_dim), "float32")


# ------------------------------------------------------------------------------ optimizer configs
@dataclasses.dataclass(frozen=True)
class CosineDecaySchedule:
    warmup_steps: int = 1_000
    peak_lr: float = 2.5e-5
    decay_steps: int = 30_000
    decay_lr: float = 2.5e-6

    def __call__(self, step: int) -> float:
        """optax.warmup_cosine_decay_schedule(init=peak/(warmup+1), peak, warmup, decay_steps, end=decay_lr)."""
        init = self.peak_lr / (self.warmup_steps + 1)
        if step < self.warmup_steps:
            return init + (self.peak_lr - init) * step / self.warmup_steps
        frac = min(max((step - self.warmup_steps) / max(self.decay_steps - self.warmup_steps, 1), 0.0), 1.0)
        return self.decay_lr + (self.peak_lr - self.decay_lr) * 0.5 * (1 + math.cos(math.pi * frac))


@dataclasses.dataclass(frozen=True)
class AdamW:
    b1: float = 0.9
    b2: float = 0.95
    eps: float = 1e-8
    weight_decay: float = 1e-10
    clip_gradient_norm: float = 1.0


def build_cosine_lr(*, warmup_steps=5_000, peak_lr=1e-4, decay_steps=40_000, decay_lr=1e-4) -> CosineDecaySchedule:
    return CosineDecaySchedule(warmup_steps, peak_lr, decay_steps, decay_lr)  # training/config.py:69-82


# ------------------------------------------------------------------------------ EMA (config.py:372-504)
@dataclasses.dataclass(frozen=True)
class EmaStage:
    """One step interval [start_step, end_step) with a fixed EMA decay; `end_step=None` is open-ended and
    `decay=None` means "no EMA update inside this interval"."""
    start_step: int
    end_step: int | None = None
    decay: float | None = None

    def covers(self, step: int) -> bool:
        return step >= self.start_step and (self.end_step is None or step < self.end_step)

    def problems(self) -> list[str]:
        out = []
        if self.start_step < 0:
            out.append(f"negative start_step {self.start_step}")
        if self.end_step is not None and self.end_step <= self.start_step:
            out.append(f"empty interval [{self.start_step}, {self.end_step})")
        if self.decay is not None and not (0.0 < self.decay < 1.0):
            out.append(f"decay {self.decay} outside (0, 1)")
        return out


@dataclasses.dataclass(frozen=True)
class EmaSchedule:
    """Piecewise-constant EMA decay over the step axis: stages in increasing order, non-overlapping (gaps allowed — a
    step in a gap has no stage), only the last one may be open-ended."""
    stages: tuple[EmaStage, ...]

    def __post_init__(self):
        errors = [] if self.stages else ["no stages given"]
        for n, st in enumerate(self.stages):
            errors += [f"stage {n}: {msg}" for msg in st.problems()]
        for n, (a, b) in enumerate(zip(self.stages, self.stages[1:])):
            if a.end_step is None:
                errors.append(f"stage {n} is open-ended but stage {n + 1} follows it")
            elif b.start_step < a.end_step:
                errors.append(f"stage {n + 1} starts at {b.start_step}, inside stage {n} which ends at {a.end_step}")
        if errors:
            raise ValueError("invalid EmaSchedule: " + "; ".join(errors))

    def _find(self, step: int) -> EmaStage | None:
        return next((st for st in self.stages if st.covers(step)), None)

    def get_stage_for_step(self, step: int) -> EmaStage:
        st = self._find(step)
        if st is None:
            spans = ", ".join(f"[{s.start_step}, {'inf' if s.end_step is None else s.end_step})" for s in self.stages)
            raise ValueError(f"step {step} lies in no EMA stage (stages: {spans})")
        return st

    def get_decay_for_step(self, step: int) -> tuple[float, bool]:
        """(decay, enabled) for `step`; a step outside every stage, or in a stage without decay, has EMA off."""
        st = self._find(step)
        if st is None or st.decay is None:
            return 0.0, False
        return float(st.decay), True

    def has_ema(self) -> bool:
        return any(st.decay is not None for st in self.stages)

    def default_decay(self) -> float | None:
        return next((st.decay for st in self.stages if st.decay is not None), None)


@dataclasses.dataclass(frozen=True)
class EmaScheduleChoice:
    kind: Literal["disabled", "constant", "delayed", "cosine_delayed"] = "delayed"
    start_step: int = 10000

    def build(self, *, decay: float | None) -> EmaSchedule | None:
        if self.kind in ("disabled", "cosine_delayed") or decay is None:
            return None
        if self.kind == "constant" or self.start_step <= 0:
            return EmaSchedule((EmaStage(0, None, decay),))
        if self.kind == "delayed":
            return EmaSchedule((EmaStage(0, self.start_step, None), EmaStage(self.start_step, None, decay)))
        raise ValueError(f"Unsupported EMA schedule kind: {self.kind}")


# ------------------------------------------------------------------------------ data / weights (carried, not executed)
@dataclasses.dataclass(frozen=True)
class RLDSDataConfig:
    """training/config.py:85-148 (`DataConfig`) + 310-319 (`RLDSDataConfig`): the fields the data path here consumes, with the
    reference's defaults (held to them by tests/golden/train_configs_v1.json).  Not carried: TensorFlow pipeline knobs (thread counts,
    determinism), the Gemma-3 / DROID-variant fields."""
    repo_id: str | None = "oxe"        # config.py:317-319: the defaults are set for OXE training
    asset_id: str | None = "oxe"
    data_mix: str | None = "oxe_magic_soup"
    balance_weights: bool = True
    rlds_data_dir: str | None = "./data"
    shuffle_buffer_size: int = 1_000_000
    max_samples: int | None = None
    val_max_samples: int | None = None
    val_fraction: float | None = 0.025
    use_wrist_image: bool = True
    wrist_image_dropout_prob: float = 0.1
    action_proprio_normalization_type: str = "bounds_q99"   # config.py:98-99 (normal | bounds | bounds_q99)
    resize_resolution: tuple = (224, 224)
    # augmentation (CoTInputs, config.py:336-352)
    aug_wrist_image: bool = True
    random_base_prob: float = 0.0
    random_mask_prob: float = 0.2
    not_rotate_wrist_prob: float = 0.0
    use_rough_scale: bool = False
    language_action_format_name: str = "verbose_eef_with_rotation"
    transform_strategy: str = "standard"                    # "vla0": labels from the normalised action chunk (config.py:715,737)
    horizon_seconds: tuple = (1.0,)                         # label window(s) in seconds, one drawn per frame (base_dataset.py:493-531)
    # prediction samples and their questions
    max_prediction_horizon: int = 30
    pred_prob: float = 0.3
    primary_pred_prob: float = 0.8
    enable_diverse_questions: bool = True
    question_type_weights: dict | None = None
    delta_motion_format_weights: dict | None = None
    use_diverse_prompts: bool = True
    direction_prob: float = 0.0


@dataclasses.dataclass(frozen=True)
class WeightLoaderChoice:
    """weight_loaders.py:622-689.  kinds built here: "none", "checkpoint" (params_path = a checkpoint's `params` item) and
    "paligemma" — the reference's DEFAULT (weight_loaders.py:648-654), which is what its main `lap` config trains from.  The
    reference downloads gs://vertex-model-garden-paligemma-us/paligemma/pt_224.npz; there is no network here, so the file is
    looked up locally: params_path, else $LAP_PALIGEMMA_NPZ, else openpi's download cache
    ~/.cache/openpi/vertex-model-garden-paligemma-us/paligemma/pt_224.npz — and its absence is an error, never a silent
    random init."""
    kind: str = "paligemma"
    params_path: str | None = None

    def resolve_paligemma_path(self) -> pathlib.Path:
        import os

        cand = self.params_path or os.environ.get("LAP_PALIGEMMA_NPZ")
        if not cand:
            cache = os.environ.get("OPENPI_DATA_HOME", "~/.cache/openpi")
            cand = str(pathlib.Path(cache).expanduser() / "vertex-model-garden-paligemma-us" / "paligemma" / "pt_224.npz")
        return pathlib.Path(cand).expanduser()


# ------------------------------------------------------------------------------ TrainConfig (config.py:507-603)
@dataclasses.dataclass(frozen=True)
class TrainConfig:
    name: str = "lap"
    project_name: str = "lap"
    exp_name: str = ""
    model: LAPConfig = dataclasses.field(default_factory=lambda: LAPConfig(action_horizon=32, max_token_len=110))   # build_lap_model, config.py:53-66
    weight_loader: WeightLoaderChoice = dataclasses.field(default_factory=WeightLoaderChoice)
    data: RLDSDataConfig = dataclasses.field(default_factory=RLDSDataConfig)
    lr_schedule: CosineDecaySchedule = dataclasses.field(default_factory=build_cosine_lr)
    optimizer: AdamW = dataclasses.field(default_factory=lambda: AdamW(weight_decay=0.0001))
    batch_size: int = 32
    num_train_steps: int = 40_000
    save_interval: int = 1000
    log_interval: int = 50
    keep_period: int | None = 5000
    resume: bool = True
    overwrite: bool = False
    seed: int = 0
    fsdp_devices: int = 1
    ema_decay: float | None = 0.999
    ema_schedule_choice: EmaScheduleChoice = dataclasses.field(
        default_factory=lambda: EmaScheduleChoice(kind="cosine_delayed", start_step=5000))
    checkpoint_base_dir: str = "./checkpoints"
    assets_base_dir: str = "./assets"
    allow_partial_weights: bool = True
    use_validation: bool = False
    val_interval: int = 2000
    # openpi TrainConfig.freeze_filter (scripts/train.py:225-240,358-363): parameters it selects are kept out of the
    # gradient / optimizer and stored at bf16 precision.  None = nnx.Nothing; a regex string (full match on the
    # reference's '/'-joined path), a PathFilter or any predicate over the path.
    freeze_filter: object | None = None
    # Engine option (BASELINE.json config 5, not a reference field): "fp8" runs the forward and data-gradient GEMMs of
    # the VLM expert's projections on e4m3 operands with per-tensor scaling (csrc/gemm_fp8.hip); weight gradients,
    # the action expert, SigLIP and every non-GEMM op stay bf16.
    gemm_dtype: str = "bf16"

    def is_frozen(self, path: str) -> bool:
        f = self.freeze_filter
        if f is None:
            return False
        return PathFilter(all_of=(f,))(path) if isinstance(f, str) else bool(f(path))

    @property
    def trainable_filter(self):
        """openpi: nnx.All(nnx.Param, nnx.Not(freeze_filter)) as a predicate over parameter paths."""
        return lambda path: not self.is_frozen(path)

    @property
    def ema_schedule(self) -> EmaSchedule | None:
        return self.ema_schedule_choice.build(decay=self.ema_decay)

    def get_ema_init(self) -> tuple[float | None, bool]:
        if self.ema_schedule_choice.kind == "cosine_delayed":
            return (None, False) if self.ema_decay is None else (0.0, True)
        sched = self.ema_schedule
        if sched is None:
            return self.ema_decay, self.ema_decay is not None
        return sched.get_stage_for_step(0).decay, sched.has_ema()

    def get_ema_decay_for_step(self, step: int) -> tuple[float, bool]:
        if self.ema_schedule_choice.kind == "cosine_delayed":
            if self.ema_decay is None:
                return 0.0, False
            start = self.ema_schedule_choice.start_step
            dur = max(self.num_train_steps - start, 1)
            prog = min(max((step - start) / dur, 0.0), 1.0)
            return self.ema_decay * (1.0 - math.cos(math.pi * prog)) / 2.0, step >= start
        sched = self.ema_schedule
        if sched is not None:
            return sched.get_decay_for_step(step)
        if self.ema_decay is None:
            return 0.0, False
        return float(self.ema_decay), True

    @property
    def assets_dirs(self) -> pathlib.Path:
        return pathlib.Path(self.assets_base_dir) / self.name

    @property
    def checkpoint_dir(self) -> pathlib.Path:
        if not self.exp_name:
            raise ValueError("--exp_name must be set")
        return pathlib.Path(self.checkpoint_base_dir) / self.name / self.exp_name


_CONFIGS = [
    TrainConfig(  # config.py:608-619
        name="lap",
        data=RLDSDataConfig(random_base_prob=0.5),
        model=LAPConfig(action_dim=7, action_horizon=16, max_token_len=180, enable_action_training=True,
                        stop_action_to_vlm_grad=True),
        batch_size=2048,
    ),
    TrainConfig(  # config.py:752-785
        name="lap_libero",
        model=LAPConfig(action_dim=7, action_horizon=10, max_token_len=180, enable_action_training=True,
                        stop_action_to_vlm_grad=False, language_loss_weight=0.4, enable_image_augmentation=False),
        data=RLDSDataConfig(shuffle_buffer_size=100000, repo_id="libero", asset_id="libero", data_mix="libero_finetune",
                            val_fraction=0.0),
        lr_schedule=CosineDecaySchedule(warmup_steps=1000, peak_lr=5e-5, decay_steps=40_000, decay_lr=5e-5),
        weight_loader=WeightLoaderChoice(kind="checkpoint", params_path="checkpoints/lap/params"),
        save_interval=2000, keep_period=2000, num_train_steps=40_001, batch_size=256,
        ema_schedule_choice=EmaScheduleChoice(kind="constant"),
    ),
    TrainConfig(  # config.py:632-642 (named after pi0; pi05 keeps its default: flow matching only, no language-action loss)
        name="pi0_replicated",
        model=LAPConfig(action_dim=7, action_horizon=16, max_token_len=220, enable_action_training=True, enable_langact_training=False),
        batch_size=2048,
    ),
    TrainConfig(  # config.py:786-797
        name="lap_cotrain",
        model=LAPConfig(action_dim=7, action_horizon=16, max_token_len=220, enable_action_training=True, enable_prediction_training=True,
                        stop_action_to_vlm_grad=True),
        batch_size=2048,
    ),
    TrainConfig(  # config.py:701-717 ("VLA-0": actions as text, no action expert)
        name="vla0_replicated",
        model=LAPConfig(action_dim=7, action_horizon=10, max_token_len=390, pi05=True, discrete_state_input=True, enable_action_training=False,
                        enable_langact_training=True, paligemma_variant="gemma_2b", action_expert_variant="gemma_300m", prompt_format="vla0_chunked"),
        data=RLDSDataConfig(language_action_format_name="vla0_chunked", transform_strategy="vla0"),
        batch_size=2048,
    ),
    TrainConfig(  # config.py:718-751
        name="vla0_replicated_libero",
        model=LAPConfig(action_dim=7, action_horizon=10, max_token_len=390, enable_action_training=False, enable_langact_training=True,
                        paligemma_variant="gemma_2b", action_expert_variant="gemma_300m", prompt_format="vla0_chunked", reasoning_mask_prob=0.2),
        data=RLDSDataConfig(shuffle_buffer_size=100000, repo_id="libero", asset_id="libero", data_mix="libero_finetune", val_fraction=0.0,
                            language_action_format_name="vla0_chunked", transform_strategy="vla0"),
        lr_schedule=CosineDecaySchedule(warmup_steps=1000, peak_lr=5e-5, decay_steps=40_000, decay_lr=5e-5),
        save_interval=2000, keep_period=2000, num_train_steps=40_001, batch_size=256,
        ema_schedule_choice=EmaScheduleChoice(kind="cosine_delayed", start_step=1000),
    ),
    TrainConfig(  # BASELINE.json synthetic shapes: 48-token prompt, 50-step chunk
        name="lap_bench",
        model=LAPConfig(action_dim=7, action_horizon=50, max_token_len=48, enable_action_training=True,
                        stop_action_to_vlm_grad=False, language_loss_weight=0.4, enable_image_augmentation=False),
        lr_schedule=CosineDecaySchedule(warmup_steps=1000, peak_lr=5e-5, decay_steps=40_000, decay_lr=5e-5),
        weight_loader=WeightLoaderChoice(kind="none"),   # benchmark: random-init weights of the LAP-3B architecture
        batch_size=32, ema_schedule_choice=EmaScheduleChoice(kind="constant"),
    ),
    TrainConfig(  # tiny model for tests (gemma "dummy" variant gemma.py:60-68, SigLIP "mu")
        name="debug",
        model=LAPConfig(paligemma_variant="dummy", action_expert_variant="dummy", siglip_variant="mu/14", image_size=56,
                        vocab_size=512, action_dim=7, action_horizon=10, max_token_len=24, enable_action_training=True,
                        language_loss_weight=0.4, enable_image_augmentation=False),
        lr_schedule=CosineDecaySchedule(warmup_steps=2, peak_lr=1e-3, decay_steps=100, decay_lr=1e-3),
        weight_loader=WeightLoaderChoice(kind="none"),
        batch_size=2, num_train_steps=10, ema_schedule_choice=EmaScheduleChoice(kind="constant"),
    ),
]
if len({c.name for c in _CONFIGS}) != len(_CONFIGS):
    raise ValueError("Config names must be unique.")
_CONFIGS_DICT = {c.name: c for c in _CONFIGS}


def get_config(config_name: str) -> TrainConfig:
    """training/config.py:843-862."""
    if config_name in _CONFIGS_DICT:
        return _CONFIGS_DICT[config_name]
    closest = difflib.get_close_matches(config_name, _CONFIGS_DICT.keys(), n=3, cutoff=0.0)
    closest_str = f" Did you mean one of: {', '.join(repr(c) for c in closest)}?" if closest else ""
    raise ValueError(f"Config '{config_name}' not found.{closest_str}")


def cli(argv=None) -> TrainConfig:
    """Minimal stand-in for tyro.extras.overridable_config_cli (config.py:839): `<name> [--field value ...]`
    for top-level scalar fields (exp-name, batch-size, fsdp-devices, num-train-steps, seed, ...)."""
    import sys

    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        fields = ", ".join("--" + f.name.replace("_", "-") for f in dataclasses.fields(TrainConfig)
                           if str(f.type).replace(" ", "").replace("|None", "") in ("int", "float", "str", "bool"))
        raise SystemExit(f"usage: <config> [--field value ...]\nconfigs: {sorted(_CONFIGS_DICT)}\nscalar fields: {fields}")
    cfg = get_config(argv.pop(0))
    upd = {}
    fields = {f.name: f for f in dataclasses.fields(cfg)}
    casts = {"int": int, "float": float, "str": str}
    while argv:
        key = argv.pop(0)
        if not key.startswith("--") or not argv:
            raise SystemExit(f"bad argument {key}")
        name = key[2:].replace("-", "_")
        field = fields.get(name)
        if field is None:
            raise SystemExit(f"unknown option {key}")
        val = argv.pop(0)
        ann = str(field.type).replace(" ", "")              # "int", "bool", "float|None", "int|None", "str"
        base = ann.replace("|None", "")
        if base == "bool":
            if val.lower() not in ("1", "0", "true", "false"):
                raise SystemExit(f"{key} expects true / false, got {val!r}")
            upd[name] = val.lower() in ("1", "true")
        elif base in casts:
            if val.lower() == "none" and ann.endswith("|None"):
                upd[name] = None
            else:
                try:
                    upd[name] = casts[base](val)
                except ValueError:
                    raise SystemExit(f"{key} expects {base}, got {val!r}") from None
        else:   # nested dataclasses (model, data, optimizer, ...) cannot be set from a single string
            raise SystemExit(f"{key} is a structured field ({ann}); only scalar top-level fields can be overridden here")
    return dataclasses.replace(cfg, **upd)
