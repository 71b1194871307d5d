"""Prompt templating and state discretisation of the LAP policies.

Restates `src/lap/models/prompt_utils/{prompt,state,checkers}.py` of the reference: the text the tokenizer sees for a
robot sample is

    Task: <cleaned instruction>, predict the robot's action in the <frame>; State: <256-bin integers>; Answer: <space>

(prompt.py:156-169, the "lap" format) and the VQA / prediction / VLA-0 variants registered next to it
(prompt.py:172-332).  Parity is PINNED: these reference modules are pure Python/numpy and import in the build
container, `tests/golden/make_prompt_golden.py` runs them on a case table and `tests/golden/prompt_format_v1.json`
holds their outputs; `tests/test_policy_io_cpu.py` replays the table through this module.

One flat description per format instead of the reference's module objects; the registries keep the reference's names.
"""
from __future__ import annotations

import dataclasses
import random
import re
from typing import Callable, Sequence

import numpy as np

# ------------------------------------------------------------------------------------------------ token classes
_DIRECTION_WORDS = ("right", "left", "forward", "up", "down", "back", "clockwise", "counterclockwise")


def is_number(piece: str) -> bool:
    """checkers.py:4-6: the piece contains a digit."""
    return re.search(r"[0-9]", piece) is not None


def is_direction_natural(piece: str) -> bool:
    """checkers.py:9-13: a direction word occurs inside the (lower-cased) piece."""
    low = piece.lower()
    return any(w in low for w in _DIRECTION_WORDS)


def is_direction_schema(piece: str) -> bool:
    """checkers.py:16-18."""
    return ("+" in piece) or ("-" in piece)


def is_direction_none(piece: str) -> bool:
    return False


def is_critical_directional(piece: str) -> bool:
    return is_number(piece) or is_direction_natural(piece)


def is_critical_schema(piece: str) -> bool:
    return is_number(piece) or is_direction_schema(piece)


# ------------------------------------------------------------------------------------------------ state -> text
@dataclasses.dataclass(frozen=True)
class StateText:
    """How a state vector becomes text (state.py:111-157 + the templates of state.py:6-108, 211-250).

    `layout`: "plain" -> "12 200 7"; "labelled" -> per-dimension `item` strings joined by `sep` (labels beyond the list are
    dim<i>); "grouped" -> "position 134 088 076, rotation ..." with `groups` = ((label, size), ...)."""
    bins: int = 256
    min_dim: int = 10
    lo: float = -1.0
    hi: float = 1.0
    layout: str = "plain"
    labels: tuple[str, ...] = ()
    item: str = "{value}"
    sep: str = " "
    groups: tuple[tuple[str, int], ...] = ()
    group_sep: str = ", "

    def bin_indices(self, state) -> np.ndarray:
        """Trailing all-zero dimensions (|x| <= 1e-8, but never below `min_dim`) are cut, the rest is mapped to
        `bins` equal cells of [lo, hi): index = digitize(x, left edges) - 1, i.e. -1 below `lo`, bins-1 at and above the
        last edge.  A 2-D input is cut on its last axis and flattened."""
        a = np.asarray(state)
        if a.ndim == 1:
            used = np.abs(a) > 1e-8
        else:
            used = np.any(np.abs(a.reshape(-1, a.shape[-1])) > 1e-8, axis=0)
        keep = (int(np.nonzero(used)[0][-1]) + 1) if used.any() else 0
        keep = max(keep, self.min_dim)
        cut = a[:keep] if a.ndim == 1 else a[..., :keep].reshape(-1)
        if cut.size == 0:
            return np.zeros((0,), dtype=np.int64)
        edges = np.linspace(self.lo, self.hi, self.bins + 1)[:-1]
        return np.digitize(cut, bins=edges) - 1

    def render(self, state) -> str:
        idx = self.bin_indices(state)
        if idx.size == 0:
            return ""
        if self.layout == "plain":
            return " ".join(str(v) for v in idx)
        if self.layout == "labelled":
            names = [self.labels[i] if i < len(self.labels) else f"dim{i}" for i in range(len(idx))]
            return self.sep.join(self.item.format(label=n, value=int(v)) for n, v in zip(names, idx))
        if self.layout == "grouped":
            out, at = [], 0
            for label, size in self.groups:
                if at >= len(idx):
                    break
                vals = idx[at:at + size]   # a short state simply yields shorter / fewer groups
                out.append(f"{label} " + self.sep.join(self.item.format(value=int(v)) for v in vals))
                at += size
            return self.group_sep.join(out)
        raise ValueError(f"unknown state layout {self.layout!r}")


DEFAULT_STATE_TEXT = StateText()
NAMED_PARAMS_STATE_TEXT = StateText(layout="labelled", item="{label}={value:03d}", sep=" ",
                                    labels=("x", "y", "z", "rot1x", "rot1y", "rot1z", "rot2x", "rot2y", "rot2z", "grip"))
VERBOSE_STATE_TEXT = StateText(layout="labelled", item="{label}={value:03d}", sep=", ",
                               labels=("position_x", "position_y", "position_z", "rotation_1_x", "rotation_1_y", "rotation_1_z",
                                       "rotation_2_x", "rotation_2_y", "rotation_2_z", "gripper"))
GROUPED_STATE_TEXT = StateText(layout="grouped", item="{value:03d}", sep=" ",
                               groups=(("position", 3), ("rotation", 3), ("gripper", 1)))

_STATE_TYPE_LABEL = {"joint_pos": " (joint position)", "eef_pose": " (end-effector pose)"}


# ------------------------------------------------------------------------------------------------ prompt formats
@dataclasses.dataclass(frozen=True)
class PromptFormat:
    """prompt.py:88-154 with its four optional parts flattened: `prefix`, task sentence, state sentence, answer prefix."""
    name: str
    prefix: str | None = None
    task: str | None = "Task: {prompt}, predict the robot's action in the {frame_description}"
    with_time_horizon: bool = False
    horizon: str = "predict the robot's action in the future {time_horizon_seconds} seconds in the {frame_description}"
    state: StateText | None = None
    state_sentence: str = "State{state_label}: {state}"
    show_state_type: bool = True
    answer: str | None = "Action: "
    separator: str = ""
    critical_token_checker: Callable[[str], bool] | None = is_number
    direction_token_checker: Callable[[str], bool] | None = is_direction_none

    @property
    def include_state(self) -> bool:
        return self.state is not None

    def _task_text(self, prompt: str, time_horizon_seconds, frame_description: str) -> str:
        text = prompt.strip().replace("_", " ").replace("\n", " ").rstrip(".")
        if self.with_time_horizon:
            assert time_horizon_seconds is not None, "Time horizon must be provided if include_time_horizon is True"
            half_steps = round(time_horizon_seconds * 2) / 2.0
            # (the reference formats only the seconds here; a {frame_description} in `horizon` would raise, as there)
            text += ", " + self.horizon.format(time_horizon_seconds=half_steps)
        return self.task.format(prompt=text, frame_description=frame_description)

    def _state_text(self, state, state_type) -> str:
        if state is None or state_type == "none":
            return self.state_sentence.format(state="", state_label="None" if self.show_state_type else "")
        label = (_STATE_TYPE_LABEL.get(state_type, state_type) if state_type else "") if self.show_state_type else ""
        return self.state_sentence.format(state=self.state.render(state), state_label=label)

    def format_prompt(self, prompt: str, state=None, state_type: str | None = None, time_horizon_seconds: float | None = None,
                      frame_description: str = "robot base frame", state_dropout: float = 0.0) -> str:
        parts: list[str] = []
        if self.prefix is not None:
            parts.append(self.prefix)
        if self.task is not None:
            parts.append(self._task_text(prompt, time_horizon_seconds, frame_description))
        # the dropout draw happens only when there is a state to drop (prompt.py:139-141, short-circuit order kept)
        if not (self.state is None or state is None or (state_dropout > 0.0 and random.random() < state_dropout)):
            sentence = self._state_text(state, state_type)
            if sentence:
                parts.append(sentence)
        if self.answer is not None:
            parts.append(self.answer)
        return self.separator.join(parts)


def _question(name, critical=None, direction=None, state=None) -> PromptFormat:
    return PromptFormat(name=name, task="Task: {prompt}", state=state, show_state_type=False, answer="Answer: ", separator="; ",
                        critical_token_checker=critical, direction_token_checker=direction)


LAP_PROMPT_FORMAT = PromptFormat(name="lap", state=StateText(bins=256), show_state_type=False, answer="Answer: ", separator="; ",
                                 critical_token_checker=is_critical_directional, direction_token_checker=is_direction_natural)
VLA0_CHUNKED_PROMPT_FORMAT = PromptFormat(
    name="vla0_chunked",
    prefix=("Analyze the input image and predict robot actions for the next 10 timesteps. "
            "Each action has 7 dimensions. Output a single sequence of 70 integers (0-1000 each), "
            "representing the 10 timesteps sequentially. Provide only space-separated numbers. Nothing else."),
    task="Task: {prompt}", answer="", separator="\n", critical_token_checker=is_number, direction_token_checker=is_direction_none)
DEFAULT_VQA_PROMPT_FORMAT = _question("default_vqa")
DEFAULT_PREDICTION_PROMPT_FORMAT = _question("default_prediction", is_critical_schema, is_direction_schema, state=StateText(bins=256))

PROMPT_FORMAT_REGISTRY: dict[str, PromptFormat] = {"lap": LAP_PROMPT_FORMAT, "vla0_chunked": VLA0_CHUNKED_PROMPT_FORMAT}
PREDICTION_PROMPT_FORMAT_REGISTRY: dict[str, PromptFormat] = {
    "default": DEFAULT_PREDICTION_PROMPT_FORMAT,
    "task_prediction": _question("task_prediction"),
    "direction_classification": _question("direction_classification", is_direction_natural, is_direction_natural),
    "gripper_prediction": _question("gripper_prediction"),
    "magnitude_estimation": _question("magnitude_estimation"),
    "temporal_ordering": _question("temporal_ordering"),
    "embodiment_identification": _question("embodiment_identification"),
}


def resolve_prompt_format(fmt: str | PromptFormat, registry: dict[str, PromptFormat] | None = None) -> PromptFormat:
    """tokenizer.py:51-71: registry lookup with the reference's error."""
    if isinstance(fmt, PromptFormat):
        return fmt
    registry = PROMPT_FORMAT_REGISTRY if registry is None else registry
    if fmt not in registry:
        kind = "prediction" if registry is PREDICTION_PROMPT_FORMAT_REGISTRY else "prompt"
        raise ValueError(f"Unknown {kind} format: {fmt}. Available formats: {list(registry.keys())}")
    return registry[fmt]


def format_cases() -> Sequence[dict]:
    """The case table shared by the golden generator (reference side) and the parity test (this module)."""
    rs = np.random.RandomState(0)
    states = {
        "eef10": np.round(rs.uniform(-1, 1, 10), 4).tolist(),
        "padded32": np.concatenate([np.round(rs.uniform(-1.2, 1.2, 7), 4), np.zeros(25)]).tolist(),
        "edges": [-1.0, -0.9999999, -0.9921875, 0.0, 0.9921874, 0.9921875, 0.999, 1.0, 1.5, -1.5, 0.5, 0.25],
        "zeros": [0.0] * 8,
        "short3": [0.3, -0.2, 0.9],
        "two_rows": np.round(rs.uniform(-1, 1, (2, 12)), 4).tolist(),
        "tiny": [1e-9, 0.5, 0.0, 0.0],
    }
    cases = []
    prompts = ["pick up the red_block and place it\nin the bowl.", "  open the drawer ", "Close the microwave..", ""]
    for fmt in ["lap", "vla0_chunked"]:
        for pi, p in enumerate(prompts):
            for sn in [None, "eef10", "padded32", "edges", "zeros", "two_rows"]:
                for st in [None, "eef_pose", "none"]:
                    if (pi + len(sn or "") + len(st or "")) % 2 and sn not in (None, "edges"):
                        continue
                    cases.append({"registry": "prompt", "format": fmt, "prompt": p, "state": sn, "state_type": st,
                                  "frame": "robot base frame" if pi % 2 == 0 else "end-effector frame", "horizon": None})
    for fmt in PREDICTION_PROMPT_FORMAT_REGISTRY:
        for sn in [None, "short3", "tiny", "eef10"]:
            cases.append({"registry": "prediction", "format": fmt, "prompt": "what moved between the frames?", "state": sn,
                          "state_type": "joint_pos", "frame": "robot base frame", "horizon": 1.26})
    cases.append({"registry": "vqa", "format": "default_vqa", "prompt": "How many cups are on the table?", "state": "eef10",
                  "state_type": None, "frame": "robot base frame", "horizon": None})
    return [dict(c, state_values=states.get(c["state"])) for c in cases]


STATE_TEXTS = {"default": DEFAULT_STATE_TEXT, "named_params": NAMED_PARAMS_STATE_TEXT, "verbose": VERBOSE_STATE_TEXT,
               "grouped": GROUPED_STATE_TEXT}
CHECKERS = {"is_number": is_number, "is_direction_natural": is_direction_natural, "is_direction_schema": is_direction_schema,
            "is_direction_none": is_direction_none, "is_critical_directional": is_critical_directional,
            "is_critical_schema": is_critical_schema}


def state_text_cases() -> Sequence[dict]:
    rs = np.random.RandomState(1)
    vecs = [np.round(rs.uniform(-1, 1, n), 4).tolist() for n in (10, 12, 7, 4, 2, 1)] + [[0.0] * 10, [-1.0, 1.0, 0.0]]
    return [{"template": t, "min_dim": md, "state_values": v} for t in STATE_TEXTS for md in (10, 0) for v in vecs]


CHECKER_PIECES = ["▁move", "▁12", "3cm", "▁Right", "▁upward", "▁backwards", "counterclockwise", "+", "-5", "▁gripper", "",
                  "▁x=", "▁left-hand", "▁0", "▁Clock"]
