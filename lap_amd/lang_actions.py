"""Language actions: numeric end-effector deltas <-> text ("move forward 3 cm, tilt left 10 degrees, open gripper").

Restates `src/lap/policies/lang_action_formats.py`, `policies/transforms/action_text.py` and
`policies/transforms/frame_transforms.py` of the reference — the text side of the LAP_AR serving mode (decoded tokens ->
deltas, `CoTOutputs`) and of the language-action training labels (trajectory chunk -> summary text).  Parity is PINNED:
these modules are numpy / scipy only and run in the build container (`tests/golden/make_lang_action_golden.py`, fixture
`tests/golden/lang_action_v1.json`, replayed by `tests/test_policy_io_cpu.py`).

Conventions kept from the reference: actions are [dx, dy, dz, droll, dpitch, dyaw, gripper] in metres / radians, x forward,
y left, z up; text is in centimetres / degrees; gripper >= 0.5 reads "open gripper".  Tables instead of if-chains.
"""
from __future__ import annotations

import dataclasses
import logging
import re

import numpy as np

# direction word -> (axis, sign) as used when PARSING text
_MOVE_AXIS = {"forward": (0, 1.0), "backward": (0, -1.0), "back": (0, -1.0), "left": (1, 1.0), "right": (1, -1.0),
              "up": (2, 1.0), "down": (2, -1.0)}
_ROT_AXIS_PARSE = {"tilt left": (0, 1.0), "tilt right": (0, -1.0), "tilt down": (1, 1.0), "tilt back": (1, 1.0),
                   "tilt up": (1, -1.0), "tilt forward": (1, -1.0), "rotate counterclockwise": (2, 1.0), "rotate clockwise": (2, -1.0)}
# (the idle filter uses the opposite pitch sign, action_text.py:289-292; only the norm is used there)
_ROT_AXIS_IDLE = dict(_ROT_AXIS_PARSE, **{"tilt down": (1, -1.0), "tilt back": (1, -1.0), "tilt up": (1, 1.0), "tilt forward": (1, 1.0)})
_ROT_WORDS = "tilt left|tilt right|tilt up|tilt down|tilt back|tilt forward|rotate clockwise|rotate counterclockwise"
_ROT_RE = re.compile(rf"({_ROT_WORDS})\s+([\d.]+)\s*degrees", re.IGNORECASE)
_COMPACT6_RE = re.compile(r"<([+\-]\d+)\s+([+\-]\d+)\s+([+\-]\d+)\s+([+\-]\d+)\s+([+\-]\d+)\s+([+\-]\d+)\s+(\d)>")
_COMPACT3_RE = re.compile(r"<([+\-]\d+)\s+([+\-]\d+)\s+([+\-]\d+)\s+\d>")
# axis -> (word for +, word for -) when WRITING text
_MOVE_WORDS = (("move forward", "move back"), ("move left", "move right"), ("move up", "move down"))
_ROT_WRITE = (("tilt left", "tilt right"), ("tilt back", "tilt forward"), ("rotate counterclockwise", "rotate clockwise"))


# ------------------------------------------------------------------------------------------------ frames
def rot6d_to_rotmat(rot6d) -> np.ndarray:
    """frame_transforms.py:7-18: Gram-Schmidt on the two 3-vectors; the orthonormal triple forms the COLUMNS."""
    r = np.asarray(rot6d)
    b1 = r[..., 0:3] / np.linalg.norm(r[..., 0:3], axis=-1, keepdims=True)
    rest = r[..., 3:6] - np.sum(b1 * r[..., 3:6], axis=-1, keepdims=True) * b1
    b2 = rest / np.linalg.norm(rest, axis=-1, keepdims=True)
    return np.stack([b1, b2, np.cross(b1, b2, axis=-1)], axis=-1)


def _euler_to_mat(e):
    from scipy.spatial.transform import Rotation
    return Rotation.from_euler("xyz", e).as_matrix()


def _mat_to_euler(m):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(m).as_euler("xyz")


def _initial_rotation(state: np.ndarray) -> np.ndarray:
    return _euler_to_mat(state[3:6]) if len(state) == 7 else rot6d_to_rotmat(state[3:9])


def transform_actions_to_eef_frame(actions, initial_state, dataset_name, needs_wrist_rotation: bool = False) -> np.ndarray:
    """frame_transforms.py:21-70: one base-frame delta -> the gripper camera's frame (y, z mirrored), with the
    per-dataset axis conventions of the reference."""
    a = np.asarray(actions, dtype=float)
    s = np.asarray(initial_state, dtype=float)
    assert a.ndim == 1
    out = a.copy()
    to_eef = rot6d_to_rotmat(s[3:9]).T
    p = to_eef @ a[:3]
    p[1], p[2] = -p[1], -p[2]
    if "jaco_play" in dataset_name:
        p = np.array([p[1], p[0], -p[2]])
    elif "berkeley_autolab_ur5" in dataset_name:
        p = np.array([-p[1], p[0], p[2]])
    out[:3] = p
    e = _mat_to_euler(to_eef @ _euler_to_mat(a[3:6]) @ to_eef.T)
    if not needs_wrist_rotation:
        e[1], e[2] = -e[1], -e[2]
    if any(k in dataset_name for k in ("furniture_bench_dataset_converted_externally_to_rlds", "austin", "fmb", "viola")):
        e[1], e[2] = -e[1], -e[2]
    elif "berkeley_autolab_ur5" in dataset_name:
        e[1] = -e[1]
    out[3:6] = e
    return out


def transform_actions_from_eef_frame(actions, initial_state, dataset_name: str = "") -> np.ndarray:
    """frame_transforms.py:73-129: the inverse direction for one action or a [T, >=3] chunk; state is [xyz, euler, grip]
    (7 values) or [xyz, rot6d, ...]."""
    a = np.asarray(actions, dtype=float)
    s = np.asarray(initial_state, dtype=float)
    if s.ndim == 2:
        assert s.shape[0] == 1
        s = s[0]
    if a.ndim == 1:
        a = a[None, :]
    out = a.copy()
    to_base = _initial_rotation(s)
    for i in range(len(a)):
        p = a[i, :3].copy()
        if "jaco_play" in dataset_name:
            p = np.array([p[1], p[0], -p[2]])
        elif "berkeley_autolab" in dataset_name:
            p = np.array([p[1], -p[0], p[2]])
        else:
            p[1], p[2] = -p[1], -p[2]
        out[i, :3] = to_base @ p
        if a.shape[-1] >= 6:
            e = a[i, 3:6].copy()
            mirrored = any(k in dataset_name for k in ("furniture_bench", "utaustin", "fmb"))   # checked first, as there
            if mirrored or not ("berkeley_autolab" in dataset_name or "jaco_play" in dataset_name):
                e[1], e[2] = -e[1], -e[2]
            elif "berkeley_autolab" in dataset_name:
                e[1] = -e[1]
            out[i, 3:6] = _mat_to_euler(to_base @ _euler_to_mat(e) @ to_base.T)
    return out


# ------------------------------------------------------------------------------------------------ numbers -> text
def _nearest(value: float, n: int) -> int:
    return int(round(value / n) * n)


def _decimals(sum_decimal: str) -> int:
    if sum_decimal in ("no_number", "nearest_10"):
        return 0
    return int(re.fullmatch(r"(\d+)f", sum_decimal).group(1))


def _number(val: float, sum_decimal: str) -> str:
    if sum_decimal == "no_number":
        return ""
    if sum_decimal == "nearest_10":
        return str(int(round(val / 10) * 10))
    m = re.fullmatch(r"(\d+)f", sum_decimal)
    return f"{val:.{int(m.group(1)) if m else 0}f}"


def _compact(arr: np.ndarray, include_rotation: bool) -> str:
    cm = [int(round(float(arr[..., i].sum()) * 100.0)) for i in range(3)]
    parts = [f"{v:+03d}" for v in cm]
    if include_rotation:
        parts += [f"{_nearest(float(arr[..., i].sum()) * 180.0 / np.pi, 5):+03d}" for i in (3, 4, 5)]
    parts.append("1" if float(arr[-1, 6]) >= 0.5 else "0")
    return "<" + " ".join(parts) + ">"


def summarize_numeric_actions(arr_like, sum_decimal: str, include_rotation: bool = False, rotation_precision: int = 10) -> str | None:
    """action_text.py:47-141: the chunk's summed translation (cm) / rotation (degrees, to `rotation_precision`) and the
    LAST gripper value as text.  `sum_decimal`: "<N>f" | "nearest_10" | "no_number" | "compact"."""
    arr = np.asarray(arr_like, dtype=float)
    if arr.ndim == 1:
        arr = arr[None, :]
    if arr.shape[-1] < 7:
        return None
    if sum_decimal == "compact":
        return _compact(arr, include_rotation)
    nd = _decimals(sum_decimal)
    metres = [float(arr[..., i].sum()) for i in range(3)]
    cm = [round(abs(m * 100.0), nd) for m in metres]
    rad = [float(arr[..., i].sum()) for i in (3, 4, 5)] if include_rotation else []
    deg = [_nearest(abs(r * 180.0 / np.pi), rotation_precision) for r in rad]
    words = sum_decimal == "no_number"
    parts: list[str] = []
    # numeric text lists x, z, y; the number-free variant lists x, y, z (and does not gate rotations on their rounded size)
    for ax in ((0, 1, 2) if words else (0, 2, 1)):
        if cm[ax] != 0 and metres[ax] != 0:
            name = _MOVE_WORDS[ax][0 if metres[ax] > 0 else 1]
            parts.append(name if words else f"{name} {_number(cm[ax], sum_decimal)} cm")
    for ax in range(len(rad)):
        if rad[ax] != 0 and (words or deg[ax] != 0):
            name = _ROT_WRITE[ax][0 if rad[ax] > 0 else 1]
            parts.append(name if words else f"{name} {deg[ax]} degrees")
    parts.append("open gripper" if float(arr[-1, 6]) >= 0.5 else "close gripper")
    return ", ".join(parts)


def summarize_bimanual_numeric_actions(arr_like, sum_decimal: str, include_rotation: bool = False) -> str | None:
    """action_text.py:186-210: two 7-D arms side by side."""
    arr = np.asarray(arr_like, dtype=float)
    if arr.ndim == 1:
        arr = arr[None, :]
    if arr.shape[-1] < 14:
        return None
    left, right = arr[..., :7], arr[..., 7:14]
    if sum_decimal == "compact":
        return f"<L {_compact(left, include_rotation)[1:-1]} R {_compact(right, include_rotation)[1:-1]}>"
    ls, rs = (summarize_numeric_actions(x, sum_decimal, include_rotation) for x in (left, right))
    return None if ls is None or rs is None else f"Left arm: {ls}. Right arm: {rs}"


_SCALE_MOVE_RE = re.compile(r"(move\s+(?:forward|back|left|right|up|down))\s+([+\-]?\d+(?:\.\d+)?)\s*cm")
_SCALE_ROT_RE = re.compile(r"((?:tilt\s+(?:left|right|back|forward))|(?:rotate\s+(?:clockwise|counterclockwise)))\s+([+\-]?\d+(?:\.\d+)?)\s*degrees")


def describe_language_action_scale(language_action):
    """action_text.py:144-183: numbers -> "slightly" / "moderately" / "a lot" (<= 3 cm, < 8 cm; < 10 deg, < 30 deg)."""
    if language_action is None:
        return None
    if not isinstance(language_action, str) or not language_action.strip():
        return language_action

    def swap(pattern, small, mid):
        def rep(m):
            v = float(m.group(2))
            return f"{m.group(1)} " + ("slightly" if small(v) else "moderately" if mid(v) else "a lot")
        return lambda text: pattern.sub(rep, text)

    text = swap(_SCALE_MOVE_RE, lambda v: v <= 3.0, lambda v: v < 8.0)(language_action)
    return swap(_SCALE_ROT_RE, lambda v: v < 10.0, lambda v: v < 30.0)(text)


def _sum_matches(pattern, text, table, group_value=2):
    acc = np.zeros(3)
    for m in pattern.finditer(text):
        ax, sign = table[m.group(1).lower()]
        acc[ax] += sign * (float(m.group(group_value)) if m.group(group_value) is not None else 0.0)
    return acc


def is_idle_language_action(language_action, sum_decimal: str, include_rotation: bool = False, translation_threshold: float = 1.0,
                            rotation_threshold_deg: float = 10.0) -> bool:
    """action_text.py:213-302: a label whose total translation is < 1 cm (and rotation < 10 degrees) is filtered out."""
    if not language_action or not isinstance(language_action, str):
        return True
    if sum_decimal == "compact":
        m = (_COMPACT6_RE if include_rotation else _COMPACT3_RE).search(language_action)
        if not m:
            return True
        v = [int(g) for g in m.groups()[:6 if include_rotation else 3]]
        still = np.sqrt(sum(x * x for x in v[:3])) < translation_threshold
        return bool(still and (not include_rotation or np.sqrt(sum(x * x for x in v[3:6])) < rotation_threshold_deg))
    if sum_decimal == "no_number":
        moves = re.search(r"move\s+(right|left|forward|backward|back|up|down)(?!\s+[\d.])", language_action, re.IGNORECASE) is not None
        turns = include_rotation and re.search(rf"({_ROT_WORDS})(?!\s+[\d.])", language_action, re.IGNORECASE) is not None
        return not (moves or turns)
    move_re = re.compile(r"move\s+(right|left|forward|backward|back|up|down)\s+([\d.]+)\s*cm", re.IGNORECASE)
    still = np.sqrt(np.sum(_sum_matches(move_re, language_action, _MOVE_AXIS) ** 2)) < translation_threshold
    if not include_rotation:
        return bool(still)
    return bool(still and np.sqrt(np.sum(_sum_matches(_ROT_RE, language_action, _ROT_AXIS_IDLE) ** 2)) < rotation_threshold_deg)


# ------------------------------------------------------------------------------------------------ text -> numbers
@dataclasses.dataclass(frozen=True)
class LanguageActionFormat:
    """lang_action_formats.py:11-138."""
    name: str
    style: str = "verbose"            # "verbose" | "compact" | "vla0"
    decimal_places: int = 0
    include_rotation: bool = False
    translation_unit: str = "cm"
    use_eef_frame: bool = False

    def get_sum_decimal(self) -> str:
        return "compact" if self.style == "compact" else f"{self.decimal_places}f"

    def parse_language_to_deltas(self, reasoning, *, initial_state=None):
        """-> (movement [dx, dy, dz, droll, dpitch, dyaw] in m / rad, gripper | None).  Unparseable text gives zeros."""
        movement = np.zeros(6, dtype=float)
        gripper = None
        if self.style == "compact":
            m = _COMPACT6_RE.search(reasoning) if self.include_rotation else None   # (the 3-value form is never parsed, as there)
            if m:
                g = m.groups()
                movement[:3] = np.array(g[0:3], dtype=float) / 100.0
                movement[3:6] = np.array(g[3:6], dtype=float) * np.pi / 180.0
                gripper = float(g[-1])
        else:
            text = reasoning.replace("slightly", "1.5 cm").replace("moderately", "5 cm").replace("a lot", "10 cm")
            move_re = re.compile(rf"move\s+(right|left|forward|backward|back|up|down)(?:\s+([\-\d\.]+)\s*{self.translation_unit})?",
                                 re.IGNORECASE)
            movement[:3] = _sum_matches(move_re, text, _MOVE_AXIS) / 100.0
            if self.include_rotation:
                movement[3:6] = _sum_matches(_ROT_RE, text, _ROT_AXIS_PARSE) * np.pi / 180.0
            low = text.lower()
            set_to = re.search(r"set\s+gripper\s+to\s+([\-+]?\d+\.?\d*)", text, re.IGNORECASE)
            if "open gripper" in low:
                gripper = 1.0
            elif "close gripper" in low:
                gripper = 0.0
            elif set_to:
                gripper = float(set_to.group(1))
        if self.use_eef_frame and initial_state is not None:
            movement = transform_actions_from_eef_frame(movement, initial_state)[0]
        return movement, gripper


@dataclasses.dataclass(frozen=True)
class VLA0ActionFormat(LanguageActionFormat):
    """lang_action_formats.py:141-270: actions as integers in [0, num_bins] ("523 127 890 ...")."""
    name: str = "vla0"
    style: str = "vla0"
    num_bins: int = 1000
    action_horizon: int = 1
    action_dim: int = 7

    def get_sum_decimal(self) -> str:
        return "vla0"

    def summarize_actions(self, actions) -> str:
        a = np.asarray(actions, dtype=float)
        a = np.clip(a[None, :] if a.ndim == 1 else a, -1.0, 1.0)
        q = np.clip(np.round((a + 1.0) / 2.0 * self.num_bins).astype(int), 0, self.num_bins)
        return " ".join(map(str, q.flatten()))

    def _grid(self, ints) -> np.ndarray:
        c = np.array(ints, dtype=float) / self.num_bins * 2.0 - 1.0
        n = self.action_horizon * self.action_dim
        c = np.pad(c, (0, n - len(c))) if len(c) < n else c[:n]
        return c.reshape(self.action_horizon, self.action_dim)

    @staticmethod
    def _ints(reasoning):
        if isinstance(reasoning, list):
            reasoning = " ".join(reasoning)
        try:
            return reasoning, [int(x) for x in reasoning.split()]
        except ValueError:
            return reasoning, None

    def parse_language_to_deltas(self, reasoning, *, initial_state=None):
        _, ints = self._ints(reasoning)
        if not ints:
            return np.zeros(6, dtype=float), None
        a = self._grid(ints)
        movement = a[0, :6] if a.shape[1] >= 6 else np.zeros(6)
        return movement, (float(a[0, 6]) if a.shape[1] >= 7 else None)

    def parse_to_full_actions(self, reasoning) -> np.ndarray:
        reasoning, ints = self._ints(reasoning)
        if not re.search(r"([\d\s]+)", reasoning) or not ints:
            logging.info(f"Failed to parse VLA0 format: {reasoning}")
            return np.zeros((self.action_horizon, self.action_dim), dtype=float)
        return self._grid(ints)


VERBOSE_WITH_ROTATION_FORMAT = LanguageActionFormat(name="verbose_with_rotation", include_rotation=True)
VERBOSE_EEF_WITH_ROTATION_FORMAT = LanguageActionFormat(name="verbose_eef_with_rotation", include_rotation=True, use_eef_frame=True)
VLA0_CHUNKED_FORMAT = VLA0ActionFormat(name="vla0_chunked", num_bins=1000, action_horizon=10, action_dim=7)
LANGUAGE_ACTION_FORMAT_REGISTRY = {f.name: f for f in (VERBOSE_WITH_ROTATION_FORMAT, VERBOSE_EEF_WITH_ROTATION_FORMAT, VLA0_CHUNKED_FORMAT)}


def get_language_action_format(name: str) -> LanguageActionFormat:
    if name not in LANGUAGE_ACTION_FORMAT_REGISTRY:
        raise ValueError(f"Unknown language action format: {name}. Available formats: {list(LANGUAGE_ACTION_FORMAT_REGISTRY.keys())}")
    return LANGUAGE_ACTION_FORMAT_REGISTRY[name]


# ------------------------------------------------------------------------------------------------ training labels
@dataclasses.dataclass
class ActionProcessor:
    """policies/transforms/action_processor.py:19-193: raw language actions of a trajectory chunk -> (label text, frame
    description).  End-effector-frame formats rotate the (single, already summed) delta into the gripper camera's frame first;
    `random_base_prob` keeps a share of the wrist-camera samples in the base frame."""
    language_action_format: LanguageActionFormat
    random_base_prob: float = 0.0

    def _frame(self, initial_state, has_wrist_image: bool) -> tuple[bool, str]:
        import random
        eef = self.language_action_format.use_eef_frame and initial_state is not None
        if self.random_base_prob > 0.0:
            eef = eef and has_wrist_image and random.random() < (1 - self.random_base_prob)
        return eef, ("end-effector frame" if eef else "robot base frame")

    def transform_to_frame(self, raw_actions, initial_state, dataset_name: str, rotation_applied: bool, has_wrist_image: bool):
        eef, frame = self._frame(initial_state, has_wrist_image)
        return (transform_actions_to_eef_frame(raw_actions, initial_state, dataset_name, rotation_applied) if eef else raw_actions), frame

    def summarize_language_actions(self, data: dict, lang_action_key: str = "language_actions", initial_state=None,
                                   dataset_name: str | None = None, rotation_applied: bool = False):
        acts, frame = self.transform_to_frame(data[lang_action_key], initial_state, dataset_name, rotation_applied,
                                              data.get("has_wrist_image", False))
        f = self.language_action_format
        if data.get("is_bimanual", False):
            return summarize_bimanual_numeric_actions(acts, f.get_sum_decimal(), f.include_rotation), frame
        if data.get("is_navigation", False):
            return summarize_numeric_actions(acts, "nearest_10", include_rotation=True, rotation_precision=10), frame
        return summarize_numeric_actions(acts, sum_decimal=f.get_sum_decimal(), include_rotation=f.include_rotation), frame

    @staticmethod
    def extract_motion_components(language_actions) -> dict:
        """[dx, dy, dz, droll, dpitch, dyaw, gripper] in m / rad (first row of a chunk) -> cm / degrees."""
        a = np.asarray(language_actions, dtype=float)
        a = a[0] if a.ndim == 2 else a
        deg = lambda i: a[i] * 180.0 / np.pi if len(a) > i else 0.0
        return {"dx_cm": a[0] * 100.0, "dy_cm": a[1] * 100.0, "dz_cm": a[2] * 100.0, "droll_deg": deg(3), "dpitch_deg": deg(4),
                "dyaw_deg": deg(5), "gripper": a[6] if len(a) > 6 else 0.5}


# ------------------------------------------------------------------------------------------------ golden case tables
def case_tables() -> dict:
    """Inputs shared by the golden generator (reference side) and the parity test (this module)."""
    rs = np.random.RandomState(7)
    chunks = [np.round(np.concatenate([rs.uniform(-0.02, 0.02, (T, 3)), rs.uniform(-0.08, 0.08, (T, 3)), rs.uniform(0, 1, (T, 1))], 1), 5).tolist()
              for T in (1, 4, 10, 16)]
    chunks.append([[0.0] * 6 + [1.0]])                                    # idle
    chunks.append([[0.004, -0.004, 0.0, 0.0, 0.0, 0.0, 0.49]])            # rounds to zero at 0 decimals
    chunks.append([[0.051, 0.0, -0.0349, 0.26, -0.09, 0.0, 0.5]])
    chunks.append([[0.3, 0.2, 0.1, 0.0, 0.0, 0.0]])                        # too narrow -> None
    texts = ["move forward 3 cm, move up 1 cm, move right 12 cm, tilt left 10 degrees, rotate clockwise 20 degrees, open gripper",
             "move back 2.5 cm and move down 4 cm, tilt forward 30 degrees, tilt back 10 degrees, close gripper",
             "Move Left slightly, move up moderately, move forward a lot, set gripper to 0.35",
             "move backward 7cm move left", "rotate counterclockwise 15 degrees tilt up 5 degrees tilt down 5 degrees tilt right 20 degrees",
             "<+03 -12 +00 +10 -05 +00 1>", "<-01 +00 +02 0>", "nothing to do", "", "move forward, move up", "tilt left, open gripper",
             "move forward 0.4 cm, tilt left 5 degrees, close gripper", "move right 1 cm", "523 127 890 512 512 512 500",
             "0 1000 500 250 750 100 900 " * 10, "12 x 40", "set gripper to -1"]
    states = [np.round(np.concatenate([rs.uniform(-0.5, 0.5, 3), rs.uniform(-1, 1, 3), [0.7]]), 4).tolist(),
              np.round(np.concatenate([rs.uniform(-0.5, 0.5, 3), rs.normal(size=6), [0.2]]), 4).tolist()]
    datasets = ["", "droid", "jaco_play", "berkeley_autolab_ur5", "furniture_bench_dataset_converted_externally_to_rlds", "utaustin_mutex",
                "austin_buds_dataset_converted_externally_to_rlds", "fmb", "viola"]
    frame_actions = np.round(np.concatenate([rs.uniform(-0.05, 0.05, (3, 3)), rs.uniform(-0.3, 0.3, (3, 3)), rs.uniform(0, 1, (3, 1))], 1), 5).tolist()
    return {"chunks": chunks, "texts": texts, "states": states, "datasets": datasets, "frame_actions": frame_actions,
            "sum_decimals": ["0f", "1f", "2f", "nearest_10", "no_number", "compact"]}


def processor_cases() -> list[dict]:
    rs = np.random.RandomState(11)
    state = np.round(np.concatenate([rs.uniform(-0.5, 0.5, 3), rs.normal(size=6), [0.4]]), 4).tolist()
    one = lambda: np.round(np.concatenate([rs.uniform(-0.06, 0.06, 3), rs.uniform(-0.5, 0.5, 3), rs.uniform(0, 1, 1)]), 5).tolist()
    cases = []
    for fmt in ("verbose_with_rotation", "verbose_eef_with_rotation"):
        for ds in ("droid", "jaco_play", "berkeley_autolab_ur5", "viola", "fmb"):
            for rot_applied in (False, True):
                cases.append({"format": fmt, "dataset": ds, "rotation_applied": rot_applied, "actions": one(), "state": state, "flags": {}})
        cases.append({"format": fmt, "dataset": "droid", "rotation_applied": False, "actions": one(), "state": None, "flags": {}})
    chunk = [one() for _ in range(5)]
    cases.append({"format": "verbose_with_rotation", "dataset": "x", "rotation_applied": False, "actions": chunk, "state": state, "flags": {"is_navigation": True}})
    two = np.concatenate([np.asarray(chunk), np.asarray(chunk)[::-1]], axis=1).tolist()
    cases.append({"format": "verbose_with_rotation", "dataset": "x", "rotation_applied": False, "actions": two, "state": state, "flags": {"is_bimanual": True}})
    return cases
