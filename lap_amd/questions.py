"""Question / answer synthesis for the auxiliary "prediction" samples and the VQA sample handler — the part of the
training-sample pipeline that feeds `LAP.compute_loss`'s prediction / VQA loss terms (lap.py:401-413,472-545).

Restates `src/lap/policies/question_types.py` (question kinds, prompt pools, the nine answer formats of a delta motion,
the categorical answers, `QuestionConfig` sampling) and the VQA / prediction handlers of
`src/lap/policies/transforms/sample_handlers.py:44-326`.  PARITY PINNED: the reference modules are pure Python / numpy and
import in the build container; `tests/golden/make_question_golden.py` runs them on the case table below with seeded
generators and `tests/test_questions_cpu.py` replays `tests/golden/question_v1.json` through this module with string
equality (the random draws use the same numpy Generator calls in the same order, so a seeded generator gives the
reference's choices).

Everything here is host-side text: the formatter is one table-driven routine — axis tables say which word a signed
component maps to in which vocabulary, format specs say how magnitudes are rendered.
"""
from __future__ import annotations

import dataclasses
import enum
import json

import numpy as np


class QuestionType(enum.Enum):       # question_types.py:16-38
    DELTA_MOTION = "delta_motion"
    TASK_PREDICTION = "task_prediction"
    DIRECTION_CLASSIFICATION = "direction_classification"
    GRIPPER_PREDICTION = "gripper_prediction"
    MAGNITUDE_ESTIMATION = "magnitude_estimation"
    TEMPORAL_ORDERING = "temporal_ordering"
    EMBODIMENT_IDENTIFICATION = "embodiment_identification"


class AnswerFormat(enum.Enum):       # question_types.py:41-52
    VERBOSE = "verbose"
    VERBOSE_WITH_ROTATION = "verbose_with_rotation"
    COMPACT = "compact"
    COMPACT_WITH_ROTATION = "compact_with_rotation"
    QUALITATIVE = "qualitative"
    COMPONENT = "component"
    JSON = "json"
    SENTENCE = "sentence"
    DIRECTION_ONLY = "direction_only"


# Prompt pools (question_types.py:61-124): the wording IS the training data, so the strings are the reference's.
PROMPTS: dict[QuestionType, tuple[str, ...]] = {
    QuestionType.DELTA_MOTION: (
        "Describe the robot's motion between these two frames{frame_ref}",
        "What movement did the robot make from the first image to the second{frame_ref}?",
        "Predict the change in robot position shown in these images{frame_ref}",
        "Given these before and after images, what action was taken{frame_ref}?",
        "Analyze the visual difference and describe the robot's movement{frame_ref}",
        "What is the delta motion between these two images{frame_ref}?",
        "Describe how the robot end-effector moved between frames{frame_ref}",
        "What movement occurred between these two observations{frame_ref}?",
        "Characterize the robot motion from the image pair{frame_ref}",
        "From image 1 to image 2, describe the robot's action{frame_ref}"),
    QuestionType.TASK_PREDICTION: (
        "What task is the robot performing given this motion: {action}?",
        "Based on the action '{action}', what is the robot trying to accomplish?",
        "Given the robot moved as follows: {action}, what is the task?",
        "Identify the task from this robot motion: {action}",
        "The robot performed: {action}. What task does this correspond to?",
        "What goal is the robot working towards with this action: {action}?"),
    QuestionType.DIRECTION_CLASSIFICATION: (
        "What is the dominant motion direction shown in these images?",
        "In which direction(s) did the robot primarily move?",
        "Classify the main movement direction between these frames",
        "What are the primary motion axes in this image pair?",
        "Describe the dominant direction of robot movement"),
    QuestionType.GRIPPER_PREDICTION: (
        "Did the gripper open, close, or stay the same between these images?",
        "What happened to the gripper state?",
        "Predict the gripper state change from image 1 to image 2",
        "How did the gripper position change?",
        "Was there a gripper action between these frames?"),
    QuestionType.MAGNITUDE_ESTIMATION: (
        "How much did the robot move between these images?",
        "Estimate the magnitude of the robot's motion",
        "Is the movement between these frames small, moderate, or large?",
        "Characterize the distance traveled by the robot",
        "What is the scale of the robot's displacement?"),
    QuestionType.TEMPORAL_ORDERING: (
        "Given the robot action '{action}', which image shows the earlier state - the first or second image?",
        "The robot performed: {action}. In what order do these images appear in the trajectory?",
        "Between these frames the robot did: {action}. Which frame came first chronologically?",
        "Given the motion '{action}', determine the temporal order of these two observations",
        "The robot moved as follows: {action}. Is image 1 before or after image 2 in the sequence?"),
    QuestionType.EMBODIMENT_IDENTIFICATION: (
        "What robot or dataset is this image from?",
        "Identify the robot embodiment shown in this image",
        "What type of robot is performing this task?",
        "Which dataset does this observation come from?",
        "Classify the robot platform shown here"),
}

# ---- vocabularies: (component index into (dx, dy, dz) or (roll, pitch, yaw), word for > 0, word for < 0)
_T_VERBOSE = ((0, "move forward", "move back"), (2, "move up", "move down"), (1, "move left", "move right"))      # x, z, y order
_T_PLAIN = ((0, "forward", "backward"), (1, "left", "right"), (2, "up", "down"))
_R_WORDS = ((0, "tilt left", "tilt right"), (1, "tilt back", "tilt forward"), (2, "rotate counterclockwise", "rotate clockwise"))
_SMALL_NUMBERS = ("zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen "
                  "seventeen eighteen nineteen twenty").split()


def _nearest(value: float, step: int) -> int:
    return int(round(value / step) * step)


def _pick(value: float, pos: str, neg: str) -> str:
    return pos if value > 0 else neg


def _grade(value: float, lo: float, hi: float) -> str:
    a = abs(value)
    return "slightly" if a < lo else ("moderately" if a < hi else "significantly")


def _verbose(t, r, grip, rot, decimals=0):     # question_types.py:169-231
    out = []
    for i, pos, neg in _T_VERBOSE:
        mag = round(abs(t[i]), decimals)
        if t[i] != 0 and mag != 0:
            out.append(f"{_pick(t[i], pos, neg)} {mag:.{decimals}f} cm")
    if rot:
        for i, pos, neg in _R_WORDS:
            mag = _nearest(abs(r[i]), 10)
            if r[i] != 0 and mag != 0:
                out.append(f"{_pick(r[i], pos, neg)} {mag} degrees")
    if grip:
        out.append(grip)
    return ", ".join(out) if out else "no movement"


def _compact(t, r, grip, rot):                 # question_types.py:234-262
    cells = [f"{int(round(v)):+03d}" for v in t]
    if rot:
        cells += [f"{_nearest(v, 5):+03d}" for v in r]
    cells.append("1" if "open" in grip.lower() else "0")
    return "<" + " ".join(cells) + ">"


def _qualitative(t, r, grip, rot):             # question_types.py:265-335
    out = []
    moves = [f"{_grade(t[i], 1.5, 5)} {_pick(t[i], pos, neg)}" for i, pos, neg in _T_PLAIN if abs(t[i]) >= 0.5]
    if moves:
        out.append("move " + " and ".join(moves))
    if rot:
        turns = [f"{_grade(r[i], 10, 30)} {_pick(r[i], pos, neg)}" for i, pos, neg in _R_WORDS if abs(r[i]) >= 5]
        if turns:
            out.append(" and ".join(turns))
    if grip:
        out.append(f"then {grip}" if out else grip)
    return ", ".join(out) if out else "remain stationary"


def _component(t, r, grip, rot, decimals=1):   # question_types.py:338-369
    out = ["translation: ({}, {}, {}) cm".format(*(round(v, decimals) for v in t))]
    if rot:
        out.append("rotation: ({}, {}, {}) deg".format(*(round(v, decimals) for v in r)))
    if grip:
        out.append(f"gripper: {grip}")
    return "; ".join(out)


def _json(t, r, grip, rot, decimals=1):        # question_types.py:372-403
    d = {k: round(v, decimals) for k, v in zip(("dx_cm", "dy_cm", "dz_cm"), t)}
    if rot:
        d.update({k: round(v, decimals) for k, v in zip(("droll_deg", "dpitch_deg", "dyaw_deg"), r)})
    if grip:
        d["gripper"] = grip
    return json.dumps(d)


def _sentence(t, r, grip, rot):                # question_types.py:406-456
    legs = []
    for i, pos, neg in _T_PLAIN:
        n = int(round(abs(t[i])))
        if n >= 1:
            word = _SMALL_NUMBERS[n] if n < len(_SMALL_NUMBERS) else str(n)
            legs.append(f"{_pick(t[i], pos, neg)} by {word} centimeter{'' if n == 1 else 's'}")
    if not legs:
        s = "The robot remained stationary"
    elif len(legs) <= 2:
        s = "The robot moved " + " and ".join(legs)
    else:
        s = "The robot moved " + ", ".join(legs[:-1]) + ", and " + legs[-1]
    s += {"open gripper": " while opening the gripper", "close gripper": " while closing the gripper"}.get(grip, "")
    return s + "."


def _direction_only(t, r, grip, rot):          # question_types.py:459-493
    out = [_pick(t[i], pos, neg) for i, pos, neg in _T_PLAIN if abs(t[i]) >= 0.5]
    if rot:
        out += [_pick(r[i], pos, neg) for i, pos, neg in _R_WORDS if abs(r[i]) >= 5]
    if grip:
        out.append(grip)
    return ", ".join(out) if out else "no movement"


_ROT_ALWAYS = {AnswerFormat.VERBOSE_WITH_ROTATION, AnswerFormat.COMPACT_WITH_ROTATION}
_ROT_IF_LARGE = {AnswerFormat.COMPONENT, AnswerFormat.JSON, AnswerFormat.QUALITATIVE}
_FORMATTERS = {AnswerFormat.VERBOSE: _verbose, AnswerFormat.VERBOSE_WITH_ROTATION: _verbose, AnswerFormat.COMPACT: _compact,
               AnswerFormat.COMPACT_WITH_ROTATION: _compact, AnswerFormat.QUALITATIVE: _qualitative, AnswerFormat.COMPONENT: _component,
               AnswerFormat.JSON: _json, AnswerFormat.SENTENCE: _sentence, AnswerFormat.DIRECTION_ONLY: _direction_only}


def format_delta_motion(dx_cm, dy_cm, dz_cm, droll_deg=0, dpitch_deg=0, dyaw_deg=0, gripper_action: str = "",
                        answer_format: AnswerFormat = AnswerFormat.VERBOSE) -> str:
    """question_types.py:697-750: rotation is shown by the *_with_rotation formats always, by component / json / qualitative
    when some angle reaches 5 degrees, by the other formats never."""
    t, r = (dx_cm, dy_cm, dz_cm), (droll_deg, dpitch_deg, dyaw_deg)
    rot = answer_format in _ROT_ALWAYS or (answer_format in _ROT_IF_LARGE and any(abs(a) >= 5 for a in r))
    return _FORMATTERS.get(answer_format, _verbose)(t, r, gripper_action, rot)


# ---- categorical answers (question_types.py:501-590)
def compute_dominant_directions(dx_cm, dy_cm, dz_cm, threshold_cm: float = 1.0) -> str:
    hits = [_pick(v, pos, neg) for v, (_, pos, neg) in zip((dx_cm, dy_cm, dz_cm), _T_PLAIN) if abs(v) > threshold_cm]
    return " and ".join(hits) if hits else "stationary"


def compute_gripper_change(gripper_start: float, gripper_end: float) -> str:
    was_open, is_open = gripper_start > 0.5, gripper_end > 0.5
    return "unchanged" if was_open == is_open else ("opened" if is_open else "closed")


def compute_motion_magnitude(dx_cm, dy_cm, dz_cm) -> str:
    d = float(np.sqrt(dx_cm ** 2 + dy_cm ** 2 + dz_cm ** 2))
    return "small movement" if d < 2.0 else ("moderate movement" if d < 6.0 else "large movement")


_EMBODIMENTS = (("droid", "DROID (Franka Panda)"), ("bridge", "Bridge (WidowX)"), ("bridge_dataset", "Bridge (WidowX)"),
                ("fractal", "Fractal (Google Robot)"), ("rt_1_x", "RT-1 (Google Robot)"), ("kuka", "KUKA Robot"),
                ("fmb", "FMB (Franka Manipulation Benchmark)"), ("taco_play", "TACO Play"), ("jaco_play", "Jaco Play (Kinova Jaco)"),
                ("berkeley_autolab_ur5", "Berkeley Autolab (UR5)"), ("furniture_bench", "Furniture Bench (Franka)"),
                ("austin_buds", "Austin BUDS (Franka)"), ("austin_sirius", "Austin Sirius (Franka)"),
                ("austin_sailor", "Austin Sailor (Franka)"), ("utaustin_mutex", "UT Austin MUTEX (Franka)"), ("viola", "VIOLA (Franka)"),
                ("cmu_stretch", "CMU Stretch (Hello Robot)"), ("dobbe", "DOBBE (Hello Robot)"),
                ("iamlab_cmu_pickup_insert", "CMU IAM Lab (Franka)"))


def get_embodiment_name(dataset_name: str) -> str:
    low = dataset_name.lower()
    return next((label for key, label in _EMBODIMENTS if key in low), dataset_name)


# ---- sampling (question_types.py:598-694)
_DEFAULT_TYPE_WEIGHTS = {"delta_motion": 0.55, "task_prediction": 0.15, "direction_classification": 0.15, "gripper_prediction": 0.05,
                         "magnitude_estimation": 0.05, "temporal_ordering": 0.05}
_DEFAULT_FORMAT_WEIGHTS = {"verbose": 0.35, "verbose_with_rotation": 0.15, "qualitative": 0.2, "compact": 0.0,
                           "compact_with_rotation": 0.05, "component": 0.08, "json": 0.05, "sentence": 0.05, "direction_only": 0.02}


def _draw(rng, table: dict):
    keys = list(table)
    w = np.array([table[k] for k in keys])
    return rng.choice(keys, p=w / w.sum())


@dataclasses.dataclass
class QuestionConfig:
    type_weights: dict | None = None
    delta_motion_format_weights: dict | None = None
    use_diverse_prompts: bool = True

    def __post_init__(self):
        if self.type_weights is None:
            self.type_weights = dict(_DEFAULT_TYPE_WEIGHTS)
        if self.delta_motion_format_weights is None:
            self.delta_motion_format_weights = dict(_DEFAULT_FORMAT_WEIGHTS)

    def sample_question_type(self, rng=None) -> QuestionType:
        return QuestionType(_draw(rng if rng is not None else np.random.default_rng(), self.type_weights))

    def sample_answer_format(self, rng=None) -> AnswerFormat:
        return AnswerFormat(_draw(rng if rng is not None else np.random.default_rng(), self.delta_motion_format_weights))

    def get_prompt_template(self, question_type: QuestionType, rng=None, frame_description: str = "") -> str:
        rng = rng if rng is not None else np.random.default_rng()
        pool = PROMPTS.get(question_type, PROMPTS[QuestionType.DELTA_MOTION])
        text = str(rng.choice(pool)) if self.use_diverse_prompts else pool[0]
        if question_type == QuestionType.DELTA_MOTION and "{frame_ref}" in text:
            text = text.format(frame_ref=f" (in {frame_description})" if frame_description else "")
        return text


# ================================================================================= sample handlers
def decode_text(value, default: str = "") -> str:     # transforms/text_utils.py:7-22
    if isinstance(value, bytes):
        return value.decode("utf-8")
    return value if isinstance(value, str) else default


def parse_prompt(data: dict) -> str:                  # transforms/text_utils.py:36-60 (incl. the r1_lite rule)
    prompt = data.get("prompt")
    assert prompt is not None, "Prompt missing from data"
    text = decode_text(prompt)
    if "r1_lite" in decode_text(data.get("dataset_name")):
        text = text.split("@")[-1]
    return text


@dataclasses.dataclass
class VQASampleHandler:
    """sample_handlers.py:44-69: the caption is the language-action text; VQA samples are never idle."""
    enable_diverse_questions: bool = False

    def process(self, data: dict, inputs: dict) -> dict:
        cap = data.get("caption")
        inputs["language_actions"] = "" if cap is None else decode_text(cap)
        inputs["sample_mask"] = True
        return inputs


@dataclasses.dataclass
class PredictionSampleHandler:
    """sample_handlers.py:72-326: a question about the motion between two frames and its answer, drawn per sample.
    `rng` (an addition): a numpy Generator for reproducible draws; the reference draws from a fresh default_rng()."""
    question_config: QuestionConfig
    action_processor: object            # lap_amd.lang_actions.ActionProcessor
    rng: object = None

    def process(self, data: dict, inputs: dict, dataset_name: str, rotation_applied: bool) -> dict:
        raw = data.get("language_actions")
        if raw is None:
            inputs["sample_mask"] = True
            return inputs
        raw = np.asarray(raw, dtype=float)
        state0 = np.asarray(data.get("raw_state", np.zeros(10)))
        acts, frame = self.action_processor.transform_to_frame(raw, state0, dataset_name, rotation_applied, data.get("has_wrist_image", False))
        motion = self.action_processor.extract_motion_components(acts)
        rng = self.rng if self.rng is not None else np.random.default_rng()
        kind = self.question_config.sample_question_type(rng)
        prompt, answer = self.format_question_answer(data, inputs, kind, motion, dataset_name, state0, frame, rng)
        if kind == QuestionType.TEMPORAL_ORDERING and inputs.get("_temporal_swap", False):
            self._swap_first_two_images(inputs)
        inputs.pop("_temporal_swap", None)
        inputs.update(prompt=prompt, language_actions=answer, frame_description=frame, sample_mask=True)
        return inputs

    def format_question_answer(self, data, inputs, kind: QuestionType, motion: dict, dataset_name: str, state0, frame: str, rng):
        cfg = self.question_config
        m = [motion[k] for k in ("dx_cm", "dy_cm", "dz_cm", "droll_deg", "dpitch_deg", "dyaw_deg")]
        grip = "open gripper" if motion["gripper"] >= 0.5 else "close gripper"
        if kind == QuestionType.TASK_PREDICTION:      # inverse question: the answer is the task prompt
            q = cfg.get_prompt_template(kind, rng).format(action=format_delta_motion(*m, grip, answer_format=AnswerFormat.VERBOSE))
            return q, parse_prompt(data)
        if kind == QuestionType.DIRECTION_CLASSIFICATION:
            return cfg.get_prompt_template(kind, rng), compute_dominant_directions(*m[:3])
        if kind == QuestionType.GRIPPER_PREDICTION:
            start = state0[6] if len(state0) > 6 else 0.5
            return cfg.get_prompt_template(kind, rng), compute_gripper_change(start, motion["gripper"])
        if kind == QuestionType.MAGNITUDE_ESTIMATION:
            return cfg.get_prompt_template(kind, rng), compute_motion_magnitude(*m[:3])
        if kind == QuestionType.TEMPORAL_ORDERING:    # the two frames may be swapped; the answer says which came first
            q = cfg.get_prompt_template(kind, rng).format(action=format_delta_motion(*m, grip, answer_format=AnswerFormat.VERBOSE))
            swap = bool(rng.random() < 0.5)
            inputs["_temporal_swap"] = swap
            return q, ("second" if swap else "first")
        if kind == QuestionType.EMBODIMENT_IDENTIFICATION:
            return cfg.get_prompt_template(kind, rng), get_embodiment_name(dataset_name)
        fmt = cfg.sample_answer_format(rng)           # DELTA_MOTION (also the fallback)
        prompt = cfg.get_prompt_template(QuestionType.DELTA_MOTION, rng, frame_description=frame)
        return prompt, format_delta_motion(*m, grip, answer_format=fmt)

    @staticmethod
    def _swap_first_two_images(inputs: dict) -> None:
        imgs = inputs.get("image")
        if not imgs or len(imgs) < 2:
            return
        k0, k1 = list(imgs)[:2]
        imgs[k0], imgs[k1] = imgs[k1], imgs[k0]
        masks = inputs.get("image_mask")
        if masks is not None:
            masks[k0], masks[k1] = masks[k1], masks[k0]


def case_table() -> dict:
    """Inputs shared by the golden generator (reference side) and the parity test (this module)."""
    rs = np.random.RandomState(11)
    motions = [[0.0] * 6, [3.2, -1.4, 0.6, 12.0, -4.0, 31.0], [-0.4, 0.49, -0.51, 4.9, 5.0, -5.1], [1.5, 5.0, -21.0, -95.0, 10.0, 0.0],
               [0.5, -0.5, 1.0, 0.0, 0.0, 0.0], [-2.5, 2.5, 0.05, 15.0, -25.0, 35.0], [6.0, 0.0, 0.0, 0.0, 0.0, -7.5], [1.49, -4.99, 20.4, 9.99, 29.9, -30.0]]
    motions += np.round(np.concatenate([rs.uniform(-8, 8, (12, 3)), rs.uniform(-40, 40, (12, 3))], 1), 3).tolist()
    return {"motions": motions, "grippers": ["open gripper", "close gripper", ""],
            "datasets": ["droid", "bridge_dataset", "fractal20220817_data", "austin_sirius_dataset_converted_externally_to_rlds", "my_new_robot",
                         "utaustin_mutex", "KUKA", "r1_lite_pick"],
            "gripper_pairs": [[0.2, 0.8], [0.8, 0.2], [0.5, 0.5], [0.5, 0.51], [0.51, 0.5], [0.9, 0.95]],
            "frames": ["", "robot base frame", "end-effector frame"], "seeds": list(range(24))}
