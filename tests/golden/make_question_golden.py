"""Generates tests/golden/question_v1.json by running the REFERENCE's question / answer synthesis (pure Python + numpy:
lap/policies/question_types.py and the VQA / prediction handlers of lap/policies/transforms/sample_handlers.py) on the case
table of lap_amd/questions.py.  `lap/policies/__init__.py` eagerly imports the openpi-dependent transforms, so the package
object is pre-registered as a bare namespace and the modules are loaded unmodified from /root/reference.  The handlers'
fresh `np.random.default_rng()` is replaced by a seeded generator for the duration of a call so that the draws can be
replayed.  Run in the build container only:  python tests/golden/make_question_golden.py"""
import json
import pathlib
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import lap  # noqa: E402,F401

_pkg = types.ModuleType("lap.policies")
_pkg.__path__ = ["/root/reference/src/lap/policies"]
sys.modules["lap.policies"] = _pkg

from lap.policies import lang_action_formats as ref_fmt                      # noqa: E402
from lap.policies import question_types as ref_q                             # noqa: E402
from lap.policies.transforms import sample_handlers as ref_h                 # noqa: E402
from lap.policies.transforms.action_processor import ActionProcessor as RefProcessor   # noqa: E402

from lap_amd import questions as mine                                        # noqa: E402  (case table only)

T = mine.case_table()
out = {"formats": [], "categorical": [], "embodiment": [], "gripper": [], "sampling": [], "qa": [], "process": [], "vqa": []}
for mi, m in enumerate(T["motions"]):
    for g in T["grippers"]:
        for f in ref_q.AnswerFormat:
            out["formats"].append({"motion": mi, "gripper": g, "format": f.value, "expected": ref_q.format_delta_motion(*m, g, answer_format=f)})
    out["categorical"].append({"motion": mi, "directions": ref_q.compute_dominant_directions(*m[:3]),
                               "directions_2cm": ref_q.compute_dominant_directions(*m[:3], threshold_cm=2.0),
                               "magnitude": ref_q.compute_motion_magnitude(*m[:3])})
out["embodiment"] = [{"dataset": d, "expected": ref_q.get_embodiment_name(d)} for d in T["datasets"]]
out["gripper"] = [{"pair": p, "expected": ref_q.compute_gripper_change(*p)} for p in T["gripper_pairs"]]
cfg = ref_q.QuestionConfig()
plain = ref_q.QuestionConfig(use_diverse_prompts=False)
for seed in T["seeds"]:
    rng = np.random.default_rng(seed)
    qt = cfg.sample_question_type(rng)
    af = cfg.sample_answer_format(rng)
    out["sampling"].append({"seed": seed, "type": qt.value, "format": af.value,
                            "prompts": [cfg.get_prompt_template(k, rng, frame_description=T["frames"][seed % 3]) for k in ref_q.QuestionType],
                            "canonical": [plain.get_prompt_template(k, rng, frame_description=T["frames"][seed % 3]) for k in ref_q.QuestionType]})
proc = RefProcessor(language_action_format=ref_fmt.get_language_action_format("verbose_eef_with_rotation"), random_base_prob=0.0)
handler = ref_h.PredictionSampleHandler(question_config=ref_q.QuestionConfig(type_weights={k.value: 1.0 for k in ref_q.QuestionType}),
                                        action_processor=proc)
rs = np.random.RandomState(5)
for seed in T["seeds"]:
    m = T["motions"][seed % len(T["motions"])]
    motion = dict(zip(("dx_cm", "dy_cm", "dz_cm", "droll_deg", "dpitch_deg", "dyaw_deg"), m), gripper=float(seed % 3) / 2)
    ds = T["datasets"][seed % len(T["datasets"])]
    state0 = np.round(rs.uniform(0, 1, 8), 3)
    for kind in ref_q.QuestionType:
        rng = np.random.default_rng(1000 + seed)
        inputs = {}
        q, a = handler._format_question_answer(data={"prompt": b"zone@pick up the cup" if "r1_lite" in ds else "stack the blocks", "dataset_name": ds},
                                               inputs=inputs, question_type=kind, motion_components=motion, dataset_name=ds,
                                               initial_state=state0, frame_description=T["frames"][seed % 3], rng=rng)
        out["qa"].append({"seed": seed, "kind": kind.value, "motion": motion, "dataset": ds, "state": state0.tolist(), "frame": T["frames"][seed % 3],
                          "question": q, "answer": a, "swap": inputs.get("_temporal_swap")})
# full process(): the handler's own generator replaced by a seeded one
real_default_rng = np.random.default_rng
for seed in T["seeds"][:12]:
    raw = (np.asarray(T["motions"][seed % len(T["motions"])]) * np.array([0.01] * 3 + [np.pi / 180] * 3)).tolist() + [float(seed % 2)]
    state0 = np.round(np.concatenate([rs.uniform(-0.3, 0.3, 3), rs.uniform(-1, 1, 3), [0.3 + 0.5 * (seed % 2)]]), 3)
    data = {"language_actions": raw, "raw_state": state0.tolist(), "has_wrist_image": bool(seed % 2), "prompt": "open the drawer",
            "dataset_name": T["datasets"][seed % len(T["datasets"])]}
    inputs = {"image": {"base_0_rgb": "B", "left_wrist_0_rgb": "W"}, "image_mask": {"base_0_rgb": True, "left_wrist_0_rgb": False}, "prompt": "default"}
    np.random.default_rng = lambda *a, _s=seed, **k: real_default_rng(77 + _s)
    try:
        res = ref_h.PredictionSampleHandler(question_config=ref_q.QuestionConfig(), action_processor=proc).process(
            dict(data), inputs, data["dataset_name"], rotation_applied=bool(seed % 3 == 0))
    finally:
        np.random.default_rng = real_default_rng
    out["process"].append({"seed": seed, "data": data, "rotation_applied": bool(seed % 3 == 0),
                           "result": {k: res[k] for k in ("prompt", "language_actions", "frame_description", "sample_mask", "image", "image_mask")}})
out["process_no_actions"] = ref_h.PredictionSampleHandler(question_config=ref_q.QuestionConfig(), action_processor=proc).process(
    {"prompt": "x"}, {"prompt": "p"}, "droid", False)
for cap in (b"a red cup on a table", "two robots", None):
    d = {} if cap is None else {"caption": cap}
    out["vqa"].append({"caption": cap.decode() if isinstance(cap, bytes) else cap, "bytes": isinstance(cap, bytes),
                       "result": ref_h.VQASampleHandler().process(d, {"prompt": "describe", "sample_mask": False})})
path = pathlib.Path(__file__).with_name("question_v1.json")
path.write_text(json.dumps(out, indent=0, default=lambda o: o.tolist() if isinstance(o, np.ndarray) else (bool(o) if isinstance(o, np.bool_) else float(o))))
print(f"wrote {path}: " + ", ".join(f"{k} {len(v)}" for k, v in out.items() if isinstance(v, list)))
