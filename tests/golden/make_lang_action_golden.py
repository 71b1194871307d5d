"""Generates tests/golden/lang_action_v1.json by running the REFERENCE's language-action modules (numpy / scipy only) on the
case tables of lap_amd/lang_actions.py.  `lap/policies/__init__.py` eagerly imports the openpi-dependent input transforms, so
the package object is pre-registered as a bare namespace; the three modules exercised here are loaded unmodified from
/root/reference.  Run in the build container only:  python tests/golden/make_lang_action_golden.py"""
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
from lap.policies.transforms import action_text as ref_text                  # noqa: E402
from lap.policies.transforms import frame_transforms as ref_frame            # noqa: E402
from lap.policies.transforms.action_processor import ActionProcessor as RefProcessor   # noqa: E402

from lap_amd import lang_actions as mine                                     # noqa: E402  (case tables only)

T = mine.case_tables()
js = lambda x: None if x is None else (x.tolist() if isinstance(x, np.ndarray) else x)
out = {"summaries": [], "bimanual": [], "scale": [], "idle": [], "parse": [], "vla0": [], "frames": []}
for ci, chunk in enumerate(T["chunks"]):
    for sd in T["sum_decimals"]:
        for rot in (False, True):
            out["summaries"].append({"chunk": ci, "sum_decimal": sd, "rot": rot, "expected": ref_text.summarize_numeric_actions(chunk, sd, rot)})
    if len(chunk[0]) >= 7:
        two = np.concatenate([np.asarray(chunk), np.asarray(chunk)[::-1] * 0.5], axis=1).tolist()
        for sd in ("0f", "compact"):
            out["bimanual"].append({"chunk": ci, "sum_decimal": sd, "expected": ref_text.summarize_bimanual_numeric_actions(two, sd, True)})
texts = T["texts"] + [s["expected"] for s in out["summaries"] if isinstance(s["expected"], str)][::7]
out["texts"] = texts
for ti, t in enumerate(texts):
    out["scale"].append({"text": ti, "expected": ref_text.describe_language_action_scale(t)})
    for sd in ("0f", "no_number", "compact"):
        for rot in (False, True):
            out["idle"].append({"text": ti, "sum_decimal": sd, "rot": rot, "expected": bool(ref_text.is_idle_language_action(t, sd, rot))})
    for name in ("verbose_with_rotation", "verbose_eef_with_rotation"):
        for si in (None, 0, 1):
            mv, g = ref_fmt.get_language_action_format(name).parse_language_to_deltas(
                t, initial_state=None if si is None else np.asarray(T["states"][si]))
            out["parse"].append({"text": ti, "format": name, "state": si, "movement": js(mv), "gripper": g})
    compact = ref_fmt.LanguageActionFormat(name="c", style="compact", include_rotation=True)
    mv, g = compact.parse_language_to_deltas(t)
    out["parse"].append({"text": ti, "format": "compact_rot", "state": None, "movement": js(mv), "gripper": g})
    v = ref_fmt.VLA0_CHUNKED_FORMAT
    mv, g = v.parse_language_to_deltas(t)
    out["vla0"].append({"text": ti, "movement": js(mv), "gripper": g, "full": js(v.parse_to_full_actions(t))})
acts = np.asarray(T["chunks"][2])[:, :7]
out["vla0_summary"] = {"expected": ref_fmt.VLA0_CHUNKED_FORMAT.summarize_actions(acts * 20), "single": ref_fmt.VLA0ActionFormat().summarize_actions(acts[0] * 20)}
for ds in T["datasets"]:
    for si, st in enumerate(T["states"]):
        for ai, a in enumerate(T["frame_actions"]):
            rec = {"dataset": ds, "state": si, "action": ai, "from_eef": js(ref_frame.transform_actions_from_eef_frame(np.asarray(a), np.asarray(st), ds))}
            if si == 1:   # to_eef reads a rot6d state
                for wrist in (False, True):
                    rec[f"to_eef_{int(wrist)}"] = js(ref_frame.transform_actions_to_eef_frame(np.asarray(a), np.asarray(st), ds, wrist))
            out["frames"].append(rec)
out["processor"] = []
for c in mine.processor_cases():
    proc = RefProcessor(language_action_format=ref_fmt.get_language_action_format(c["format"]))
    text, frame = proc.summarize_language_actions({"language_actions": np.asarray(c["actions"]), **c["flags"]}, "language_actions",
                                                  None if c["state"] is None else np.asarray(c["state"]), c["dataset"], c["rotation_applied"])
    out["processor"].append({"text": text, "frame": frame, "motion": {k: float(v) for k, v in RefProcessor.extract_motion_components(np.asarray(c["actions"])).items()}})
out["from_eef_chunk"] = js(ref_frame.transform_actions_from_eef_frame(np.asarray(T["frame_actions"]), np.asarray([T["states"][0]])))
out["rot6d"] = js(ref_frame.rot6d_to_rotmat(np.asarray(T["states"][1])[3:9]))
path = pathlib.Path(__file__).with_name("lang_action_v1.json")
path.write_text(json.dumps(out, indent=0))
print({k: (len(v) if isinstance(v, list) else "-") for k, v in out.items()}, "->", path, path.stat().st_size, "bytes")
