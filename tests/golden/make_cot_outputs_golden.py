"""Generates tests/golden/cot_outputs_v1.json: the REFERENCE's `CoTOutputs` (policies/transforms/output_transforms.py:20-214), imported
unmodified (`lap.policies` registered as a bare namespace so that its openpi-dependent `__init__` does not run): flow-matching outputs
passing through, decoded language actions parsed back to deltas in the base and in the end-effector frame (rotated with the request's
raw state), a text without gripper command, and the VLA-0 strategy with the three kinds of un-normalisation (statistics as objects with
attributes, which is how the reference reads them).  Run in the build container only."""
import json
import pathlib
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference/src")
import lap  # noqa: E402,F401

pkg = types.ModuleType("lap.policies"); pkg.__path__ = ["/root/reference/src/lap/policies"]; sys.modules["lap.policies"] = pkg
from lap.policies import lang_action_formats as fmt  # noqa: E402
from lap.policies.transforms import output_transforms as ref_out  # noqa: E402

rng = np.random.default_rng(3)
th = 0.6
R0 = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
state = np.concatenate([[0.3, -0.1, 0.25], R0[:, :2].T.reshape(-1), [0.7]])       # xyz, rot6d (first two columns), gripper
texts = ["move forward 4 cm, move up 2 cm, move right 1 cm, rotate clockwise 10 degrees, open gripper", "move left 5 cm and move down 3 cm and close gripper",
         "move back 2 cm", "tilt up 5 degrees, roll left 3 degrees, open gripper", "nonsense text"]
lo = rng.normal(size=7) - 1; hi = lo + np.abs(rng.normal(size=7)) + 0.2
st = types.SimpleNamespace(q01=lo, q99=hi, min=lo - 0.3, max=hi + 0.3, mean=rng.normal(size=7), std=np.abs(rng.normal(size=7)) + 0.1)
jl = lambda a: None if a is None else np.asarray(a, dtype=np.float64).tolist()
out = {"state": jl(state), "stats": {k: jl(getattr(st, k)) for k in ("q01", "q99", "min", "max", "mean", "std")}, "cases": []}
out["cases"].append({"kind": "flow", "actions": jl(rng.normal(size=(3, 7))), })
out["cases"][-1]["out"] = {"actions": out["cases"][-1]["actions"], "reasoning": None}
assert ref_out.CoTOutputs("verbose_eef_with_rotation")({"actions": np.asarray(out["cases"][0]["actions"])})["reasoning"] is None
names = [n for n in ("verbose", "verbose_with_rotation", "verbose_eef", "verbose_eef_with_rotation", "compact", "compact_with_rotation") if n in fmt.LANGUAGE_ACTION_FORMAT_REGISTRY] \
    if hasattr(fmt, "LANGUAGE_ACTION_FORMAT_REGISTRY") else ["verbose_eef_with_rotation", "verbose_with_rotation"]
for name in names:
    for ti, t in enumerate(texts):
        for with_state in (False, True):
            data = {"actions": np.zeros((1, 7)), "reasoning": t, **({"raw_state": state} if with_state else {})}
            try:
                r = ref_out.CoTOutputs(name)(data)
                out["cases"].append({"kind": "text", "format": name, "text": ti, "with_state": with_state, "out": {"actions": jl(r["actions"]), "reasoning": r["reasoning"]}})
            except Exception as e:
                out["cases"].append({"kind": "text", "format": name, "text": ti, "with_state": with_state, "error": type(e).__name__})
out["texts"] = texts
# VLA-0: a full grid of integers decoded and un-normalised
v = fmt.VLA0_CHUNKED_FORMAT
acts = rng.uniform(-1, 1, size=(v.action_horizon if hasattr(v, "action_horizon") else 10, 7))
text = v.summarize_actions(acts)
for ntype in ("bounds_q99", "bounds", "normal", "other"):
    r = ref_out.CoTOutputs(v, norm_stats={"actions": st}, normalization_type=ntype, transform_strategy="vla0")({"actions": np.zeros((1, 7)), "reasoning": text})
    out["cases"].append({"kind": "vla0", "ntype": ntype, "text": text, "out": {"actions": jl(r["actions"]), "reasoning": r["reasoning"]}})
r = ref_out.CoTOutputs(v, norm_stats=None, transform_strategy="vla0")({"actions": np.zeros((1, 7)), "reasoning": text})
out["cases"].append({"kind": "vla0", "ntype": None, "text": text, "out": {"actions": jl(r["actions"]), "reasoning": r["reasoning"]}})
pathlib.Path(__file__).with_name("cot_outputs_v1.json").write_text(json.dumps(out))
print("wrote cot_outputs_v1.json:", len(out["cases"]), "cases;", names, sum("error" in c for c in out["cases"]), "rejected")
