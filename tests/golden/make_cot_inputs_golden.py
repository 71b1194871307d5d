"""Generates tests/golden/cot_inputs_v1.json: the REFERENCE's `CoTInputs` transform (policies/transforms/input_transforms.py:25-249,
with its image / text / sample handlers), imported UNMODIFIED and run end to end on inference requests and training samples.
What it imports from packages absent here is provided as stand-in modules first: `openpi.transforms` (`DataTransformFn`, an empty base
class; `pad_to_dim`, openpi's three-line zero padding restated), `openpi.models.model.ModelType` (an enum with the one member the
transform names), `lap.datasets.utils.helpers.ActionEncoding` and `lap.models.model_adapter.{IMAGE_KEYS, ExtendedModelType}` —
those three compiled from their own definitions in the reference files (whose modules import TensorFlow / JAX).  `lap.policies` is
registered as a bare namespace so that its `__init__` (openpi-dependent) does not run.  np.random is seeded per case.  Images are
small (8 x 8) so that the fixture stays small; outputs are stored as nested lists.  Run in the build container only."""
import ast
import enum
import json
import pathlib
import random
import sys
import types

import numpy as np

REF = pathlib.Path("/root/reference/src/lap")
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import lap  # noqa: E402,F401


def compile_nodes(path, want, **extra):
    tree = ast.parse((REF / path).read_text())
    ns = {"Enum": enum.Enum, "IntEnum": enum.IntEnum, **extra}
    for n in tree.body:
        name = getattr(n, "name", None) or next((t.id for t in getattr(n, "targets", []) if isinstance(t, ast.Name)), None)
        if name in want:
            exec(compile(ast.Module([n], []), "<ref>", "exec"), ns)
    return ns


def pad_to_dim(x, target_dim, axis=-1):     # openpi.transforms.pad_to_dim
    cur = x.shape[axis]
    if cur < target_dim:
        w = [(0, 0)] * len(x.shape)
        w[axis] = (0, target_dim - cur)
        return np.pad(x, w)
    return x


def mod(name, **attrs):
    m = sys.modules.setdefault(name, types.ModuleType(name))
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


mod("openpi"); mod("openpi.models")
mod("openpi.transforms", DataTransformFn=type("DataTransformFn", (), {}), pad_to_dim=pad_to_dim)
mod("openpi.models.model", ModelType=enum.Enum("ModelType", {"PI0": "pi0", "PI0_FAST": "pi0_fast", "PI05": "pi05"}))
sys.modules["openpi"].transforms = sys.modules["openpi.transforms"]
h = compile_nodes("datasets/utils/helpers.py", {"ActionEncoding"})
mod("lap.datasets"); mod("lap.datasets.utils"); mod("lap.datasets.utils.helpers", ActionEncoding=h["ActionEncoding"])
a = compile_nodes("models/model_adapter.py", {"IMAGE_KEYS", "ExtendedModelType"}, _model=sys.modules["openpi.models.model"])
mod("lap.models"); mod("lap.models.model_adapter", IMAGE_KEYS=a["IMAGE_KEYS"], ExtendedModelType=a["ExtendedModelType"])
pkg = mod("lap.policies"); pkg.__path__ = [str(REF / "policies")]
from lap.policies.transforms import input_transforms as ref_in  # noqa: E402

rng = np.random.default_rng(11)
img = lambda: rng.integers(1, 255, size=(8, 8, 3), dtype=np.uint8)
state9 = np.round(np.concatenate([rng.uniform(-0.3, 0.3, 3), [1, 0, 0, 0, 1, 0], [0.4]]), 4)          # xyz, rot6d (identity), gripper
# raw language action of a sample: the summed delta of the next steps with the last gripper value (ONE 7-vector: the end-effector-frame
# transform asserts ndim == 1, frame_transforms.py:28)
chunk = np.array([0.04, 0.013, -0.02, 0.0, 0.05, 0.1, 1.0])
idle = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.4])
cases = [
    ("inference request: CHW float image, byte prompt, frame description, no wrist camera", dict(action_dim=32),
     {"observation": {"base_0_rgb": rng.random((3, 8, 8)).astype(np.float32), "state": state9[:8].copy()}, "prompt": b"put_the cup on the plate.",
      "frame_description": b"end-effector frame"}),
    ("training sample: raw language-action chunk, wrist image, dataset name", dict(action_dim=32),
     {"observation": {"base_0_rgb": img(), "left_wrist_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": b"droid",
      "actions": rng.normal(size=(4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
    ("training sample: idle chunk -> sample_mask False", dict(action_dim=32),
     {"observation": {"base_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "bridge_v2_oxe",
      "actions": np.zeros((4, 7)), "language_actions": idle.copy(), "raw_state": state9.copy()}),
    ("training sample: rough scale labels", dict(action_dim=32, use_rough_scale=True),
     {"observation": {"base_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "droid",
      "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
    ("training sample: base frame forced (random_base_prob 1)", dict(action_dim=32, random_base_prob=1.0),
     {"observation": {"base_0_rgb": img(), "left_wrist_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "droid",
      "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy(), "has_wrist_image": True}),
    ("training sample: random_base_prob 0.5, wrist image present (python random, seeded)", dict(action_dim=32, random_base_prob=0.5),
     {"observation": {"base_0_rgb": img(), "left_wrist_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "droid",
      "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy(), "has_wrist_image": True, "rotation_applied": True}),
    ("training sample: language-action training off", dict(action_dim=32, enable_langact_training=False),
     {"observation": {"base_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "droid",
      "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
    ("training sample: wrist dropout 1, random un-masking 1", dict(action_dim=32, wrist_image_dropout_prob=1.0, random_mask_prob=1.0),
     {"observation": {"base_0_rgb": img(), "left_wrist_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "droid",
      "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
    ("VQA sample: caption as label", dict(action_dim=32),
     {"observation": {"base_0_rgb": img(), "state": np.zeros(8)}, "prompt": "what is in the image", "caption": "a red block", "dataset_name": "coco_captions",
      "is_vqa_sample": True, "vqa_dataset_id": 3, "actions": np.zeros((4, 7))}),
    ("VLA-0 strategy: label text = the (normalised) action chunk as integers", dict(action_dim=7, transform_strategy="vla0", language_action_format="vla0_chunked"),
     {"observation": {"base_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block", "dataset_name": "libero_10_no_noops",
      "actions": np.clip(rng.normal(size=(10, 7)) * 0.6, -1, 1), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
    ("VLA-0 strategy: inference request (no actions)", dict(action_dim=7, transform_strategy="vla0", language_action_format="vla0_chunked"),
     {"observation": {"base_0_rgb": img(), "state": state9.copy()}, "prompt": "pick up the block"}),
    ("prediction sample: default prompt", dict(action_dim=32),
     {"observation": {"base_0_rgb": img(), "left_wrist_0_rgb": img(), "state": state9.copy()}, "prompt": "ignored", "dataset_name": "droid",
      "is_prediction_sample": True, "time_horizon_seconds": 1.5, "actions": np.zeros((4, 7)), "language_actions": chunk.copy(), "raw_state": state9.copy()}),
]


def dump(v):
    if isinstance(v, dict):
        return {k: dump(x) for k, x in v.items()}
    if isinstance(v, np.ndarray):
        return {"__nd__": v.tolist(), "dtype": str(v.dtype)}
    if isinstance(v, (np.bool_, np.integer, np.floating)):
        return v.item()
    if isinstance(v, bytes):
        return {"__bytes__": v.decode()}
    return v


out = []
for title, cfg, data in cases:
    np.random.seed(5); random.seed(5)
    try:
        res = ref_in.CoTInputs(**cfg)(dict(data, observation=dict(data["observation"])))
        out.append({"title": title, "config": cfg, "data": dump(data), "out": dump(res)})
    except Exception as e:      # (a case the reference rejects stays in the fixture as such)
        out.append({"title": title, "config": cfg, "data": dump(data), "error": type(e).__name__ + ": " + str(e)[:200]})
pathlib.Path(__file__).with_name("cot_inputs_v1.json").write_text(json.dumps(out))
for c in out:
    o = c.get("out", {})
    print(c["title"][:60], "|", c.get("error") or (o.get("language_actions"), o.get("frame_description"), o.get("sample_mask"), o.get("prompt")))
