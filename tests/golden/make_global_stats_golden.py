"""Generates tests/golden/global_stats_v1.json from the REFERENCE's `GlobalStatisticsBuilder` (datasets/utils/statistics.py:17-236,
numpy only).  The file is loaded by path, unmodified (importing the `lap.datasets` package would pull TensorFlow); the two names
it imports inside its methods are provided as stand-in modules: `lap.datasets.utils.helpers.state_encoding_to_type` (never called:
the generator passes its own mapping function) and `lap.shared.normalize_adapter.ExtendedNormStats` (a plain record of the eight
fields the builder fills).  Case: five datasets of different action / state widths and sizes — two end-effector-pose sets, one
joint-position set, one without state, one VQA set that must contribute nothing.  Run in the build container only."""
import dataclasses
import importlib.util
import json
import pathlib
import sys
import types

import numpy as np


@dataclasses.dataclass
class ExtendedNormStats:
    mean: np.ndarray
    std: np.ndarray
    q01: np.ndarray | None = None
    q99: np.ndarray | None = None
    min: np.ndarray | None = None
    max: np.ndarray | None = None
    num_transitions: int = 0
    num_trajectories: int = 0


for name in ("lap", "lap.datasets", "lap.datasets.utils", "lap.datasets.utils.helpers", "lap.shared", "lap.shared.normalize_adapter"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["lap.datasets.utils.helpers"].state_encoding_to_type = lambda e: (_ for _ in ()).throw(AssertionError("not used"))
sys.modules["lap.shared.normalize_adapter"].ExtendedNormStats = ExtendedNormStats
spec = importlib.util.spec_from_file_location("ref_statistics", "/root/reference/src/lap/datasets/utils/statistics.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(4)


def stats(dim, n, traj):
    lo = rng.normal(size=dim).astype(np.float32) - 1
    hi = lo + np.abs(rng.normal(size=dim)).astype(np.float32) + 0.1
    return ExtendedNormStats(mean=rng.normal(size=dim).astype(np.float32), std=(np.abs(rng.normal(size=dim)) + 0.1).astype(np.float32),
                             q01=lo, q99=hi, min=lo - 0.5, max=hi + 0.5, num_transitions=n, num_trajectories=traj)


sets = {"droid": (7, 8, 120000, 900, "eef_pose"), "bridge": (7, 10, 45000, 300, "eef_pose"), "aloha": (14, 14, 30000, 50, "joint_pos"),
        "stateless": (7, 6, 5000, 20, "none"), "coco_captions": (7, 8, 99999, 99999, "eef_pose")}
allst = {k: {"actions": stats(a, n, t), "state": stats(s, n, t)} for k, (a, s, n, t, _) in sets.items()}
enc = {k: v[4] for k, v in sets.items()}
g = ref.GlobalStatisticsBuilder(action_dim=32, state_dim=10).compute_global_stats(allst, enc, {"coco_captions"}, state_encoding_to_type_fn=lambda e: e)
fields = ("mean", "std", "q01", "q99", "min", "max")
dump = lambda s: {**{f: np.asarray(getattr(s, f), dtype=np.float64).tolist() for f in fields}, "num_transitions": int(s.num_transitions), "num_trajectories": int(s.num_trajectories)}
out = {"action_dim": 32, "state_dim": 10, "vqa": ["coco_captions"], "state_types": enc,
       "per_dataset": {k: {g2: dump(v[g2]) for g2 in ("actions", "state")} for k, v in allst.items()},
       "global": {k: dump(v) for k, v in g.items()}}
pathlib.Path(__file__).with_name("global_stats_v1.json").write_text(json.dumps(out))
print("wrote global_stats_v1.json:", sorted(g), {k: v.num_transitions for k, v in g.items()})
