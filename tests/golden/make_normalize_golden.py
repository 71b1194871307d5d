"""Generates tests/golden/normalize_v1.json from the REFERENCE's normalisation transforms (`src/lap/transforms.py`): `Normalize`,
`Unnormalize` (policy side: policy_config_adapter.py:79,146) and `NormalizeActionAndProprio` (training data side: the mixer maps
it over every robot dataset, dataset_mixer.py:352-359).  The module imports openpi (absent here), so nothing is imported: the three
ClassDef nodes and `pad_to_dim` are compiled from the parsed source with stand-ins for the names they only use as annotations or
base classes (`DataTransformFn`, `at`, `NormStats`, `DataDict`), `tf = None` (the numpy branches run), the real `NormalizationType`
enum compiled from `datasets/utils/helpers.py`, and openpi's two helpers restated in three lines each (`apply_tree`: apply to every
flat key of the data that has statistics; `_assert_quantile_stats`).  Inputs are a seeded case grid; outputs are stored as lists.
Run in the build container only:  python tests/golden/make_normalize_golden.py"""
import ast
import dataclasses
import enum
import json
import pathlib
import types

import numpy as np

SRC = pathlib.Path("/root/reference/src/lap")
tree = ast.parse((SRC / "transforms.py").read_text())
helpers = ast.parse((SRC / "datasets/utils/helpers.py").read_text())


def node(t, name, kind):
    return next(n for n in t.body if isinstance(n, kind) and n.name == name)


@dataclasses.dataclass
class Stats:      # openpi.shared.normalize.NormStats: six optional arrays
    mean: np.ndarray | None = None
    std: np.ndarray | None = None
    q01: np.ndarray | None = None
    q99: np.ndarray | None = None
    min: np.ndarray | None = None
    max: np.ndarray | None = None


def apply_tree(data, stats, fn, strict=False):     # openpi.transforms.apply_tree on a one-level dict
    if strict and (missing := [k for k in stats if k not in data]):
        raise ValueError(missing)
    return {k: (fn(v, stats[k]) if k in stats else v) for k, v in data.items()}


def assert_q(stats):
    for k, s in stats.items():
        if s.q01 is None or s.q99 is None:
            raise ValueError(k)


ns = {"np": np, "dataclasses": dataclasses, "Enum": enum.Enum, "tf": None, "DataTransformFn": object, "DataDict": dict, "NormStats": Stats,
      "at": types.SimpleNamespace(PyTree=dict), "apply_tree": apply_tree, "_assert_quantile_stats": assert_q}
exec(compile(ast.Module([node(helpers, "NormalizationType", ast.ClassDef)], []), "<ref>", "exec"), ns)
exec(compile(ast.Module([node(tree, "pad_to_dim", ast.FunctionDef)], []), "<ref>", "exec"), ns)
for cls in ("Normalize", "Unnormalize", "NormalizeActionAndProprio"):
    c = node(tree, cls, ast.ClassDef)
    c.decorator_list = []     # (the frozen-dataclass decorator sits on the line above the ClassDef in the file; re-applied below)
    exec(compile(ast.Module([c], []), "<ref>", "exec"), ns)
    ns[cls] = dataclasses.dataclass(frozen=True)(ns[cls])

rng = np.random.default_rng(20260929)
D = 7
lo = rng.normal(size=D) - 1.0
hi = lo + np.abs(rng.normal(size=D)) + 0.3
hi[3] = lo[3]                      # a constant dimension: zero range
raw = {"mean": rng.normal(size=D), "std": np.abs(rng.normal(size=D)) + 0.05, "q01": lo, "q99": hi, "min": lo - 0.4, "max": hi + 0.4}
raw["max"][3] = raw["min"][3]
raw8 = {k: np.concatenate([v, [v[0] + 0.25]]) for k, v in raw.items()}      # state statistics: 8 wide
x_act = np.concatenate([rng.normal(size=(5, D)) * 2.0, (hi + 3.0)[None], (lo - 3.0)[None]])      # incl. rows far outside the quantiles
x_state = rng.normal(size=8) * 2.0
wide = rng.normal(size=(4, 32))                                                               # model-side actions: 32 wide
jl = lambda a: np.asarray(a, dtype=np.float64).tolist()
out = {"stats": {"actions": {k: jl(v) for k, v in raw.items()}, "state": {k: jl(v) for k, v in raw8.items()}},
       "x_actions": jl(x_act), "x_state": jl(x_state), "wide_actions": jl(wide), "cases": []}
stats = {"actions": Stats(**raw), "state": Stats(**raw8)}
for t in ("normal", "bounds", "bounds_q99"):
    n = ns["Normalize"](stats, t)({"actions": x_act.copy(), "state": x_state.copy(), "other": np.ones(2)})
    u = ns["Unnormalize"](stats, t)({"actions": wide.copy(), "state": x_state.copy()})
    case = {"type": t, "normalize": {k: jl(v) for k, v in n.items()}, "unnormalize": {k: jl(v) for k, v in u.items()}}
    # the training-side transform: dict statistics (as norm_stats.json holds them), "actions" / "state" groups; a trajectory without state;
    # the singular group name "action"
    dstats = {"actions": {k: jl(v) for k, v in raw.items()}, "state": {k: jl(v) for k, v in raw8.items()}}
    tr = ns["NormalizeActionAndProprio"](dstats, t, action_key="actions", state_key="state")({"actions": x_act.astype(np.float64), "observation": {"state": x_state.copy()}})
    case["traj"] = {"actions": jl(tr["actions"]), "state": jl(tr["observation"]["state"]), "dtype": str(tr["actions"].dtype)}
    tr2 = ns["NormalizeActionAndProprio"]({"action": dstats["actions"]}, t, action_key="actions", state_key="state")({"actions": x_act.copy(), "observation": {}})
    case["traj_no_state"] = {"actions": jl(tr2["actions"])}
    out["cases"].append(case)
pathlib.Path(__file__).with_name("normalize_v1.json").write_text(json.dumps(out))
print("wrote normalize_v1.json", {c["type"]: (np.abs(np.asarray(c["traj"]["actions"])).max(), np.abs(np.asarray(c["normalize"]["actions"])).max()) for c in out["cases"]})
