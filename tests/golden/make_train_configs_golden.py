"""Generates tests/golden/train_configs_v1.json: the keyword arguments of every `TrainConfig(...)` in the REFERENCE's registry
(`training/config.py` `_CONFIGS`) and the field defaults of `LAPConfig` (`models/lap_config.py`), read from the parsed source with
`ast.literal_eval` (nothing is imported or executed: the modules need openpi / JAX).  Nested constructor calls become dicts with the
callee's name under "__call__".  Run in the build container only."""
import ast
import json
import pathlib

REF = pathlib.Path("/root/reference/src/lap")


def conv(node):
    if isinstance(node, ast.Call):
        return {"__call__": ast.unparse(node.func), **{k.arg: conv(k.value) for k in node.keywords}}
    try:
        return ast.literal_eval(node)
    except Exception:
        return {"__expr__": ast.unparse(node)}


tree = ast.parse((REF / "training/config.py").read_text())
cfgs = next(n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "_CONFIGS").value
registry = {}
for e in cfgs.elts:
    if isinstance(e, ast.Call):
        d = conv(e)
        registry[d["name"]] = d
cls = next(n for n in ast.parse((REF / "models/lap_config.py").read_text()).body if isinstance(n, ast.ClassDef) and n.name == "LAPConfig")
defaults = {n.target.id: conv(n.value) for n in cls.body if isinstance(n, ast.AnnAssign) and n.value is not None and isinstance(n.target, ast.Name)}
def class_defaults(t, name):
    c = next(n for n in t.body if isinstance(n, ast.ClassDef) and n.name == name)
    return {n.target.id: conv(n.value) for n in c.body if isinstance(n, ast.AnnAssign) and n.value is not None and isinstance(n.target, ast.Name)}


data_defaults = {**class_defaults(tree, "DataConfig"), **class_defaults(tree, "RLDSDataConfig")}
train_defaults = class_defaults(tree, "TrainConfig")
pathlib.Path(__file__).with_name("train_configs_v1.json").write_text(json.dumps(
    {"registry": registry, "lap_config_defaults": defaults, "data_config_defaults": data_defaults, "train_config_defaults": train_defaults}, indent=1))
print("wrote train_configs_v1.json:", sorted(registry))
