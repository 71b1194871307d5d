"""Generates tests/golden/dataset_configs_v1.json: `OXE_DATASET_METADATA` of the REFERENCE's datasets/utils/configs.py (a dict of plain
literals) read with ast.literal_eval — nothing is imported.  Run in the build container only."""
import ast
import json
import pathlib

tree = ast.parse(pathlib.Path("/root/reference/src/lap/datasets/utils/configs.py").read_text())
meta = ast.literal_eval(next(n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "OXE_DATASET_METADATA").value)
out = {"control_frequency": {k: v["control_frequency"] for k, v in meta.items() if "control_frequency" in v}}
pathlib.Path(__file__).with_name("dataset_configs_v1.json").write_text(json.dumps(out, indent=1, sort_keys=True))
print("wrote dataset_configs_v1.json:", len(out["control_frequency"]), "datasets")
