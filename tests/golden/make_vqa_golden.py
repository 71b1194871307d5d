"""Generates tests/golden/vqa_v1.json from the REFERENCE's VQA readers (lap/datasets/vqa/**): the prompt tables those modules
define and the outputs of their two pure-Python geometry functions.  The modules import TensorFlow at the top and cannot be
imported in the build container, so nothing is imported: the sources are parsed with `ast`; constant tables are evaluated from
their literal nodes (for `X = tf.constant([...], dtype=tf.string)` the list literal), and the two functions that use nothing but
floats and strings (`bbox_to_loc_tokens`, `compute_direction_from_bbox` with add_move_prefix=False) are compiled from their own
FunctionDef nodes and run on a case grid.  Run in the build container only:  python tests/golden/make_vqa_golden.py"""
import ast
import json
import pathlib

REF = pathlib.Path("/root/reference/src/lap/datasets/vqa")


def module(path):
    return ast.parse((REF / path).read_text())


def tf_constant_list(tree, name):
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets):
            v = node.value
            if isinstance(v, ast.Call):            # tf.constant([...], dtype=tf.string)
                v = v.args[0]
            return ast.literal_eval(v)
    raise KeyError(name)


def exec_assignments(tree, names):
    """Evaluate module-level assignments (plain lists / comprehensions over earlier ones) in a clean namespace, in order."""
    ns = {}
    for node in tree.body:
        targets = [t.id for t in getattr(node, "targets", []) if isinstance(t, ast.Name)]
        if isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            targets = [node.target.id]
        if targets and (set(targets) & names or targets[0].startswith("_ROBOT_")):
            exec(compile(ast.Module([node], []), "<ref>", "exec"), ns)
    return {k: ns[k] for k in names}


def function(tree, name):
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            ns = {}
            exec(compile(ast.Module([node], []), "<ref>", "exec"), ns)
            return ns[name]
    raise KeyError(name)


out = {}
out["COCO_CAPTION_PROMPTS"] = tf_constant_list(module("coco_caption_dataset.py"), "COCO_CAPTION_PROMPTS")
out["PIXMO_CAP_PROMPTS"] = tf_constant_list(module("pixmo_cap_dataset.py"), "PIXMO_CAP_PROMPTS")
pp = module("pixmo_point_dataset.py")
out["PIXMO_POINT_PROMPT_PARTS"] = [list(p) for p in tf_constant_list(pp, "PIXMO_POINT_PROMPT_PARTS")]
out["MAX_POINTS"] = tf_constant_list(pp, "MAX_POINTS")
names = {"GENERAL_BBOX_PROMPT_PARTS", "ROBOT_BBOX_PROMPT_PARTS", "ROBOT_BBOX_PROMPT_PARTS_OXE", "ROBOT_BBOX_PROMPT_PARTS_EE", "DIRECTION_PROMPT_PARTS",
         "ROBOT_DIRECTION_PROMPT_PARTS_OXE", "ROBOT_DIRECTION_PROMPT_PARTS_EE"}
for k, v in exec_assignments(module("bbox/prompts.py"), names).items():
    out[k] = [list(p) for p in v]

# the size constants the VQA sets enter the mixture weights with: `return <int>` of every reader's get_num_transitions, by registered name
sizes = {}
for f in sorted(REF.glob("*_dataset.py")):
    for c in [n for n in ast.walk(ast.parse(f.read_text())) if isinstance(n, ast.ClassDef)]:
        name = next((ast.literal_eval(k.value) for d in c.decorator_list if isinstance(d, ast.Call) for k in d.keywords if k.arg == "name"), None)
        for fn in c.body:
            if isinstance(fn, ast.FunctionDef) and fn.name == "get_num_transitions" and name:
                ret = [n for n in ast.walk(fn) if isinstance(n, ast.Return)]
                if ret:
                    sizes[name] = ast.literal_eval(ret[0].value)
out["NUM_TRANSITIONS"] = sizes

loc = function(module("bbox/coord_utils.py"), "bbox_to_loc_tokens")
direction = function(module("bbox/direction.py"), "compute_direction_from_bbox")
grid = [0.0, 0.03, 0.12, 0.25, 0.333, 0.4995, 0.5, 0.5005, 0.62, 0.75, 0.9, 0.999, 1.0]
boxes = [(a, b, c, d) for a in grid for b in grid[::3] for c in grid[1::4] for d in grid[2::5] if c >= a and d >= b]
out["loc_cases"] = [{"box": list(b), "expected": loc(*b)} for b in boxes]
out["loc_cases"] += [{"box": list(b), "bins": 256, "expected": loc(*b, num_bins=256)} for b in boxes[::7]]
out["direction_cases"] = [{"box": list(b), "slope": s, "expected": direction(*b, slope=s, add_move_prefix=False)} for b in boxes for s in (2.0, 1.0, 3.5)]
path = pathlib.Path(__file__).with_name("vqa_v1.json")
path.write_text(json.dumps(out, indent=0, ensure_ascii=False))
print({k: len(v) if hasattr(v, "__len__") else v for k, v in out.items()})
