"""Generates tests/golden/tokenize_v1.json: the REFERENCE's `PaligemmaTokenizer.tokenize` (models/tokenizer.py:221-315) and its
`TokenizePromptAndReasoning` transform (transforms.py:27-110) run on a tiny SentencePiece model (PaliGemma's special ids; the real
tokenizer model cannot be fetched offline) — token ids, the attention / language-action / number / direction / loss masks, the
left-padded dataset-name tokens, truncation, the VQA and prediction formats, seeded reasoning dropout and state dropout.
`lap.models.tokenizer` is imported UNMODIFIED; what it imports from the absent openpi package (a base class it only inherits
`__init__`-less behaviour from) and `lap.shared.download` are registered as empty stand-in modules first, and the tokenizer object is
built without its downloading constructor (`__new__` + the same three attribute assignments the constructor makes).  The transform
class is compiled from its own ClassDef with `DataTransformFn = object`.  The serialized SentencePiece model travels inside the
fixture so that the test tokenizes with the very same vocabulary.  Run in the build container only."""
import ast
import base64
import zlib
import dataclasses
import json
import pathlib
import random
import sys
import types

import numpy as np
import sentencepiece

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, str(ROOT))
for name in ("openpi", "openpi.models", "openpi.models.tokenizer", "lap.shared", "lap.shared.download"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["openpi.models.tokenizer"].PaligemmaTokenizer = type("PaligemmaTokenizer", (), {})
sys.modules["openpi.models"].tokenizer = sys.modules["openpi.models.tokenizer"]
import lap  # noqa: E402,F401
sys.modules["lap.shared"].__path__ = []
sys.modules["lap.shared"].download = sys.modules["lap.shared.download"]
from lap.models import tokenizer as ref_tok  # noqa: E402
from tests.common import tiny_sentencepiece_proto  # noqa: E402

tree = ast.parse(pathlib.Path("/root/reference/src/lap/transforms.py").read_text())
cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TokenizePromptAndReasoning")
cls.decorator_list = []
ns = {"np": np, "dataclasses": dataclasses, "DataTransformFn": object, "DataDict": dict, "PaligemmaTokenizer": ref_tok.PaligemmaTokenizer}
exec(compile(ast.Module([cls], []), "<ref>", "exec"), ns)
Transform = dataclasses.dataclass(frozen=True)(ns["TokenizePromptAndReasoning"])

proto = tiny_sentencepiece_proto()


def make(max_len, prompt_format="lap", prediction_format="default", p=0.0):
    tk = ref_tok.PaligemmaTokenizer.__new__(ref_tok.PaligemmaTokenizer)
    tk._tokenizer = sentencepiece.SentencePieceProcessor(model_proto=proto)
    tk._max_len = max_len
    tk._init_formats(prompt_format, prediction_format, p)
    return tk


jl = lambda a: None if a is None else np.asarray(a).astype(int).tolist()
state = [0.1, -0.5, 0.9, 0.0, 0.25, -0.75, 1.0, -1.0]
cases = []
grid = [
    dict(max_len=48, prompt="pick up the block", reasoning=None, state=state),
    dict(max_len=160, prompt="pick up the block", reasoning="move right 3 cm\nand open_gripper", state=state),
    dict(max_len=160, prompt="Put_the cup on the plate.", reasoning="move left 5 cm and move up 2 cm and rotate clockwise 10 degrees", state=None),
    dict(max_len=40, prompt="pick up the block", reasoning="move right 3 cm and move forward 1 cm and close gripper", state=state),       # truncated inside the reasoning
    dict(max_len=160, prompt="what is in the image", reasoning="a block on the table 3 cm left", state=None, is_vqa_sample=True),
    dict(max_len=160, prompt="pick up the block", reasoning="move back 4 cm", state=state, is_prediction_sample=True, time_horizon_seconds=2.0),
    dict(max_len=160, prompt="pick up the block", reasoning="move up 2 cm", state=state, frame_description="end-effector frame"),
    dict(max_len=160, prompt="pick up the block", reasoning="move right 3 cm and move up 2 cm and open gripper", state=state, p=0.5, seed=7),
    dict(max_len=160, prompt="pick up the block", reasoning="move up 2 cm", state=state, state_dropout=1.0, seed=3),
]
for g in grid:
    g = dict(g)
    tk = make(g.pop("max_len"), p=g.pop("p", 0.0))
    seed = g.pop("seed", None)
    if seed is not None:
        np.random.seed(seed); random.seed(seed)
    kw = {k: g[k] for k in ("is_vqa_sample", "is_prediction_sample", "time_horizon_seconds", "frame_description", "state_dropout") if k in g}
    st = None if g["state"] is None else np.asarray(g["state"])
    toks, attn, reason, num, direc, loss = tk.tokenize(g["prompt"], g["reasoning"], st, **kw)
    cases.append({"args": {**g, "max_len": tk._max_len, "p": tk.reasoning_mask_prob, "seed": seed}, "tokens": jl(toks), "attn": jl(attn), "reason": jl(reason),
                  "number": jl(num), "direction": jl(direc), "loss": jl(loss)})

# the transform: robot training sample, inference request (no language action / dataset name), verbose masks
tcases = []
for verbose, sample in [
    (True, {"prompt": "pick up the block", "language_actions": "move right 3 cm and open gripper", "dataset_name": "droid", "state": np.asarray(state),
            "is_vqa_sample": False, "is_prediction_sample": False, "frame_description": "robot base frame", "actions": np.zeros((2, 7))}),
    (False, {"prompt": np.asarray("pick up the block"), "state": np.asarray(state), "is_vqa_sample": False, "is_prediction_sample": False}),
    (True, {"prompt": "what is in the image", "language_actions": "a block", "dataset_name": "coco_captions", "state": np.asarray(state),
            "is_vqa_sample": True, "is_prediction_sample": False, "time_horizon_seconds": 1.0}),
]:
    tf = Transform(make(96), discrete_state_input=True, dataset_name_pad_len=20, verbose_mode=verbose)
    out = tf(dict(sample))
    rec = {"verbose": verbose, "sample": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in sample.items()}, "out": {}}
    for k, v in out.items():
        if k in ("state", "actions"):
            continue
        rec["out"][k] = jl(v) if isinstance(v, np.ndarray) else v
    rec["out_keys"] = sorted(out)
    tcases.append(rec)

pathlib.Path(__file__).with_name("tokenize_v1.json").write_text(json.dumps(
    {"sentencepiece_model_zlib_b64": base64.b64encode(zlib.compress(proto, 9)).decode(), "tokenize": cases, "transform": tcases}))
print("wrote tokenize_v1.json:", len(cases), "tokenize cases,", len(tcases), "transform cases;", [sum(c["reason"] or []) for c in cases])
