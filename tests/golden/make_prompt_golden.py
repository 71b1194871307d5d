"""Generates tests/golden/prompt_format_v1.json by running the REFERENCE's prompt modules (pure Python / numpy, importable
in the build container) on the case tables of lap_amd/prompt.py.  Run here only:  python tests/golden/make_prompt_golden.py
(/root/reference does not exist on the GPU box; the committed JSON is what the tests read)."""
import json
import pathlib
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))

from lap.models.prompt_utils import checkers as ref_checkers          # noqa: E402
from lap.models.prompt_utils import prompt as ref_prompt              # noqa: E402
from lap.models.prompt_utils import state as ref_state                # noqa: E402

from lap_amd import prompt as mine                                    # noqa: E402  (case tables only)

REG = {"prompt": ref_prompt.PROMPT_FORMAT_REGISTRY, "prediction": ref_prompt.PREDICTION_PROMPT_FORMAT_REGISTRY,
       "vqa": {"default_vqa": ref_prompt.DEFAULT_VQA_PROMPT_FORMAT}}
TEMPLATES = {"default": ref_state.DEFAULT_STATE_TEMPLATE, "named_params": ref_state.NAMED_PARAMS_STATE_TEMPLATE,
             "verbose": ref_state.VERBOSE_STATE_TEMPLATE, "grouped": ref_state.GROUPED_STATE_TEMPLATE}

out = {"format": [], "state": [], "checkers": {}}
for c in mine.format_cases():
    fmt = REG[c["registry"]][c["format"]]
    st = None if c["state_values"] is None else np.asarray(c["state_values"], dtype=np.float64)
    text = fmt.format_prompt(c["prompt"], st, c["state_type"], time_horizon_seconds=c["horizon"], frame_description=c["frame"])
    out["format"].append(dict(c, expected=text))
for c in mine.state_text_cases():
    cfg = ref_state.StateDiscretizationConfig(bins=256, min_dim=c["min_dim"], template=TEMPLATES[c["template"]])
    out["state"].append(dict(c, expected=cfg.discretize_state(np.asarray(c["state_values"], dtype=np.float64))))
for name in mine.CHECKERS:
    out["checkers"][name] = [bool(getattr(ref_checkers, name)(p)) for p in mine.CHECKER_PIECES]
out["pieces"] = mine.CHECKER_PIECES
path = pathlib.Path(__file__).with_name("prompt_format_v1.json")
path.write_text(json.dumps(out, indent=1, ensure_ascii=False))
print(len(out["format"]), "prompt cases,", len(out["state"]), "state cases ->", path)
