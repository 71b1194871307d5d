"""Generates tests/golden/lap_debug_v1.npz: seeded inputs -> outputs of the CPU oracle (f32 mode) on the debug-size
LAP model.  The reference itself cannot be run here, so these vectors pin the ORACLE (against
regressions) and give the GPU tests fixed expected values; they are data only (inputs, parameters seed, outputs)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lap_oracle as O  # noqa: E402
from tests.common import debug_model_cfg, make_inputs, oracle_cfg  # noqa: E402


def main():
    torch.set_num_threads(4)
    cfg = debug_model_cfg()
    oc = oracle_cfg(cfg)
    P = O.init_params(oc, seed=21)
    obs, actions, noise, time = make_inputs(cfg, B=3, ragged=True, seed=5)
    col = {}
    loss, m = O.compute_loss(P, oc, obs, actions, noise, time, collect=col)
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"}
    sampled = O.sample_actions(P, oc, so, noise, num_steps=10)
    last = oc.vlm.depth - 1
    out = dict(
        param_seed=np.int64(21), input_seed=np.int64(5), batch=np.int64(3),
        loss=loss.detach().numpy(), per_sample_lang=m["per_sample_lang"].detach().numpy(),
        per_sample_action=m["per_sample_action"].detach().numpy(), v_t=m["v_t"].detach().numpy(),
        img_tokens=col["img/out"].detach().numpy(), x0_last=col[f"llm/layer{last:02d}/x0"].detach().numpy(),
        x1_last=col[f"llm/layer{last:02d}/x1"].detach().numpy(), positions=col["positions"].numpy(),
        mask_rowsum=col["mask"].sum(-1).numpy(), sampled_actions=sampled.detach().numpy(),
        # spot parameters so that a changed initialiser is detected
        p_q0=P["PaliGemma/llm/layers/attn/q_einsum/w"][0, 0, :4, :4].numpy(), p_head=P["PaliGemma/img/head/kernel"][:4, :4].numpy(),
    )
    path = os.path.join(ROOT, "tests", "golden", "lap_debug_v1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; loss", float(out["loss"]))


if __name__ == "__main__":
    main()
