"""Train loop entry point (scripts/train.py:422-640 surface) on the debug model: interval logging, checkpointing,
resume from the latest checkpoint with the data position, and equivalence of "6 steps + resume + 2 steps" with an
uninterrupted 8-step run (up to the f32 atomics order of the small-unit gradients)."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_resume_matches_uninterrupted_run(hip, tmp_path):
    from lap_amd import checkpoints as ck
    from lap_amd.config import get_config
    from lap_amd.train import main

    base = dataclasses.replace(get_config("debug"), checkpoint_base_dir=str(tmp_path), batch_size=4, log_interval=2,
                               save_interval=3, keep_period=None, seed=3)
    lines = []
    a = main(dataclasses.replace(base, exp_name="full", num_train_steps=8), log=lines.append)
    assert a.step == 8 and any(l.startswith("step 8:") for l in lines) and "samples/s" in lines[-1]
    assert ck.CheckpointManager(base.checkpoint_base_dir and (tmp_path / base.name / "full")).all_steps() == (8,)
    b = main(dataclasses.replace(base, exp_name="split", num_train_steps=6), log=lambda s: None)
    assert b.step == 6
    lines2 = []
    c = main(dataclasses.replace(base, exp_name="split", num_train_steps=8), log=lines2.append)   # resume=True by default
    assert c.step == 8 and lines2[0].startswith("resumed from step 6")
    # yardstick: two uninterrupted runs differ by the f32 atomics order of some gradient reductions, amplified by bf16
    # rounding over the steps; a resume bug (moments, step count, data position) is orders of magnitude above that
    a2 = main(dataclasses.replace(base, exp_name="full2", num_train_steps=8), log=lambda s: None)

    def worst(p, q):
        w = 0.0
        for u in p.units:
            for buf in ("master", "m", "v"):
                x, y = getattr(p, buf)[u.name], getattr(q, buf)[u.name]
                w = max(w, float((x - y).norm() / (x.norm() + 1e-12)))
        return w

    noise = worst(a.model.ps, a2.model.ps)
    assert worst(a.model.ps, c.model.ps) <= max(5 * noise, 1e-3), (worst(a.model.ps, c.model.ps), noise)
    with pytest.raises(FileExistsError):
        main(dataclasses.replace(base, exp_name="split", num_train_steps=8, resume=False), log=lambda s: None)
