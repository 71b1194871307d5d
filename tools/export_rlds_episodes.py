"""Offline exporter: RLDS / TFDS shards -> the `.npz` episode store `lap_amd/data.py` reads.

Runs where `tensorflow_datasets` exists (it does not in the build image: the record -> sample mapping is in
`lap_amd/rlds_export.py` and unit-tested on hand-built trajectories; only the shard iteration below is untested here):

    python tools/export_rlds_episodes.py --dataset libero_10_no_noops --data-dir /data/rlds --out /data/episodes/libero_10 [--max-episodes N]

The reference iterates the same builder (`tfds.builder(name, data_dir=...)`, `datasets/base_dataset.py`) through dlimp; one
output file per RLDS episode, standardised per dataset (`STANDARDIZE`), instruction-less / empty episodes dropped.
"""
from __future__ import annotations

import argparse
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lap_amd import rlds_export as R


def stack_steps(steps) -> dict:
    """A list of per-step nested dicts -> one nested dict of arrays stacked over time."""
    first = steps[0]
    if isinstance(first, dict):
        return {k: stack_steps([s[k] for s in steps]) for k in first}
    return np.stack([np.asarray(s) for s in steps], 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", required=True, choices=sorted(R.STANDARDIZE))
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--split", default="train")
    ap.add_argument("--max-episodes", type=int, default=None)
    ap.add_argument("--droid-keep-ranges", default=None, help="DROID only: the keep-ranges JSON of the reference's idle-frame filter (droid_mixins.py:113-143)")
    args = ap.parse_args()
    keep_ranges = None
    if args.droid_keep_ranges:
        import json
        keep_ranges = json.loads(pathlib.Path(args.droid_keep_ranges).read_text())
    try:
        import tensorflow_datasets as tfds
    except ImportError as e:   # pragma: no cover - the build image has no TensorFlow
        raise SystemExit(f"tensorflow_datasets is required to read RLDS shards ({e}); run this tool where the reference's data stack is installed")
    out = pathlib.Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    builder = tfds.builder(args.dataset, data_dir=args.data_dir)
    ds = tfds.as_numpy(builder.as_dataset(split=args.split))
    kept = dropped = 0
    rng = np.random.default_rng(0)

    def tf_hash_bucket(value, n):   # transform_helpers.py:110-114: the reference's own hash of the decimal string of sum(state[0])
        import tensorflow as tf
        return int(tf.strings.to_hash_bucket_fast(tf.strings.as_string(tf.constant(value, tf.float32)), n))

    for i, episode in enumerate(ds):
        if args.max_episodes is not None and kept >= args.max_episodes:
            break
        traj = stack_steps(list(episode["steps"]))
        if "episode_metadata" in episode:          # (DROID: success filter, keep ranges)
            traj["traj_metadata"] = {"episode_metadata": {k: np.asarray(v) for k, v in episode["episode_metadata"].items()}}
        mask = R.droid_keep_mask(keep_ranges, traj) if (keep_ranges is not None and args.dataset == "droid") else None
        ep = R.episode_from_rlds(args.dataset, traj, hash_bucket=tf_hash_bucket, rng=rng, keep_mask=mask)
        if ep is None:
            dropped += 1
            continue
        R.write_episode(out / f"episode_{i:07d}.npz", ep)
        kept += 1
    print(f"{args.dataset}: wrote {kept} episodes to {out} ({dropped} dropped: empty instruction / zero length)")


if __name__ == "__main__":
    main()
