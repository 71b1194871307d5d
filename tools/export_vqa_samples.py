"""Offline exporter: TFDS records of the reference's vision-language datasets -> the `.npz` samples `lap_amd.data.VqaDataset` reads.

Runs where `tensorflow_datasets` exists (not in the build image: the record -> sample mapping is `lap_amd/vqa_export.py`, pinned by
reference-generated fixtures; only the shard iteration and the two TensorFlow primitives passed in below are untested here):

    python tools/export_vqa_samples.py --dataset coco_captions --data-dir /data/tfds --out /data/vqa/coco [--max-samples N] [--seed 0]

The reference builds the same TFDS builders (`tfds.builder("coco_captions" | "vqa" | "pixmo_cap" | "pixmo_point" | "lvis:1.0.0" |
"paco_lvis:1.0.0" | "paco_ego4d:1.0.0", data_dir=...)`, datasets/vqa/*.py), splits train / validation by the hash of the
trajectory id and draws prompt / caption with stateless TensorFlow RNG: the same hash and the same draws are used here.
"""
from __future__ import annotations

import argparse
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lap_amd import vqa_export as V

BUILDERS = {"coco_captions": "coco_captions", "vqa": "vqa", "pixmo_cap": "pixmo_cap", "pixmo_point": "pixmo_point", "lvis": "lvis:1.0.0",
            "paco_lvis": "paco_lvis:1.0.0", "paco_ego4d": "paco_ego4d:1.0.0"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", required=True, choices=sorted(BUILDERS))
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--split", default="train", choices=["train", "val"])
    ap.add_argument("--val-fraction", type=float, default=0.02)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--direction-prob", type=float, default=0.0)
    ap.add_argument("--max-samples", type=int, default=None)
    args = ap.parse_args()
    try:
        import tensorflow as tf
        import tensorflow_datasets as tfds
    except ImportError as e:   # pragma: no cover - the build image has no TensorFlow
        raise SystemExit(f"tensorflow_datasets is required to read the TFDS shards ({e}); run this tool where the reference's data stack is installed")

    def hash_bucket(text, n):
        return int(tf.strings.to_hash_bucket_fast(tf.constant(text), n))

    def choose(seed_pair, n):
        return int(tf.random.stateless_uniform([], seed=[int(seed_pair[0]), int(seed_pair[1])], minval=0, maxval=n, dtype=tf.int32))

    def uniform(seed_pair):
        return float(tf.random.stateless_uniform([], seed=[int(seed_pair[0]), int(seed_pair[1])], dtype=tf.float32))

    out = pathlib.Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    ds = tfds.as_numpy(tfds.builder(BUILDERS[args.dataset], data_dir=args.data_dir).as_dataset(split="train"))
    kept = dropped = 0
    for i, rec in enumerate(ds):
        if args.max_samples is not None and kept >= args.max_samples:
            break
        if V.is_validation(args.dataset, rec, args.seed, args.val_fraction, hash_bucket) != (args.split == "val"):
            continue
        s = V.sample_from_record(args.dataset, rec, args.seed, direction_prob=args.direction_prob, hash_bucket=hash_bucket, choose=choose, uniform=uniform)
        if s is None:
            dropped += 1
            continue
        np.savez_compressed(out / f"sample_{i:08d}.npz", **s)
        kept += 1
    print(f"{args.dataset}: wrote {kept} samples to {out} ({dropped} dropped: empty question / answer)")


if __name__ == "__main__":
    main()
