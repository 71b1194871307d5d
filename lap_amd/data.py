"""Host-side data path: episodes on disk -> transformed, batched `(CoTObservation, actions)`.

The reference reads RLDS / TFDS shards through TensorFlow (`src/lap/datasets/**`, `datasets/data_loader.py:97-326`), applies
the per-sample transform stack to every element of a TF batch and yields `CoTObservation.from_dict(batch), batch["actions"]`
(`data_loader.py:286-326`).  TensorFlow and the RLDS corpora do not exist here, so the STORAGE format is our own (one `.npz`
per episode, below); everything after a raw sample dict exists is the reference's pipeline: `CoTInputs` (label text, frame,
idle mask) -> `NormalizeActionAndProprio` (the mixer's: clipped q01/q99 bounds) -> `TokenizePromptAndReasoning` -> `PadStatesAndActions` -> stack ->
`CoTObservation`.  The loader keeps the contract the train loop and the checkpoint code rely on: endless iteration for
`split="train"`, one pass for `"val"`, `get_state` / `set_state` resume with the batches the interrupted run would have seen
(the reference skips `batches_seen` batches of the host's shard, data_loader.py:289-306), per-rank shards of every epoch's
permutation, `get_norm_stats_for_checkpoint`.

Episode file (`np.savez`): `base_0_rgb` / `left_wrist_0_rgb` uint8 [T, H, W, 3] (wrist optional), `state` f32 [T, >= 9]
([xyz, rot6d, gripper, ...]; 7 values = [xyz, euler, gripper] also work for base-frame labels), `actions` f32 [T, 7]
per-step deltas [dx, dy, dz, droll, dpitch, dyaw, gripper] in m / rad, `prompt` str, `dataset_name` str.
A sample at step t carries the action chunk `actions[t : t + action_horizon]` (the last row repeated past the episode end,
with zero motion) and, as raw language action, the summed delta of the next `summation_steps` steps with the last gripper
value — what the label text describes (`action_text.py:47-141` sums a chunk the same way).
"""
from __future__ import annotations

import pathlib
from typing import Callable, Sequence

import numpy as np
import torch

from lap_amd import policy_io as pio
from lap_amd.observation import CoTObservation

IMAGE_KEYS = pio.IMAGE_KEYS


# datasets/registry.py:104-163,402-407: datasets whose wrist camera is mounted upside down — the frame is rotated by 180 degrees, and the
# end-effector-frame label text is told so (`rotation_applied`, frame_transforms.py:52-54)
WRIST_ROTATION_PATTERNS = ("droid", "aloha", "mobile_aloha", "furniture_bench_dataset_converted_externally_to_rlds",
                           "berkeley_fanuc_manipulation", "berkeley_autolab_ur5", "fmb")


def needs_wrist_rotation(dataset_name: str) -> bool:
    return any(p in dataset_name for p in WRIST_ROTATION_PATTERNS)


def sum_language_actions(window: np.ndarray) -> np.ndarray:
    """base_dataset.py:722-776 (`sum_actions`, one window of per-step language actions [dx, dy, dz, droll, dpitch, dyaw, tail...]):
    translations add, rotations COMPOSE in order (R = R_1 R_2 ..., extrinsic XYZ angles of the product: not the sum of the angles), the
    tail (gripper ...) is the last row's."""
    from lap_amd.rlds_export import euler_to_rotation_matrix
    w = np.asarray(window, dtype=np.float64)
    if w.shape[-1] < 6:
        w = np.pad(w, [(0, 0), (0, 6 - w.shape[-1])])
    R = np.eye(3)
    for rpy in w[:, 3:6]:
        R = R @ euler_to_rotation_matrix(rpy)
    sy = np.sqrt(max(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0], 1e-12))        # _matrix_to_euler_xyz_extrinsic (base_dataset.py:702-719)
    if sy < 1e-6:
        rpy = [np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), 0.0]
    else:
        rpy = [np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])]
    return np.concatenate([w[:, :3].sum(0), rpy, w[-1, 6:]]).astype(np.float32)


class EpisodeDataset:
    """Random access over (episode, step) pairs of a list of episode dicts or of `*.npz` files in a directory; a sample is what the
    reference's trajectory-level transforms leave per frame (`datasets/base_dataset.py:428-590,603-697`):

    * state: an episode marked `state_encoding = "pos_euler"` (the exporter's) stores [xyz, euler, gripper]; it becomes
      [xyz, rot6d, gripper] (`state_euler_to_rot6d`, :437-456) — what the model sees and what the end-effector-frame labels rotate with;
    * action chunk (`chunk_mode`): "relative" — over a last-value-padded window of `target_actions` (absolute pose + gripper command),
      row k = [pose[t+1+k] - pose[t] (translation), euler_diff(pose[t+1+k], pose[t]), gripper[t+k]] (:387-427: every end-effector dataset
      incl. DROID); "window_zero" — rows t .. t+H-1 of `target_actions`, zeros past the end (LIBERO, oxe_datasets.py:259-269); absent —
      "steps": rows of `actions` (per-step deltas), motion zeroed and gripper held past the end (hand-built stores);
    * raw language action: `sum_language_actions` over the next round(horizon_seconds x control_frequency) steps (one of
      `horizon_seconds` drawn per frame), cut at the episode's end, and `time_horizon_seconds` = steps used / frequency (:493-531).
      Without a control frequency (episode field or argument) the window is `summation_steps` (default: the action horizon);
    * prediction samples (`enable_prediction_training`): with probability `pred_prob` a frame becomes (frame t, frame min(t + m, T-1)),
      m = clamp(int(2.5 x frequency), 1, T-1), of the base camera (probability `primary_pred_prob`, or always without a wrist camera) or of
      the wrist camera, in the base / wrist slots; the label is the summed movement over those m steps (ZERO-padded window: past the end the
      tail, i.e. the gripper, reads 0 as in the reference), the horizon m / frequency (:534-590,603-697);
    * images (datasets/utils/image_utils.py:192-380): with `resize_to` every frame is resized with padding first; the wrist frame of a
      dataset in `WRIST_ROTATION_PATTERNS` is then rotated by 180 degrees (skipped with probability `not_rotate_wrist_prob`), for a
      prediction pair of the wrist camera both frames, for one of the base camera none; `rotation_applied` says what happened;
    * DROID (droid_dataset.py:104-232): an episode with `prompt_alternatives` / `base_0_rgb_alt` gets one of its instructions and one
      of its two exterior cameras per (seed, episode); frames whose `frame_mask` is False (the reference's idle-range filter, read from
      a table next to the shards) are not samples — they still belong to their neighbours' chunks and windows.
    Draws are a pure function of (seed, episode, step) (the reference keys TensorFlow's stateless generator with hashes of the trajectory id)."""

    def __init__(self, episodes: Sequence[dict] | str | pathlib.Path, *, action_horizon: int, summation_steps: int | None = None,
                 control_frequency: float | None = None, horizon_seconds: Sequence[float] = (1.0,), enable_prediction_training: bool = False,
                 pred_prob: float = 0.3, primary_pred_prob: float = 0.8, seed: int = 0, split: str | None = None,
                 val_fraction: float | None = None, resize_to: tuple | None = None, not_rotate_wrist_prob: float = 0.0):
        if isinstance(episodes, (str, pathlib.Path)):
            files = sorted(pathlib.Path(episodes).glob("*.npz"))
            if not files:
                raise FileNotFoundError(f"no episode files (*.npz) under {episodes}")
            episodes = [dict(np.load(f, allow_pickle=False)) for f in files]
        self.episodes = [self._check(e, i) for i, e in enumerate(episodes)]
        if split is not None:          # base_dataset.py:375-385: whole trajectories go to one side, by a salted hash of their id
            if split not in ("train", "val"):
                raise ValueError(f"split must be 'train', 'val' or None, got {split!r}")
            val = [self.is_val_episode(e, i, seed, val_fraction or 0.0) for i, e in enumerate(self.episodes)]
            self.episodes = [e for e, v in zip(self.episodes, val) if v == (split == "val")]
            if not self.episodes:
                raise ValueError(f"the {split} split of this store is empty (val_fraction = {val_fraction})")
        self.action_horizon = action_horizon
        self.summation_steps = summation_steps or action_horizon
        self.control_frequency, self.horizon_seconds = control_frequency, tuple(float(h) for h in horizon_seconds)
        self.enable_prediction_training, self.pred_prob, self.primary_pred_prob, self.seed = enable_prediction_training, pred_prob, primary_pred_prob, seed
        self.resize_to, self.not_rotate_wrist_prob = (tuple(resize_to) if resize_to else None), not_rotate_wrist_prob
        self._frames = [np.flatnonzero(np.asarray(e["frame_mask"], dtype=bool)) if "frame_mask" in e else np.arange(len(e["actions"])) for e in self.episodes]
        self._starts = np.cumsum([0] + [len(f) for f in self._frames])

    @classmethod
    def is_val_episode(cls, e: dict, index: int, seed: int, val_fraction: float) -> bool:
        """bucket(salt + trajectory id) < int(val_fraction x 1000) of 1000 buckets, the salt being the split seed (base_dataset.py:375-385).
        The reference hashes with TensorFlow's FarmHash, which does not exist here: the same rule with BLAKE2 picks other trajectories, the
        same share of them.  Id: the episode's `trajectory_id`, else `<dataset name>-<position in the store>`."""
        import hashlib
        tid = cls._text(e.get("trajectory_id")) or f"{cls._text(e.get('dataset_name'))}-{index}"
        bucket = int.from_bytes(hashlib.blake2b(f"{seed}{tid}".encode(), digest_size=8).digest(), "little") % 1000
        return bucket < int(val_fraction * 1000)

    @staticmethod
    def _text(v, default: str = "") -> str:
        if v is None:
            return default
        return v if isinstance(v, str) else str(np.asarray(v).item())

    @classmethod
    def _check(cls, e: dict, i: int) -> dict:
        for k in ("base_0_rgb", "state", "actions", "prompt"):
            if k not in e:
                raise KeyError(f"episode {i} has no '{k}'")
        T = len(e["actions"])
        if len(e["state"]) != T or len(e["base_0_rgb"]) != T or np.asarray(e["actions"]).shape[-1] < 7:
            raise ValueError(f"episode {i}: state / images / actions must share T and actions need >= 7 columns")
        mode = cls._text(e.get("chunk_mode"), "steps")
        if mode not in ("steps", "relative", "window_zero"):
            raise ValueError(f"episode {i}: unknown chunk_mode {mode!r}")
        if mode != "steps" and ("target_actions" not in e or len(e["target_actions"]) != T):
            raise ValueError(f"episode {i}: chunk_mode {mode!r} needs target_actions [T, A]")
        if cls._text(e.get("state_encoding")) == "pos_euler":        # base_dataset.py:437-456
            from lap_amd.rlds_export import euler_to_r6
            st = np.asarray(e["state"], dtype=np.float64)
            e = dict(e, state=np.concatenate([st[:, :3], euler_to_r6(st[:, 3:6]), st[:, 6:]], -1).astype(np.float32), state_encoding="pos_r6")
        return e

    def __len__(self) -> int:
        return int(self._starts[-1])

    @property
    def num_transitions(self) -> int:
        """The size this dataset enters the mixture weights with (dataset_mixer.py:141): the reference takes it from the statistics,
        which count the rows of the flattened action CHUNKS of every frame — frames x action horizon (normalize_adapter.py:108-113)."""
        return int(sum(len(e["actions"]) for e in self.episodes) * self.action_horizon)

    def _frequency(self, e: dict):
        f = e.get("control_frequency")
        return float(np.asarray(f)) if f is not None else self.control_frequency

    def chunks(self, e: dict, ts=None) -> np.ndarray:
        """The action chunks of frames `ts` (default: every frame) of one episode, [len(ts), action_horizon, A] (class docstring)."""
        H, T = self.action_horizon, len(e["actions"])
        ts = np.arange(T) if ts is None else np.asarray(ts, dtype=np.int64).reshape(-1)
        mode = self._text(e.get("chunk_mode"), "steps")
        idx = ts[:, None] + np.arange(H)[None]                      # [n, H]
        if mode == "steps":
            acts = np.asarray(e["actions"], dtype=np.float32)
            out = acts[np.minimum(idx, T - 1)].copy()
            out[..., :6] = np.where((idx >= T)[..., None], 0.0, out[..., :6])      # past the end: hold still, keep the gripper
            return out
        tgt = np.asarray(e["target_actions"], dtype=np.float64)
        if mode == "window_zero":
            out = tgt[np.minimum(idx, T - 1)].copy()
            out[idx >= T] = 0.0
            return out.astype(np.float32)
        from lap_amd.rlds_export import euler_diff
        w = tgt[np.minimum(ts[:, None] + np.arange(H + 1)[None], T - 1)]          # [n, H + 1, A]: last-value padding
        rot = euler_diff(w[:, 1:, 3:6], np.broadcast_to(w[:, 0:1, 3:6], w[:, 1:, 3:6].shape))
        return np.concatenate([w[:, 1:, :3] - w[:, 0:1, :3], rot, w[:, :-1, 6:7]], -1).astype(np.float32)

    def chunk(self, e: dict, t: int) -> np.ndarray:
        return self.chunks(e, [t])[0]

    def _draws(self, ep: int, t: int) -> np.ndarray:
        return np.random.Generator(np.random.Philox(key=self.seed, counter=[1, 0, ep, t])).random(4)

    def __getitem__(self, index: int) -> dict:
        ep = int(np.searchsorted(self._starts, index, side="right") - 1)
        t = int(self._frames[ep][int(index - self._starts[ep])])
        e = self.episodes[ep]
        acts = np.asarray(e["actions"], dtype=np.float32)
        T = len(acts)
        u = self._draws(ep, t)
        ue = np.random.Generator(np.random.Philox(key=self.seed, counter=[2, 0, ep, 0])).random(2)      # per-episode draws
        if "base_0_rgb_alt" in e and not ue[0] > 0.5:          # random_val > 0.5: the first camera, else the second
            e = dict(e, base_0_rgb=e["base_0_rgb_alt"])
        if "prompt_alternatives" in e:
            alts = [self._text(a) for a in np.asarray(e["prompt_alternatives"]).reshape(-1)]
            e = dict(e, prompt=alts[min(int(ue[1] * len(alts)), len(alts) - 1)])
        freq = self._frequency(e)
        if freq:
            steps = max(int(round(self.horizon_seconds[min(int(u[0] * len(self.horizon_seconds)), len(self.horizon_seconds) - 1)] * freq)), 1)
        else:
            steps = self.summation_steps
        used = max(min(steps, T - t), 1)
        lang = sum_language_actions(acts[t:t + used])
        horizon = used / freq if freq else None
        state = np.asarray(e["state"][t], dtype=np.float32)
        obs = {"base_0_rgb": e["base_0_rgb"][t], "state": state}
        has_wrist = "left_wrist_0_rgb" in e
        if has_wrist:
            obs["left_wrist_0_rgb"] = e["left_wrist_0_rgb"][t]
        sample = {"observation": obs, "prompt": self._text(e["prompt"]), "dataset_name": self._text(e.get("dataset_name")),
                  "actions": self.chunk(e, t), "language_actions": lang, "raw_state": state.copy(), "has_wrist_image": has_wrist,
                  "is_prediction_sample": False, "pred_use_primary": False}
        if horizon is not None:
            sample["time_horizon_seconds"] = float(horizon)
        if self.enable_prediction_training and u[1] < self.pred_prob:
            f = freq or float(self.summation_steps)            # (no clock: the summation window stands in for 1 s)
            m = max(min(int(2.5 * f), T - 1), 1)
            fut = min(t + m, T - 1)
            primary = (not has_wrist) or u[2] < self.primary_pred_prob
            cam = e["base_0_rgb"] if primary else e["left_wrist_0_rgb"]
            obs["base_0_rgb"], obs["left_wrist_0_rgb"] = cam[t], cam[fut]
            window = np.zeros((m, acts.shape[-1]), dtype=np.float32)          # zero-padded, NOT cut (sum_actions(window, deltas))
            window[:max(min(m, T - t), 0)] = acts[t:t + m]
            sample.update(is_prediction_sample=True, pred_use_primary=bool(primary), language_actions=sum_language_actions(window),
                          time_horizon_seconds=float(m / f))
        if self.resize_to is not None:
            for k in list(obs):
                if k != "state":
                    obs[k] = pio.dataset_resize_with_pad(np.asarray(obs[k]), *self.resize_to)
        rotated = False
        if needs_wrist_rotation(sample["dataset_name"]) and "left_wrist_0_rgb" in obs and not (sample["is_prediction_sample"] and sample["pred_use_primary"]):
            if not (self.not_rotate_wrist_prob > 0.0 and u[3] < self.not_rotate_wrist_prob):
                rotated = True
                obs["left_wrist_0_rgb"] = np.ascontiguousarray(np.asarray(obs["left_wrist_0_rgb"])[::-1, ::-1])
                if sample["is_prediction_sample"]:          # a pair of the wrist camera: both frames
                    obs["base_0_rgb"] = np.ascontiguousarray(np.asarray(obs["base_0_rgb"])[::-1, ::-1])
        sample["rotation_applied"] = rotated
        return sample


def episode_dataset_from_config(config, episodes, *, seed: int | None = None, split: str | None = None) -> EpisodeDataset:
    """An `EpisodeDataset` with the knobs the reference's dataset classes take from the train config (dataset_mixer.py:262-290,
    base_dataset.py:240-283): action horizon and prediction co-training from the model config; label windows, prediction probabilities
    from the data config."""
    dc, mc = config.data, config.model
    return EpisodeDataset(episodes, action_horizon=mc.action_horizon, horizon_seconds=tuple(getattr(dc, "horizon_seconds", (1.0,))),
                          enable_prediction_training=bool(getattr(mc, "enable_prediction_training", False)),
                          pred_prob=getattr(dc, "pred_prob", 0.3), primary_pred_prob=getattr(dc, "primary_pred_prob", 0.8),
                          seed=config.seed if seed is None else seed, split=split, val_fraction=getattr(dc, "val_fraction", None),
                          resize_to=getattr(dc, "resize_resolution", None), not_rotate_wrist_prob=getattr(dc, "not_rotate_wrist_prob", 0.0))


class VqaDataset:
    """Vision-language samples in a mixture (datasets/vqa/vqa_base.py:53-265 through output_schema.py:78-133): one frame per
    sample, a prompt and a caption, a zero state, zero actions, `is_vqa_sample`, the dataset's id.  Items are the dicts of
    `lap_amd.vqa_export.sample_from_record` — a list of them, or a directory of `.npz` files written by
    `tools/export_vqa_samples.py` (an encoded image is decoded with PIL on access).  Indexable like `EpisodeDataset`, carries no
    episodes (VQA sets are left out of the normalisation statistics, dataset_mixer.py:166-214)."""

    def __init__(self, samples: Sequence[dict] | str | pathlib.Path, *, action_horizon: int, action_dim: int = 7, state_dim: int = 7,
                 split: str | None = None, val_fraction: float | None = None, seed: int = 0):
        if isinstance(samples, (str, pathlib.Path)):
            files = sorted(pathlib.Path(samples).glob("*.npz"))
            if not files:
                raise FileNotFoundError(f"no sample files (*.npz) under {samples}")
            samples = [dict(np.load(f, allow_pickle=False)) for f in files]
        for i, e in enumerate(samples):
            for k in ("image", "prompt", "caption", "dataset_name", "vqa_dataset_id"):
                if k not in e:
                    raise KeyError(f"VQA sample {i} has no '{k}'")
        self.samples = list(samples)
        if split is not None:      # vqa_base.py:190-202: the same salted-hash rule as the robot sets, one sample = one trajectory
            if split not in ("train", "val"):
                raise ValueError(f"split must be 'train', 'val' or None, got {split!r}")
            val = [EpisodeDataset.is_val_episode(e, i, seed, val_fraction or 0.0) for i, e in enumerate(self.samples)]
            self.samples = [e for e, v in zip(self.samples, val) if v == (split == "val")]
            if not self.samples:
                raise ValueError(f"the {split} split of this VQA store is empty (val_fraction = {val_fraction})")
        self.action_horizon, self.action_dim, self.state_dim = action_horizon, action_dim, state_dim
        self.episodes: list = []

    def __len__(self) -> int:
        return len(self.samples)

    @property
    def num_transitions(self) -> int:
        """The reference's constant for the set (vqa_export.VQA_NUM_TRANSITIONS), whatever the store holds; an unknown set: its length."""
        from lap_amd.vqa_export import VQA_NUM_TRANSITIONS
        e = self.samples[0] if self.samples else {}
        name = e.get("dataset_name", "")
        name = name if isinstance(name, str) else str(np.asarray(name).item())
        return int(VQA_NUM_TRANSITIONS.get(name, len(self.samples)))

    @staticmethod
    def _image(e: dict) -> np.ndarray:
        img = np.asarray(e["image"])
        if bool(np.asarray(e.get("image_encoded", False))):
            import io

            from PIL import Image
            img = np.asarray(Image.open(io.BytesIO(img.tobytes())).convert("RGB"))
        return img

    def __getitem__(self, index: int) -> dict:
        e = self.samples[int(index)]
        txt = lambda v: str(np.asarray(v).item()) if not isinstance(v, str) else v
        return {"observation": {"base_0_rgb": self._image(e), "state": np.zeros(self.state_dim, dtype=np.float32)},
                "prompt": txt(e["prompt"]), "caption": txt(e["caption"]), "dataset_name": txt(e["dataset_name"]),
                "is_vqa_sample": True, "is_prediction_sample": False, "vqa_dataset_id": int(np.asarray(e["vqa_dataset_id"])),
                "time_horizon_seconds": 1.0, "actions": np.zeros((self.action_horizon, self.action_dim), dtype=np.float32),
                "language_actions": np.zeros(7, dtype=np.float32), "raw_state": np.zeros(self.state_dim, dtype=np.float32),
                "has_wrist_image": False}


def compute_norm_stats(dataset: EpisodeDataset, *, action_pad_to: int | None = None) -> dict:
    """shared/normalize_adapter.py `get_dataset_statistics` as the reference runs it (base_dataset.py:297-312): AFTER the trajectory
    transforms, i.e. over the action CHUNKS of every frame flattened to rows (mean / std / q01 / q99 / min / max, openpi NormStats
    fields; num_transitions counts those rows) and over the per-frame state as the model sees it ([xyz, rot6d, gripper] for end-effector
    states).  The mixer pads to the model's widths later (`global_norm_stats`); `action_pad_to` zero-pads here for single datasets:
    all-zero columns normalise to 0 by the q01 == q99 rule."""
    # Streaming form: the chunks of one episode at a time.  Rows are kept as float32 (the store's own dtype: the cast is a
    # no-op for stored data and is what the quantiles, min and max are taken from); mean / std come from per-episode (n, mean, M2) in
    # float64 merged by Chan's formula — E[x^2] - mean^2 loses the digits of a column with a large mean and a small spread.  Rows with a
    # non-finite entry (NaN proprioception in a few OXE episodes) are dropped from the statistics, states and chunks alike.  Peak memory
    # = 2 x rows x A x 4 bytes (+ one column's sort copy); `dataset.chunks` runs once per episode.
    def stats_of(rows_iter, width):
        parts, n = [], 0
        mean = np.zeros(width, dtype=np.float64); m2 = np.zeros(width, dtype=np.float64)
        lo = np.full(width, np.inf); hi = np.full(width, -np.inf)
        for r in rows_iter:
            r = r[np.isfinite(r).all(1)]
            if not len(r):
                continue
            parts.append(np.ascontiguousarray(r, dtype=np.float32))
            r64 = r.astype(np.float64)
            k, mk = len(r64), r64.mean(0)
            m2k = ((r64 - mk) ** 2).sum(0)
            d = mk - mean
            m2 += m2k + d * d * (n * k / (n + k))
            mean += d * (k / (n + k))
            lo = np.minimum(lo, r64.min(0)); hi = np.maximum(hi, r64.max(0))
            n += k
        buf = np.concatenate(parts, 0) if parts else np.zeros((0, width), dtype=np.float32)
        del parts
        std = np.sqrt(m2 / max(n, 1))
        q01 = np.array([np.quantile(buf[:, j].astype(np.float64), 0.01) for j in range(width)])
        q99 = np.array([np.quantile(buf[:, j].astype(np.float64), 0.99) for j in range(width)])
        return {"mean": mean.tolist(), "std": std.tolist(), "q01": q01.tolist(), "q99": q99.tolist(), "min": lo.tolist(), "max": hi.tolist()}

    eps = list(dataset.episodes)
    first = np.asarray(dataset.chunks(eps[0]))
    A = first.shape[-1]
    Aw = max(A, action_pad_to or 0)

    def chunk_rows():
        for i, e in enumerate(eps):
            c = first if i == 0 else np.asarray(dataset.chunks(e))
            c = c.reshape(-1, A).astype(np.float32, copy=False)
            yield c if Aw == A else pio.pad_to_dim(c, Aw, axis=-1)

    def state_rows():
        for e in eps:
            yield np.asarray(e["state"], dtype=np.float32).reshape(len(e["actions"]), -1)

    sw = np.asarray(eps[0]["state"]).reshape(len(eps[0]["actions"]), -1).shape[-1]
    return {"state": stats_of(state_rows(), sw), "actions": stats_of(chunk_rows(), Aw)}


# ------------------------------------------------------------------------------ dataset mixture (datasets/dataset_mixer.py)
# datasets/utils/mixtures.py: named mixtures = (dataset name, sampling weight) lists; the two the reference's LAP configs use
# plus the single-dataset entries every dataset name implies.
NAMED_MIXTURES: dict[str, list[tuple[str, float]]] = {
    "oxe_magic_soup": [
        ("bc_z", 0.05), ("droid", 2.0), ("fractal20220817_data", 1.0), ("bridge_v2_oxe", 1.0), ("taco_play", 2.0), ("jaco_play", 1.0),
        ("furniture_bench_dataset_converted_externally_to_rlds", 0.05), ("utaustin_mutex", 1.0), ("berkeley_fanuc_manipulation", 2.0),
        ("fmb", 0.05), ("berkeley_autolab_ur5", 1.0), ("austin_buds_dataset_converted_externally_to_rlds", 1.0),
        ("austin_sailor_dataset_converted_externally_to_rlds", 1.0), ("austin_sirius_dataset_converted_externally_to_rlds", 1.0),
        ("viola", 1.0), ("molmoact_dataset", 1.0)],
    "libero_finetune": [("libero_10_no_noops", 1.0), ("libero_spatial_no_noops", 1.0), ("libero_object_no_noops", 1.0),
                        ("libero_goal_no_noops", 1.0)],
}


def resolve_mixture(mix) -> list[tuple[str, float]]:
    """A named mixture, a single dataset name (weight 1, as the reference's per-dataset entries), or an explicit list."""
    if isinstance(mix, str):
        return list(NAMED_MIXTURES.get(mix, [(mix, 1.0)]))
    return [(str(n), float(w)) for n, w in mix]


def mixture_weights(sizes: Sequence[int], weights: Sequence[float], *, balance_weights: bool = True) -> tuple[np.ndarray, int]:
    """dataset_mixer.py:146-156: `balance_weights` multiplies each mixture weight by its dataset's size (number of transitions),
    the result is normalised; the effective length of the mixture is the largest size / weight — the number of draws after
    which the most under-sampled dataset has been seen once in expectation."""
    sizes, w = np.asarray(sizes, dtype=np.float64), np.asarray(weights, dtype=np.float64)
    if len(sizes) == 0 or len(sizes) != len(w) or (sizes <= 0).any() or (w <= 0).any():
        raise ValueError("a mixture needs one positive size and one positive weight per dataset")
    if balance_weights:
        w = w * sizes
    w = w / w.sum()
    return w, int((sizes / w).max())


def global_norm_stats(per_dataset: dict, *, action_dim: int, state_dim: int, state_types: dict | None = None,
                      exclude: Sequence[str] = ()) -> dict:
    """datasets/utils/statistics.py:45-236 (GlobalStatisticsBuilder): one set of normalisation statistics for a mixture from the
    per-dataset ones.  `per_dataset[name][key]` has mean / std / q01 / q99 / min / max vectors and `num_transitions` (key =
    "actions" or "state").  Mean and variance are the exact moments of the pooled data (weights = transitions, parallel-variance
    formula), q01 / min are the per-dimension minimum over datasets, q99 / max the maximum — the bounds of the pooled data can
    only be bracketed, not recovered, from per-dataset quantiles.  Vectors are cut / zero-padded to `action_dim` / `state_dim`
    (std padded with 0).  States are pooled per state type (`state_types[name]`, e.g. "eef_pose" / "joint_pos"; "none" or
    missing = skipped) into `state_<type>`; datasets in `exclude` (the VQA sets) contribute nothing."""
    def pad(v, n, fill=0.0):
        v = np.asarray(v, dtype=np.float32)[:n]
        return np.pad(v, (0, n - len(v)), constant_values=fill)

    def pool(names, key, dim):
        names = [n for n in names if key in per_dataset[n] and per_dataset[n][key].get("num_transitions", 0) > 0]
        total = sum(int(per_dataset[n][key]["num_transitions"]) for n in names)
        if total == 0:
            return None
        mean = sum(pad(per_dataset[n][key]["mean"], dim) * per_dataset[n][key]["num_transitions"] for n in names) / total
        var = sum(per_dataset[n][key]["num_transitions"] * (np.square(pad(per_dataset[n][key]["std"], dim)) +
                                                             np.square(pad(per_dataset[n][key]["mean"], dim) - mean)) for n in names) / total
        lo = lambda f: np.min([pad(per_dataset[n][key][f], dim) for n in names], axis=0)
        hi = lambda f: np.max([pad(per_dataset[n][key][f], dim) for n in names], axis=0)
        return {"mean": mean.astype(np.float32), "std": np.sqrt(var).astype(np.float32), "q01": lo("q01"), "q99": hi("q99"),
                "min": lo("min"), "max": hi("max"), "num_transitions": total,
                "num_trajectories": sum(int(per_dataset[n][key].get("num_trajectories", 0)) for n in names)}

    robot = [n for n in per_dataset if n not in set(exclude)]
    out = {}
    acts = pool(robot, "actions", action_dim)
    out["actions"] = acts if acts is not None else {
        "mean": np.zeros(action_dim, np.float32), "std": np.ones(action_dim, np.float32), "q01": np.zeros(action_dim, np.float32),
        "q99": np.zeros(action_dim, np.float32), "num_transitions": 0, "num_trajectories": 0}
    by_type: dict[str, list[str]] = {}
    for n in robot:
        t = (state_types or {}).get(n, "eef_pose")
        if t and t != "none":
            by_type.setdefault(t, []).append(n)
    for t, names in by_type.items():
        st = pool(names, "state", state_dim)
        if st is not None:
            out[f"state_{t}"] = st
    return out


class MixtureDataset:
    """Weighted mixture of episode datasets with the index protocol of `EpisodeDataset` (datasets/dataset_mixer.py:34-240).

    The reference interleaves endlessly repeated, shuffled per-dataset streams with `sample_from_datasets(weights, seed)`: every
    element of the mixed stream comes from dataset i with probability w_i.  Here element `index` of the mixture is a pure
    function of (seed, index): a counter-based generator picks the dataset by the same weights and then a transition of it
    uniformly — the same distribution, random access (what the resumable, rank-sharded `DataLoader` needs) instead of a stream.
    `len()` is the reference's effective mixture length (`mixture_weights`)."""

    def __init__(self, datasets: dict, mixture, *, balance_weights: bool = True, seed: int = 0):
        spec = resolve_mixture(mixture)
        missing = [n for n, _ in spec if n not in datasets]
        if missing:
            raise KeyError(f"mixture names datasets that were not provided: {missing}")
        self.names = [n for n, _ in spec]
        self.datasets = [datasets[n] for n in self.names]
        self.sizes = [len(d) for d in self.datasets]
        # the sizes the reference weights with are the statistics' transition counts (dataset_mixer.py:134-156): chunk rows for robot sets
        # (frames x action horizon), a per-set constant for VQA sets — not the number of samples
        self.weight_sizes = [int(getattr(d, "num_transitions", len(d))) for d in self.datasets]
        self.sample_weights, self.length = mixture_weights(self.weight_sizes, [w for _, w in spec], balance_weights=balance_weights)
        self._cdf = np.cumsum(self.sample_weights)
        self.seed = int(seed)
        horizons = {d.action_horizon for d in self.datasets}
        if len(horizons) != 1:
            raise ValueError(f"datasets of one mixture must share the action horizon, got {sorted(horizons)}")
        self.action_horizon = horizons.pop()

    def __len__(self) -> int:
        return self.length

    @property
    def num_distinct_samples(self) -> int:
        """Samples the member datasets actually hold (len() is the weighted EFFECTIVE length): what a one-pass validation loader is
        bounded by."""
        return int(sum(self.sizes))

    def set_epoch(self, epoch: int):
        """The draws are a pure function of (seed, epoch, index): a resumed loader reproduces them, a new epoch re-samples."""
        self.epoch = int(epoch)

    def locate(self, index: int) -> tuple[int, int]:
        """(dataset position in the mixture, transition index inside it) of mixture element `index`"""
        u = np.random.Generator(np.random.Philox(key=self.seed, counter=[0, 0, int(getattr(self, "epoch", 0)), int(index)])).random(2)
        d = min(int(np.searchsorted(self._cdf, u[0], side="right")), len(self.datasets) - 1)
        return d, min(int(u[1] * self.sizes[d]), self.sizes[d] - 1)

    def __getitem__(self, index: int) -> dict:
        d, i = self.locate(index)
        sample = self.datasets[d][i]
        if not sample.get("dataset_name"):
            sample["dataset_name"] = self.names[d]
        return sample

    @property
    def episodes(self):      # (compute_norm_stats over the pooled episodes; per-dataset statistics: call it per dataset)
        return [e for d in self.datasets for e in d.episodes]


def compute_mixture_norm_stats(mixture: MixtureDataset, *, action_pad_to: int, state_dim: int, state_types: dict | None = None) -> dict:
    """Per-dataset statistics (`compute_norm_stats`) pooled by `global_norm_stats`, in the norm_stats.json layout `Normalize` reads
    ("actions", "state"): the reference normalises every dataset of a mixture with the same global statistics
    (dataset_mixer.py:166-214)."""
    per = {}
    for name, ds in zip(mixture.names, mixture.datasets):
        if isinstance(ds, VqaDataset):       # no robot state / actions to normalise (dataset_mixer.py:166-214 skips VQA sets)
            continue
        st = compute_norm_stats(ds, action_pad_to=action_pad_to)
        n_tr, n_ep = int(getattr(ds, "num_transitions", len(ds))), len(ds.episodes)      # (chunk rows, as the reference's statistics count)
        per[name] = {k: {**{f: np.asarray(v[f], dtype=np.float32) for f in v}, "num_transitions": n_tr, "num_trajectories": n_ep} for k, v in st.items()}
    g = global_norm_stats(per, action_dim=action_pad_to, state_dim=state_dim, state_types=state_types)
    state = next((g[k] for k in sorted(g) if k.startswith("state_")), None)
    out = {"actions": {f: np.asarray(g["actions"][f]).tolist() for f in ("mean", "std", "q01", "q99", "min", "max") if f in g["actions"]}}
    if state is not None:
        out["state"] = {f: np.asarray(state[f]).tolist() for f in ("mean", "std", "q01", "q99", "min", "max")}
    return out


def _stack(samples: list[dict]) -> dict:
    """jax.tree.map(np.stack) over per-sample dicts (data_loader.py:111-121); None leaves must be None everywhere."""
    first = samples[0]
    out = {}
    for k, v in first.items():
        if isinstance(v, dict):
            out[k] = _stack([s[k] for s in samples])
        elif v is None:
            if any(s[k] is not None for s in samples):
                raise ValueError(f"field '{k}' is None in some samples of the batch only")
            out[k] = None
        elif isinstance(v, str):
            continue
        else:
            out[k] = np.stack([np.asarray(s[k]) for s in samples], axis=0)
    return out


class DataLoader:
    """Iterates `(CoTObservation, actions f32 [b, action_horizon, action_dim])`; see the module docstring for the contract."""

    def __init__(self, dataset, transform: Callable[[dict], dict], batch_size: int, *, shuffle: bool = True, seed: int = 0,
                 rank: int = 0, world_size: int = 1, num_batches: int | None = None, split: str = "train", device=None,
                 norm_stats: dict | None = None):
        if batch_size <= 0 or len(dataset) < batch_size * world_size:
            raise ValueError(f"dataset of {len(dataset)} samples cannot fill a global batch of {batch_size} x {world_size}")
        self.dataset, self.transform, self.batch_size = dataset, transform, batch_size
        self.shuffle, self.seed, self.rank, self.world = shuffle, seed, rank, world_size
        self.num_batches, self.split, self.device, self._norm_stats = num_batches, split, device, norm_stats
        self._seen_batches = 0
        self.per_epoch = len(dataset) // (batch_size * world_size)    # full global batches only (ragged tail dropped)
        # A mixture's len() is its EFFECTIVE length (transitions x action horizon / weight: sampling weights, dataset_mixer.py:216-240),
        # 20-40x its distinct samples; a "one pass" validation loader over it would draw that many times.  The val pass is
        # bounded by the number of distinct samples the mixture holds (the reference bounds it with val_max_samples / num_val_batches).
        distinct = getattr(dataset, "num_distinct_samples", None)
        self.val_batches = self.per_epoch if distinct is None else max(1, min(self.per_epoch, int(distinct) // (batch_size * world_size)))

    # ---- checkpoint protocol (train.main / checkpoints.save_state)
    def get_state(self) -> dict:
        return {"batches_seen": self._seen_batches, "seed": self.seed, "world_size": self.world}

    def set_state(self, s: dict):
        if int(s.get("world_size", self.world)) != self.world:
            raise ValueError("dataloader state was saved with a different world size")
        self._seen_batches, self.seed = int(s["batches_seen"]), int(s["seed"])

    def get_batches_seen(self) -> int:
        return self._seen_batches

    def get_norm_stats_for_checkpoint(self):
        return (self._norm_stats, "per-dataset") if self._norm_stats is not None else (None, "none")

    # ---- iteration
    def _indices(self, batch_index: int) -> np.ndarray:
        epoch, within = divmod(batch_index, self.per_epoch)
        if self.shuffle:
            # one permutation per EPOCH, kept until the epoch changes (a mixture's effective length is 20-40x its transitions:
            # rebuilding it for every batch cost seconds and gigabytes)
            if getattr(self, "_order_epoch", None) != epoch:
                self._order, self._order_epoch = np.random.RandomState(self.seed + 7919 * epoch).permutation(len(self.dataset)), epoch
            order = self._order
        else:
            order = np.arange(len(self.dataset))
        g0 = within * self.batch_size * self.world + self.rank * self.batch_size
        idx = order[g0:g0 + self.batch_size]
        if hasattr(self.dataset, "set_epoch"):
            self.dataset.set_epoch(epoch)      # a mixture draws NEW samples every epoch (the reference's stream never repeats a finite set)
        return idx

    def __iter__(self):
        produced = 0
        while True:
            if self.num_batches is not None and produced >= self.num_batches:
                return
            if self.split == "val" and self._seen_batches >= self.val_batches:
                return
            batch = _stack([self.transform(self.dataset[int(i)]) for i in self._indices(self._seen_batches)])
            self._seen_batches += 1
            produced += 1
            actions = torch.as_tensor(batch["actions"], dtype=torch.float32, device=self.device)
            yield CoTObservation.from_dict(batch, device=self.device), actions


def data_transform_inputs(dc, mc) -> "pio.CoTInputs":
    """`RLDSDataConfig._create_data_transforms` (training/config.py:321-352), inputs side: the ONE `CoTInputs` both the train loader and
    `create_trained_policy` use, with every field the reference passes from the data / model config."""
    question_config = None
    if getattr(dc, "enable_diverse_questions", False):      # :325-334
        from lap_amd.questions import QuestionConfig
        question_config = QuestionConfig(type_weights=getattr(dc, "question_type_weights", None),
                                         delta_motion_format_weights=getattr(dc, "delta_motion_format_weights", None),
                                         use_diverse_prompts=getattr(dc, "use_diverse_prompts", True))
    return pio.CoTInputs(
        action_dim=mc.action_dim, wrist_image_dropout_prob=getattr(dc, "wrist_image_dropout_prob", 0.0),
        language_action_format=getattr(dc, "language_action_format_name", "verbose_eef_with_rotation"),
        random_mask_prob=getattr(dc, "random_mask_prob", 0.0), random_base_prob=getattr(dc, "random_base_prob", 0.0),
        use_rough_scale=getattr(dc, "use_rough_scale", False), transform_strategy=getattr(dc, "transform_strategy", "standard"),
        enable_langact_training=mc.enable_langact_training, enable_diverse_questions=getattr(dc, "enable_diverse_questions", False),
        question_config=question_config)


def create_data_loader(config, dataset: "EpisodeDataset | MixtureDataset", tokenizer, *, norm_stats: dict | None = None, shuffle: bool = True, seed: int = 0,
                       rank: int = 0, world_size: int = 1, num_batches: int | None = None, split: str = "train", device=None) -> DataLoader:
    """datasets/data_loader.py:126-198: per-rank batch = config.batch_size // world_size; the transform stack of
    `training/config.py` (data transforms + Normalize + model transforms) for the LAP model type."""
    mc = config.model
    if norm_stats is None:
        if isinstance(dataset, MixtureDataset):     # one set of statistics for the whole mixture (dataset_mixer.py:166-214)
            state_dim = max(np.asarray(e["state"]).shape[-1] for e in dataset.episodes)
            norm_stats = compute_mixture_norm_stats(dataset, action_pad_to=mc.action_dim, state_dim=state_dim)
        else:
            norm_stats = compute_norm_stats(dataset, action_pad_to=mc.action_dim)
    dc = config.data
    ntype = getattr(dc, "action_proprio_normalization_type", "bounds_q99")
    cot = data_transform_inputs(dc, mc)
    # the training side's normalisation is the mixer's `NormalizeActionAndProprio` (clipped bounds, float32), not the policy side's
    # `Normalize` (dataset_mixer.py:334-359; the train-time transform group carries none, training/config.py:195-207).  The mixer runs
    # it on the raw trajectory, i.e. BEFORE `CoTInputs`; elementwise, so the order only matters to the VLA-0 strategy, whose label text
    # is made from the normalised chunk: there the sample's actions / state are normalised first.
    norm_robot = pio.NormalizeActionAndProprio(norm_stats, normalization_type=ntype)
    vla0 = getattr(dc, "transform_strategy", "standard") == "vla0"

    def norm(sample: dict) -> dict:
        # the mixer maps the normaliser over the ROBOT datasets only (dataset_mixer.py:338-341): a VQA sample keeps its all-zero
        # state and actions (2 * (0 - q01) / (q99 - q01) - 1 would make them non-zero)
        return sample if bool(sample.get("is_vqa_sample", False)) else norm_robot(sample)

    def norm_raw(sample: dict) -> dict:
        if bool(sample.get("is_vqa_sample", False)):
            return sample
        obs = dict(sample["observation"])
        flat = norm({"actions": sample["actions"], "state": obs.get("state")}) if "actions" in sample else {"state": norm({"actions": np.zeros(1), "state": obs.get("state")})["state"]}
        obs["state"] = flat["state"]
        return {**sample, "observation": obs, **({"actions": flat["actions"]} if "actions" in flat else {})}

    stack = pio.compose([
        *([norm_raw] if vla0 else []),
        cot,
        # (mixtures: cameras of different resolutions must batch — the reference's decode step resizes every frame, image_utils.py:192-267)
        *([pio.ResizeImages(mc.image_size, mc.image_size)] if isinstance(dataset, MixtureDataset) else []),
        *([] if vla0 else [norm]),
        pio.TokenizePromptAndReasoning(tokenizer, discrete_state_input=mc.discrete_state_input, verbose_mode=mc.verbose_mode,
                                       state_dropout=mc.state_dropout if split == "train" else 0.0),
        pio.PadStatesAndActions(mc.action_dim),
    ])
    return DataLoader(dataset, stack, max(1, config.batch_size // world_size), shuffle=shuffle, seed=seed, rank=rank,
                      world_size=world_size, num_batches=num_batches, split=split, device=device, norm_stats=norm_stats)
