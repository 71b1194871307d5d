"""Offline RLDS -> episode-store exporter: the record -> sample mapping of the reference's data path in numpy.

The reference reads RLDS / TFDS shards through TensorFlow + dlimp (`src/lap/datasets/robot/oxe_datasets.py`,
`droid_dataset.py`, `datasets/utils/transforms.py`): per dataset, a STANDARDISATION transform turns the raw trajectory into
`observation / action / language_action` with one convention — state = [xyz, extrinsic-XYZ euler, gripper (1 = open)],
language action = per-step end-effector delta `state[t+1] - state[t]` (rotation as the relative rotation's euler angles,
zero at the last step) + the gripper command (`transform_helpers.py:23-48`).  TensorFlow does not exist in this image, so the
arithmetic is restated here on numpy arrays and the part that needs `tensorflow_datasets` (iterating the shards) lives in
`tools/export_rlds_episodes.py`, which runs where TF exists and writes the `.npz` episode layout of `lap_amd/data.py`
(`base_0_rgb`, `left_wrist_0_rgb`, `state`, `actions`, `prompt`, `dataset_name`).

Built: the two dataset families the reference's LAP configs train on first — LIBERO (`lap_libero`: `libero_*_no_noops`,
transforms.py:1453-1481) and DROID (transforms.py:757-790) — plus the generic pieces every other OXE transform is made of.
Pinned by formula against scipy's rotations and by hand-computed trajectories (tests/test_data_cpu.py): the reference's
transforms cannot be imported here (module-level `import tensorflow`).
"""
from __future__ import annotations

import numpy as np

# datasets/utils/configs.py:208-272: where the raw RLDS observation keeps what the episode store calls base / wrist image
IMAGE_KEYS = {
    "droid": ("exterior_image_1_left", "wrist_image_left"),
    "libero_spatial_no_noops": ("image", "wrist_image"), "libero_object_no_noops": ("image", "wrist_image"),
    "libero_goal_no_noops": ("image", "wrist_image"), "libero_10_no_noops": ("image", "wrist_image"),
    "libero_combined": ("image", "wrist_image"),
}


# ------------------------------------------------------------------------------ rotations (datasets/utils/rotation_utils.py)
def euler_to_rotation_matrix(euler: np.ndarray) -> np.ndarray:
    """rotation_utils.py:84-119: extrinsic XYZ [roll, pitch, yaw] -> R = Rz(yaw) Ry(pitch) Rx(roll)."""
    e = np.asarray(euler, dtype=np.float64)
    roll, pitch, yaw = e[..., 0], e[..., 1], e[..., 2]
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.empty(e.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = -sp, cp * sr, cp * cr
    return R


def rotation_matrix_to_euler(R: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """rotation_utils.py:123-160 (gimbal lock: yaw = 0, roll from the middle row)."""
    R = np.asarray(R, dtype=np.float64)
    sy = np.sqrt(np.maximum(R[..., 0, 0] ** 2 + R[..., 1, 0] ** 2, eps))
    singular = sy < eps
    roll = np.where(singular, np.arctan2(-R[..., 1, 2], R[..., 1, 1]), np.arctan2(R[..., 2, 1], R[..., 2, 2]))
    pitch = np.arctan2(-R[..., 2, 0], sy)
    yaw = np.where(singular, 0.0, np.arctan2(R[..., 1, 0], R[..., 0, 0]))
    return np.stack([roll, pitch, yaw], -1)


def euler_diff(angles1: np.ndarray, angles2: np.ndarray) -> np.ndarray:
    """rotation_utils.py:453-471: angles_rel with R(angles2) R(angles_rel) = R(angles1)."""
    R1, R2 = euler_to_rotation_matrix(angles1), euler_to_rotation_matrix(angles2)
    return rotation_matrix_to_euler(np.swapaxes(R2, -1, -2) @ R1)


def axis_angle_to_extrinsic_xyz_euler(axis_angle: np.ndarray) -> np.ndarray:
    """transforms.py:103-133: rotation vector -> [roll, pitch, yaw] (Rodrigues, then the matrix entries the reference reads;
    pitch through asin of the clipped -r20)."""
    v = np.asarray(axis_angle, dtype=np.float64)
    ang = np.linalg.norm(v, axis=-1, keepdims=True)
    small = ang < 1e-8
    axis = np.where(small, np.array([1.0, 0.0, 0.0]), v / np.where(small, 1.0, ang))
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    a = ang[..., 0]
    c, s, C = np.cos(a), np.sin(a), 1.0 - np.cos(a)
    r00, r10 = c + x * x * C, y * x * C + z * s
    r20, r21, r22 = z * x * C - y * s, z * y * C + x * s, c + z * z * C
    return np.stack([np.arctan2(r21, r22), np.arcsin(np.clip(-r20, -1.0, 1.0)), np.arctan2(r10, r00)], -1)


# ------------------------------------------------------------------------------ transform_helpers.py
def compute_padded_movement_actions(eef_state: np.ndarray) -> np.ndarray:
    """transform_helpers.py:23-48: action[t] = state[t+1] - state[t] (rotation: euler_diff), zeros at the last step; [T, 6]."""
    s = np.asarray(eef_state, dtype=np.float64)
    mov = np.concatenate([s[1:, :3] - s[:-1, :3], euler_diff(s[1:, 3:6], s[:-1, 3:6])], -1)
    return np.concatenate([mov, np.zeros((1, 6))], 0)


def invert_gripper_actions(a: np.ndarray) -> np.ndarray:
    return 1.0 - np.asarray(a, dtype=np.float64)


def binarize_gripper_actions(actions: np.ndarray, threshold: float = 0.95) -> np.ndarray:
    """transform_helpers.py:133-161: > threshold open (1), < 1 - threshold closed (0); in-between steps take the value of the
    NEXT decided step (reverse scan seeded with the last action's raw value)."""
    a = np.asarray(actions, dtype=np.float64)
    flat = a.reshape(len(a), -1)
    out = np.empty_like(flat)
    carry = flat[-1].astype(np.float64).copy()
    for i in range(len(flat) - 1, -1, -1):
        open_m, closed_m = flat[i] > threshold, flat[i] < 1.0 - threshold
        carry = np.where(open_m | closed_m, open_m.astype(np.float64), carry)
        out[i] = carry
    return out.reshape(a.shape)


# ------------------------------------------------------------------------------ per-dataset standardisation
def libero_dataset_transform(traj: dict) -> dict:
    """transforms.py:1453-1481.  Raw: `action` [T, 7] (gripper in -1 open .. +1 close), `observation.state` [T, 8] =
    [xyz, axis-angle, 2 finger joints].  Out: action gripper = 1 - clip(g, 0, 1) (1 = open); state = [xyz, euler,
    clip(finger / 0.04, 0, 1)]; language_action = [padded movement of the state, gripper]."""
    act = np.asarray(traj["action"], dtype=np.float64)
    grip = invert_gripper_actions(np.clip(act[:, -1:], 0.0, 1.0))
    st = np.asarray(traj["observation"]["state"], dtype=np.float64)
    state = np.concatenate([st[:, :3], axis_angle_to_extrinsic_xyz_euler(st[:, 3:6]), np.clip(st[:, -2:-1] / 0.04, 0.0, 1.0)], 1)
    out = dict(traj)
    out["action"] = np.concatenate([act[:, :6], grip], 1)
    out["observation"] = dict(traj["observation"], state=state)
    out["language_action"] = np.concatenate([compute_padded_movement_actions(state[:, :6]), grip], 1)
    return out


def droid_dataset_transform(traj: dict) -> dict:
    """transforms.py:757-790.  Raw: `observation.cartesian_position` [T, 6], `observation.gripper_position` [T] or [T, 1]
    (0 open .. 1 closed), `action_dict.gripper_position`.  Out: state = [cartesian, binarized open-ness]; language_action =
    [padded movement of the cartesian pose, clipped binarized gripper command]; action = [cartesian, the same gripper]."""
    cart = np.asarray(traj["observation"]["cartesian_position"], dtype=np.float64)
    grip = np.asarray(traj["observation"]["gripper_position"], dtype=np.float64)
    if grip.ndim != cart.ndim:
        grip = grip[..., None]
    state = np.concatenate([cart, binarize_gripper_actions(invert_gripper_actions(grip), threshold=0.5)], -1)
    ga = np.asarray(traj["action_dict"]["gripper_position"], dtype=np.float64)
    ga = ga[..., None] if ga.ndim == 1 else ga
    gact = np.clip(binarize_gripper_actions(invert_gripper_actions(ga), threshold=0.5)[:, -1:], 0.0, 1.0)
    out = dict(traj)
    out["state"] = state
    out["language_action"] = np.concatenate([compute_padded_movement_actions(cart), gact], 1)
    out["action"] = np.concatenate([cart, gact], 1)
    return out


STANDARDIZE = {"droid": droid_dataset_transform, **{k: libero_dataset_transform for k in IMAGE_KEYS if k.startswith("libero")}}


def decode_instruction(x) -> str:
    if isinstance(x, (bytes, np.bytes_)):
        return x.decode("utf-8")
    a = np.asarray(x)
    if a.ndim:                    # per-step copies of one instruction (RLDS): the first
        return decode_instruction(a.reshape(-1)[0])
    v = a.item()
    return v.decode("utf-8") if isinstance(v, bytes) else str(v)


def episode_from_rlds(dataset_name: str, traj: dict) -> dict | None:
    """One raw RLDS trajectory (dict of numpy arrays stacked over steps: `observation`, `action` [, `action_dict`],
    `language_instruction`) -> the episode-store dict of `lap_amd/data.py`, or None when the reference's filters drop it
    (empty instruction, zero length: oxe_datasets.py SingleOXEDataset "standard filtering").  `actions` are the per-step
    language actions [dx, dy, dz, droll, dpitch, dyaw, gripper] the label text is summed from; `state` keeps
    [xyz, euler, gripper]."""
    if dataset_name not in STANDARDIZE:
        raise KeyError(f"no standardisation transform for {dataset_name!r} (built: {sorted(STANDARDIZE)})")
    prompt = decode_instruction(traj["language_instruction"]) if "language_instruction" in traj else ""
    T = len(np.asarray(traj["action"] if "action" in traj else traj["action_dict"]["gripper_position"]))
    if not prompt.strip() or T == 0:
        return None
    std = STANDARDIZE[dataset_name](traj)
    base_key, wrist_key = IMAGE_KEYS[dataset_name]
    obs = std["observation"]
    state = std["state"] if "state" in std else obs["state"]
    ep = {"base_0_rgb": np.asarray(obs[base_key], dtype=np.uint8), "state": np.asarray(state, dtype=np.float32),
          "actions": np.asarray(std["language_action"], dtype=np.float32), "prompt": prompt, "dataset_name": dataset_name}
    if wrist_key and wrist_key in obs and np.asarray(obs[wrist_key]).size:
        ep["left_wrist_0_rgb"] = np.asarray(obs[wrist_key], dtype=np.uint8)
    return ep


def write_episode(path, ep: dict):
    np.savez_compressed(path, **ep)
