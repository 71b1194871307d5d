"""Offline RLDS -> episode-store exporter: the record -> sample mapping of the reference's data path in numpy.

The reference reads RLDS / TFDS shards through TensorFlow + dlimp (`src/lap/datasets/robot/oxe_datasets.py`,
`droid_dataset.py`, `datasets/utils/transforms.py`): per dataset, a STANDARDISATION transform turns the raw trajectory into
`observation / action / language_action` with one convention — state = [xyz, extrinsic-XYZ euler, gripper (1 = open)],
language action = per-step end-effector delta `state[t+1] - state[t]` (rotation as the relative rotation's euler angles,
zero at the last step) + the gripper command (`transform_helpers.py:23-48`).  TensorFlow does not exist in this image, so the
arithmetic is restated here on numpy arrays and the part that needs `tensorflow_datasets` (iterating the shards) lives in
`tools/export_rlds_episodes.py`, which runs where TF exists and writes the `.npz` episode layout of `lap_amd/data.py`
(`base_0_rgb`, `left_wrist_0_rgb`, `state`, `actions`, `prompt`, `dataset_name`).

Built: LIBERO (`lap_libero`: `libero_*_no_noops`, transforms.py:1453-1481), DROID (transforms.py:757-790) and — round 4 — the
other fourteen datasets of the `lap` config's training mixture `oxe_magic_soup` (mixtures.py:2-22): bc_z, fractal20220817_data
(RT-1), bridge_v2_oxe, taco_play, jaco_play, furniture_bench, utaustin_mutex, berkeley_fanuc_manipulation, fmb,
berkeley_autolab_ur5, austin_buds / sailor / sirius, viola, molmoact_dataset — with the helpers they are made of
(transform_helpers.py, rotation_utils.py:382-450).  Pinned by formula against scipy's rotations and by hand-computed trajectories
(tests/test_data_cpu.py): the reference's transforms cannot be imported here (module-level `import tensorflow`).
Two things the reference takes from TensorFlow cannot be restated bit for bit and are INJECTED by the caller instead
(tools/export_rlds_episodes.py runs where TF exists): the FarmHash bucket behind the deterministic fallback instruction of the
language-free Austin datasets (`hash_bucket`), and tensorflow_graphics' quaternion -> Euler conversion at exact gimbal lock
([UPSTREAM-RECALL], `quaternion_xyzw_to_euler`).
"""
from __future__ import annotations

import numpy as np

# datasets/utils/configs.py:208-272: where the raw RLDS observation keeps what the episode store calls base / wrist image
IMAGE_KEYS = {
    "droid": ("exterior_image_1_left", "wrist_image_left"),
    "bc_z": ("image", None), "fractal20220817_data": ("image", None), "bridge_v2_oxe": ("image_0", None),
    "taco_play": ("rgb_static", "rgb_gripper"), "jaco_play": ("image", "image_wrist"),
    "furniture_bench_dataset_converted_externally_to_rlds": ("image", "wrist_image"), "utaustin_mutex": ("image", "wrist_image"),
    "berkeley_fanuc_manipulation": ("image", "wrist_image"), "fmb": ("image_side_1", "image_wrist_2"),
    "berkeley_autolab_ur5": ("image", "hand_image"), "austin_buds_dataset_converted_externally_to_rlds": ("image", "wrist_image"),
    "austin_sailor_dataset_converted_externally_to_rlds": ("image", "wrist_image"),
    "austin_sirius_dataset_converted_externally_to_rlds": ("image", "wrist_image"), "viola": ("agentview_rgb", "eye_in_hand_rgb"),
    "molmoact_dataset": ("first_view_image", "wrist_image"),
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


def euler_to_r6(euler: np.ndarray) -> np.ndarray:
    """rotation_utils.py:303-315,351-361: the first two COLUMNS of R(euler): [r11, r21, r31, r12, r22, r32]."""
    R = euler_to_rotation_matrix(euler)
    return np.concatenate([R[..., :, 0], R[..., :, 1]], -1)


# configs.py:335-521 (OXE_DATASET_METADATA): control frequency in Hz of the datasets standardised here (held to the reference's table by
# tests/golden/dataset_configs_v1.json).  It sets the summation window of the label text (horizon_seconds x frequency steps,
# base_dataset.py:493-531) and the look-ahead of prediction samples (2.5 s, base_dataset.py:542-551).
CONTROL_FREQUENCY: dict[str, int] = {
    "austin_buds_dataset_converted_externally_to_rlds": 20, "austin_sailor_dataset_converted_externally_to_rlds": 20,
    "austin_sirius_dataset_converted_externally_to_rlds": 20, "bc_z": 30, "berkeley_autolab_ur5": 5, "berkeley_fanuc_manipulation": 10,
    "bridge_v2_oxe": 5, "droid": 15, "fmb": 10, "fractal20220817_data": 3, "furniture_bench_dataset_converted_externally_to_rlds": 10,
    "jaco_play": 10, "libero_10_no_noops": 15, "libero_goal_no_noops": 15, "libero_object_no_noops": 15, "libero_spatial_no_noops": 15,
    "molmoact_dataset": 15, "taco_play": 15, "utaustin_mutex": 20, "viola": 20,
}


def chunk_mode_of(dataset_name: str) -> str:
    """Which `chunk_actions` the reference's dataset class runs: LIBERO a zero-padded window of the raw controller actions
    (oxe_datasets.py:259-269), every other end-effector dataset incl. DROID the displacement from the CURRENT pose over a last-value-padded
    window of absolute poses (base_dataset.py:387-427)."""
    return "window_zero" if dataset_name.startswith("libero") else "relative"


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


# ------------------------------------------------------------------------------ more helpers (round 4)
def quaternion_xyzw_to_euler(q: np.ndarray) -> np.ndarray:
    """tensorflow_graphics `euler.from_quaternion` as the reference calls it (transforms.py:312,584,801,1228,1346): [x, y, z, w] ->
    [theta_x, theta_y, theta_z] with R = Rz Ry Rx, i.e. the extrinsic XYZ angles of the rest of this module.  [UPSTREAM-RECALL]:
    restated through the rotation matrix (r21 / r22, -asin r20, r10 / r00); at exact gimbal lock (|r20| = 1) this module's
    `rotation_matrix_to_euler` convention (yaw = 0) is used, which may differ from the library's choice there."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)
    return rotation_matrix_to_euler(R)


def matrix_to_xyzrpy(T: np.ndarray) -> np.ndarray:
    """rotation_utils.py:504-518: [.., 4, 4] homogeneous transform -> [x, y, z, roll, pitch, yaw]."""
    T = np.asarray(T, dtype=np.float64)
    return np.concatenate([T[..., :3, 3], rotation_matrix_to_euler(T[..., :3, :3])], -1)


def extract_state_from_matrix(flat16: np.ndarray, gripper: np.ndarray, gripper_scale: float = 0.079) -> np.ndarray:
    """transform_helpers.py:57-82: the 16 numbers are a COLUMN-major 4 x 4 pose; state = [xyz, rpy, clip(gripper / scale, 0, 1)]."""
    T = np.swapaxes(np.asarray(flat16, dtype=np.float64).reshape(-1, 4, 4), 1, 2)
    return np.concatenate([matrix_to_xyzrpy(T), np.clip(np.asarray(gripper, dtype=np.float64) / gripper_scale, 0.0, 1.0)], -1)


def rel2abs_gripper_actions(actions: np.ndarray) -> np.ndarray:
    """transform_helpers.py:165-189: relative commands (+ closing, - opening; |a| <= 0.1 = none) -> absolute open-ness: the state
    holds between commands, starts as the opposite of the first command (open when there is none); 0 = closed, 1 = open."""
    a = np.asarray(actions, dtype=np.float64).reshape(-1)
    th = np.where(a < -0.1, 1, np.where(a > 0.1, -1, 0))
    nz = np.nonzero(th)[0]
    carry = -int(th[nz[0]]) if len(nz) else 0
    if carry == 0:
        carry = 1
    out = np.empty(len(a))
    for i, t in enumerate(th):
        if t != 0:
            carry = int(t)
        out[i] = carry
    return out / 2.0 + 0.5


def apply_coordinate_transform(movement: np.ndarray, C: np.ndarray) -> np.ndarray:
    """rotation_utils.py:382-417: [xyz, extrinsic-XYZ euler] into another frame: t' = C t, R' = C R C^T."""
    m = np.asarray(movement, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    R = C @ euler_to_rotation_matrix(m[..., 3:6]) @ C.T
    return np.concatenate([m[..., :3] @ C.T, rotation_matrix_to_euler(R)], -1)


TRANSFORM_BCZ = np.array([[0.0, -1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])     # rotation_utils.py:419-421: x' = -y, y' = -x, z' = -z
TRANSFORM_JACO = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])      # rotation_utils.py:426-427: x' = -y, y' = x, z' = z

FALLBACK_INSTRUCTIONS = (      # constants.py:8-30
    "Do something useful.", "Complete the task.", "Perform the task.", "Carry out the objective.", "Execute the current task.",
    "Accomplish the goal.", "Proceed with the task.", "Handle the task at hand.", "Continue the operation.", "Fulfill the task.",
    "Take meaningful steps.", "Demonstrate useful behavior.", "Act in a useful manner.", "Engage in productive actions.",
    "Make useful moves.", "Undertake useful actions.", "Behave purposefully.", "Start the activity.")


def fill_empty_language_instruction(instruction: str, first_state_sum: float, *, deterministic: bool = True, hash_bucket=None, rng=None) -> str:
    """transform_helpers.py:90-125: an empty (or blank) instruction is replaced by one of 18 stock phrases — deterministic: index =
    FarmHash bucket of the decimal string of sum(state[0]) (`tf.strings.to_hash_bucket_fast`: supplied by the caller as
    `hash_bucket(value, 18)`, see the module docstring); else a random one (`rng`: numpy Generator)."""
    if instruction.strip():
        return instruction
    if deterministic:
        if hash_bucket is None:
            raise ValueError("the deterministic fallback instruction needs hash_bucket(value, n) = tf.strings.to_hash_bucket_fast(tf.strings.as_string(value), n)")
        return FALLBACK_INSTRUCTIONS[int(hash_bucket(first_state_sum, len(FALLBACK_INSTRUCTIONS)))]
    rng = rng or np.random.default_rng()
    return FALLBACK_INSTRUCTIONS[int(rng.integers(len(FALLBACK_INSTRUCTIONS)))]


def _finish(traj: dict, state: np.ndarray, gripper_action: np.ndarray, instruction=None) -> dict:
    """What every transform below ends with: language_action = [padded movement of the state's pose, gripper command], action =
    [pose, gripper command]."""
    out = dict(traj)
    out["state"] = state
    out["language_action"] = np.concatenate([compute_padded_movement_actions(state[:, :6]), gripper_action], 1)
    out["action"] = np.concatenate([state[:, :6], gripper_action], 1)
    if instruction is not None:
        out["language_instruction"] = instruction
    return out


def _f(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float64)


def _col(x) -> np.ndarray:
    x = _f(x)
    return x[:, None] if x.ndim == 1 else x


# ------------------------------------------------------------------------------ the rest of `oxe_magic_soup` (mixtures.py:2-22)
def bridge_v2_oxe_dataset_transform(traj: dict) -> dict:
    """transforms.py:174-242 (the project-website version of Bridge V2): the first step (an all-zero action) is dropped everywhere;
    gripper command binarised; state = [state[:, :6], clip(state[:, -1], 0, 1)]."""
    cut = lambda v: {k: cut(x) for k, x in v.items()} if isinstance(v, dict) else np.asarray(v)[1:]
    t = {k: (v if k == "traj_metadata" else cut(v)) for k, v in traj.items()}
    act, st = _f(t["action"]), _f(t["observation"]["state"])
    grip = binarize_gripper_actions(act[:, -1])[:, None]
    state = np.concatenate([st[:, :6], np.clip(st[:, -1:], 0.0, 1.0)], 1)
    return _finish(t, state, grip)


def rt1_dataset_transform(traj: dict) -> dict:
    """transforms.py:288-328 (fractal20220817_data): gripper command relative -> absolute; pose = base_pose_tool_reached [xyz, quat
    xyzw]; gripper state = clip(1 - gripper_closed, 0, 1)."""
    obs = traj["observation"]
    grip = rel2abs_gripper_actions(_f(traj["action"]["gripper_closedness_action"])[:, 0])[:, None]
    pose = _f(obs["base_pose_tool_reached"])
    state = np.concatenate([pose[:, :3], quaternion_xyzw_to_euler(pose[:, 3:7]), np.clip(invert_gripper_actions(_col(obs["gripper_closed"])), 0.0, 1.0)], 1)
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def taco_play_dataset_transform(traj: dict) -> dict:
    """transforms.py:397-434: pose = robot_obs[:, :6]; gripper state = clip(12.3903 robot_obs[:, 6], 0, 1); command = clip((a + 1) / 2)."""
    obs = traj["observation"]
    ro = _f(obs["robot_obs"])
    grip = np.clip((_f(traj["action"]["rel_actions_world"])[:, -1:] + 1.0) / 2.0, 0.0, 1.0)
    state = np.concatenate([ro[:, :6], np.clip(12.3903 * ro[:, 6:7], 0.0, 1.0)], 1)
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def jaco_play_dataset_transform(traj: dict) -> dict:
    """transforms.py:437-475: pose = end_effector_cartesian_pos[:, :6] in the frame x' = -y, y' = x; gripper state = clip(4.33 x last)."""
    obs = traj["observation"]
    ee = _f(obs["end_effector_cartesian_pos"])
    grip = rel2abs_gripper_actions(_f(traj["action"]["gripper_closedness_action"])[:, 0])[:, None]
    state = np.concatenate([apply_coordinate_transform(ee[:, :6], TRANSFORM_JACO), np.clip(ee[:, -1:] * 4.33, 0.0, 1.0)], 1)
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def viola_dataset_transform(traj: dict) -> dict:
    """transforms.py:534-575: pose from the column-major ee_states matrix; gripper state = clip(gripper_states / 0.079); command =
    1 - clip(closedness, 0, 1)."""
    obs = traj["observation"]
    grip = invert_gripper_actions(np.clip(_col(traj["action"]["gripper_closedness_action"]), 0.0, 1.0))
    state = extract_state_from_matrix(_f(obs["ee_states"])[:, -16:], _col(obs["gripper_states"]))
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def berkeley_autolab_ur5_dataset_transform(traj: dict) -> dict:
    """transforms.py:578-619: robot_state[:, 6:14] = [xyz, quat xyzw, gripper closed]; command relative -> absolute."""
    obs = traj["observation"]
    rs = _f(obs["robot_state"])[:, 6:14]
    grip = rel2abs_gripper_actions(_f(traj["action"]["gripper_closedness_action"]))[:, None]
    state = np.concatenate([rs[:, :3], quaternion_xyzw_to_euler(rs[:, 3:7]), np.clip(invert_gripper_actions(rs[:, -1:]), 0.0, 1.0)], 1)
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def _matrix_state_transform(traj: dict, matrix_key: str, gripper_cols: slice, *, deterministic: bool | None, hash_bucket=None, rng=None) -> dict:
    """transform_helpers.py:200-268 and its three hand-written variants (transforms.py:869-920, 1148-1181): pose from a column-major
    4 x 4 matrix, gripper state = clip(state[gripper_cols] / 0.079), command = 1 - clip(action[:, -1], 0, 1); `deterministic` None:
    the instruction is left alone, else an empty one is filled in."""
    obs = traj["observation"]
    st = _f(obs["state"])
    state = extract_state_from_matrix(_f(obs[matrix_key])[:, -16:], st[:, gripper_cols])
    grip = invert_gripper_actions(np.clip(_f(traj["action"])[:, -1:], 0.0, 1.0))
    out = _finish(traj, state, grip)
    if deterministic is not None:
        out["language_instruction"] = fill_empty_language_instruction(decode_instruction(traj.get("language_instruction", "")), float(st[0].astype(np.float32).sum()),
                                                                      deterministic=deterministic, hash_bucket=hash_bucket, rng=rng)
    return out


def austin_buds_dataset_transform(traj: dict, **kw) -> dict:       # transforms.py:716-729
    return _matrix_state_transform(traj, "state", slice(7, 8), deterministic=True, **kw)


def austin_sailor_dataset_transform(traj: dict, **kw) -> dict:     # transforms.py:869-893
    return _matrix_state_transform(traj, "state_ee", slice(-1, None), deterministic=True, **kw)


def austin_sirius_dataset_transform(traj: dict, **kw) -> dict:     # transforms.py:896-920 (a RANDOM fallback instruction)
    return _matrix_state_transform(traj, "state_ee", slice(-1, None), deterministic=False, **kw)


def utaustin_mutex_dataset_transform(traj: dict) -> dict:          # transforms.py:1148-1181
    return _matrix_state_transform(traj, "state", slice(7, 8), deterministic=None)


def furniture_bench_dataset_transform(traj: dict) -> dict:
    """transforms.py:798-824: state = [xyz, quat xyzw -> euler, clip(last / 0.079)]; command = 1 - clip(action[:, -1], 0, 1)."""
    st = _f(traj["observation"]["state"])
    state = np.concatenate([st[:, :3], quaternion_xyzw_to_euler(st[:, 3:7]), np.clip(st[:, -1:] / 0.079, 0.0, 1.0)], 1)
    return _finish(traj, state, invert_gripper_actions(np.clip(_f(traj["action"])[:, -1:], 0.0, 1.0)))


def bc_z_dataset_transform(traj: dict) -> dict:
    """transforms.py:923-966: pose = [present/xyz, axis-angle -> euler] in the frame x' = -y, y' = -x, z' = -z; gripper state =
    clip((1 - sensed_close) / 0.8); command = 1 - future/target_close."""
    obs = traj["observation"]
    grip = invert_gripper_actions(_col(traj["action"]["future/target_close"]))[:, :1]
    pose = np.concatenate([_f(obs["present/xyz"])[:, :3], axis_angle_to_extrinsic_xyz_euler(_f(obs["present/axis_angle"])[:, :3])], 1)
    state = np.concatenate([apply_coordinate_transform(pose, TRANSFORM_BCZ), np.clip(invert_gripper_actions(_col(obs["present/sensed_close"]))[:, :1] / 0.8, 0.0, 1.0)], 1)
    return _finish(traj, state, grip, obs["natural_language_instruction"])


def molmoact_dataset_transform(traj: dict) -> dict:
    """transforms.py:1184-1206: the stored action IS the language action; only the gripper conventions are inverted."""
    act, st = _f(traj["action"]), _f(traj["observation"]["state"])
    grip = invert_gripper_actions(act[:, -1:])
    out = dict(traj)
    out["state"] = np.concatenate([st[:, :-1], invert_gripper_actions(st[:, -1:])], 1)
    out["language_action"] = np.concatenate([act[:, :-1], grip], 1)
    out["action"] = np.concatenate([out["state"][:, :6], grip], 1)
    return out


def berkeley_fanuc_dataset_transform(traj: dict) -> dict:
    """transforms.py:1209-1241: no gripper commands are stored — the gripper STATE (inverted) stands in; the stored action is the
    language action's movement part as it is; state = [end_effector_state xyz, quat xyzw -> euler, clip(1 - state[:, 6])]."""
    obs = traj["observation"]
    st, ee = _f(obs["state"]), _f(obs["end_effector_state"])
    grip = invert_gripper_actions(st[:, 6:7])
    out = dict(traj)
    out["state"] = np.concatenate([ee[:, :3], quaternion_xyzw_to_euler(ee[:, 3:7]), np.clip(grip, 0.0, 1.0)], 1)
    out["language_action"] = np.concatenate([_f(traj["action"]), grip], 1)
    out["action"] = np.concatenate([out["state"][:, :6], grip], 1)
    return out


def fmb_dataset_transform(traj: dict) -> dict:
    """transforms.py:1340-1366: state = [eef_pose xyz, quat xyzw -> euler, clip(1 - state_gripper_pose)]; command = 1 - action[:, -1]."""
    obs = traj["observation"]
    ee = _f(obs["eef_pose"])
    state = np.concatenate([ee[:, :3], quaternion_xyzw_to_euler(ee[:, 3:7]), np.clip(invert_gripper_actions(_col(obs["state_gripper_pose"])), 0.0, 1.0)], 1)
    return _finish(traj, state, invert_gripper_actions(_f(traj["action"])[:, -1:]))


STANDARDIZE = {
    "droid": droid_dataset_transform, **{k: libero_dataset_transform for k in IMAGE_KEYS if k.startswith("libero")},
    "bc_z": bc_z_dataset_transform, "fractal20220817_data": rt1_dataset_transform, "bridge_v2_oxe": bridge_v2_oxe_dataset_transform,
    "taco_play": taco_play_dataset_transform, "jaco_play": jaco_play_dataset_transform,
    "furniture_bench_dataset_converted_externally_to_rlds": furniture_bench_dataset_transform, "utaustin_mutex": utaustin_mutex_dataset_transform,
    "berkeley_fanuc_manipulation": berkeley_fanuc_dataset_transform, "fmb": fmb_dataset_transform,
    "berkeley_autolab_ur5": berkeley_autolab_ur5_dataset_transform, "austin_buds_dataset_converted_externally_to_rlds": austin_buds_dataset_transform,
    "austin_sailor_dataset_converted_externally_to_rlds": austin_sailor_dataset_transform,
    "austin_sirius_dataset_converted_externally_to_rlds": austin_sirius_dataset_transform, "viola": viola_dataset_transform,
    "molmoact_dataset": molmoact_dataset_transform,
}
# transforms that may have to invent an instruction take `hash_bucket=` / `rng=` (fill_empty_language_instruction)
NEEDS_FALLBACK = {"austin_buds_dataset_converted_externally_to_rlds", "austin_sailor_dataset_converted_externally_to_rlds",
                  "austin_sirius_dataset_converted_externally_to_rlds"}


def decode_instruction(x) -> str:
    if isinstance(x, (bytes, np.bytes_)):
        return x.decode("utf-8")
    a = np.asarray(x)
    if a.ndim:                    # per-step copies of one instruction (RLDS): the first
        return decode_instruction(a.reshape(-1)[0])
    v = a.item()
    return v.decode("utf-8") if isinstance(v, bytes) else str(v)


def droid_keep_mask(filter_dict: dict, traj: dict) -> np.ndarray:
    """droid_mixins.py:113-143 + droid_dataset.py:131-140: frame t of a DROID episode passes when the keep-ranges file lists
    `<recording_folderpath>--<file_path>` with a [start, end) range containing t; an episode the file does not list keeps nothing."""
    meta = traj["traj_metadata"]["episode_metadata"]
    key = decode_instruction(meta["recording_folderpath"]) + "--" + decode_instruction(meta["file_path"])
    T = len(np.asarray(traj["observation"]["cartesian_position"]))
    mask = np.zeros(T, dtype=bool)
    for start, end in filter_dict.get(key, []):
        mask[max(int(start), 0):max(min(int(end), T), 0)] = True
    return mask


def episode_from_rlds(dataset_name: str, traj: dict, *, hash_bucket=None, rng=None, keep_mask=None) -> dict | None:
    """One raw RLDS trajectory (dict of numpy arrays stacked over steps: `observation`, `action` [, `action_dict`],
    `language_instruction`) -> the episode-store dict of `lap_amd/data.py`, or None when the reference's filters drop it
    (empty instruction, zero length: oxe_datasets.py SingleOXEDataset "standard filtering").  `actions` are the per-step
    language actions [dx, dy, dz, droll, dpitch, dyaw, gripper] the label text is summed from; `target_actions` the reference's
    `action` field (what its `chunk_actions` turns into the training target); `state` keeps [xyz, euler, gripper].  `hash_bucket` / `rng`: see fill_empty_language_instruction (the Austin datasets only)."""
    if dataset_name not in STANDARDIZE:
        raise KeyError(f"no standardisation transform for {dataset_name!r} (built: {sorted(STANDARDIZE)})")
    kw = dict(hash_bucket=hash_bucket, rng=rng) if dataset_name in NEEDS_FALLBACK else {}
    std = STANDARDIZE[dataset_name](traj, **kw)
    prompt = decode_instruction(std["language_instruction"]) if "language_instruction" in std else ""
    if dataset_name == "droid" and not prompt.strip():
        # DROID carries three instructions per episode and the reference filters on the instruction TABLE, not on the first field
        # (droid_dataset.py:113-120,215-229): an episode whose first instruction is blank keeps training on its 2nd / 3rd
        prompt = next((a for a in (decode_instruction(std[k]) for k in ("language_instruction_2", "language_instruction_3") if k in std) if a.strip()), "")
    if not prompt.strip() or len(std["language_action"]) == 0:
        return None
    base_key, wrist_key = IMAGE_KEYS[dataset_name]
    obs = std["observation"]
    state = std["state"] if "state" in std else obs["state"]
    ep = {"base_0_rgb": np.asarray(obs[base_key], dtype=np.uint8), "state": np.asarray(state, dtype=np.float32),
          "actions": np.asarray(std["language_action"], dtype=np.float32), "prompt": prompt, "dataset_name": dataset_name,
          # what the reference CHUNKS into the flow-matching target (its `action`: absolute pose + gripper command, LIBERO: the raw
          # controller action), how, and the clock the label window / prediction look-ahead are measured on; the state is
          # [xyz, euler, gripper]: the loader converts it to [xyz, rot6d, gripper] as base_dataset.py:437-456 does
          "target_actions": np.asarray(std["action"], dtype=np.float32), "chunk_mode": chunk_mode_of(dataset_name), "state_encoding": "pos_euler"}
    if dataset_name in CONTROL_FREQUENCY:
        ep["control_frequency"] = np.int32(CONTROL_FREQUENCY[dataset_name])
    if dataset_name == "droid":      # droid_dataset.py:104-232
        # one of three instructions and one of two exterior cameras is drawn per episode: all of them travel; the loader draws
        alts = [decode_instruction(std[k]) for k in ("language_instruction", "language_instruction_2", "language_instruction_3") if k in std]
        alts = [a for a in alts if a.strip()]
        if len(alts) > 1:
            ep["prompt_alternatives"] = np.asarray(alts)
        if "exterior_image_2_left" in obs and np.asarray(obs["exterior_image_2_left"]).size:
            ep["base_0_rgb_alt"] = np.asarray(obs["exterior_image_2_left"], dtype=np.uint8)
        # only successful recordings with an instruction of more than 10 characters train (:215-229); the per-frame keep ranges of the
        # reference's idle filter come from a table outside the RLDS shards: `keep_mask` [T] when the caller has it
        meta = traj.get("traj_metadata", {}).get("episode_metadata", {}) if isinstance(traj.get("traj_metadata"), dict) else {}
        if "file_path" in meta and "success" not in decode_instruction(meta["file_path"]):
            return None
        if len(prompt) <= 10:
            return None
        if keep_mask is not None:
            ep["frame_mask"] = np.asarray(keep_mask, dtype=bool)
    if wrist_key and wrist_key in obs and np.asarray(obs[wrist_key]).size:
        ep["left_wrist_0_rgb"] = np.asarray(obs[wrist_key], dtype=np.uint8)
    return ep


def write_episode(path, ep: dict):
    np.savez_compressed(path, **ep)
