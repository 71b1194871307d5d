"""Offline VQA exporter: raw records of the reference's six vision-language datasets -> the samples the train loader mixes in.

The reference reads COCO captions, VQAv2, PixMo-Cap, PixMo-Points, LVIS and PACO through TFDS + dlimp
(`src/lap/datasets/vqa/*.py`, base class `vqa_base.py:53-265`): one frame per record, a prompt (drawn from a per-dataset table,
or the record's own question), a caption (the record's answer, a caption, a bounding box as PaliGemma `<locNNNN>` tokens, points
as loc tokens, or a direction word), a zero state, zero actions, `is_vqa_sample=True` and the dataset's id (for the per-dataset
loss weights / metrics of `compute_loss`, lap.py:401-413).  TensorFlow does not exist in the build image, so — like
`lap_amd/rlds_export.py` — the record -> sample arithmetic is restated here on plain Python / numpy values and the shard iteration
lives in `tools/export_vqa_samples.py`, which runs where TFDS exists and writes the `.npz` files `lap_amd.data.VqaDataset` reads.

Pinned by reference-generated fixtures (tests/golden/make_vqa_golden.py -> vqa_v1.json: every prompt table, 186 loc-token and
486 direction cases computed by the reference's own functions).  Two TensorFlow primitives are INJECTED by the caller because
they cannot be restated bit for bit: the FarmHash bucket of a string (`tf.strings.to_hash_bucket_fast`: trajectory ids, RNG
seeds) and the stateless uniform draws (`tf.random.stateless_uniform`: which prompt / caption).  Without them the stand-ins
below give the same distribution, not the same draws (`numpy_choices`)."""
from __future__ import annotations

import zlib

import numpy as np

# Registration order of the reference's VQA classes = their ids (registry.py:288-303 assigns 1, 2, ... as the modules of
# datasets/vqa/__init__.py:10-18 are imported; 0 = not a VQA sample)
VQA_DATASET_IDS = {"coco_captions": 1, "lvis": 2, "paco_lvis": 3, "paco_ego4d": 4, "pixmo_cap": 5, "pixmo_point": 6, "vqa": 7}
# `get_num_transitions` of the seven readers (coco_caption_dataset.py:52-54, lvis_dataset.py:41-43, paco_dataset.py:40-41,136-137,
# pixmo_cap_dataset.py:63-65, pixmo_point_dataset.py:100-102, vqav2_dataset.py:31): the SIZE a VQA set enters the mixture weights with —
# constants of the reference, not counts of what a store holds
VQA_NUM_TRANSITIONS = {"coco_captions": 80340, "lvis": 1231766, "paco_lvis": 612188, "paco_ego4d": 116356, "pixmo_cap": 620032,
                       "pixmo_point": 1702160, "vqa": 443800}
NUM_TRANSITIONS = {"coco_captions": 80340, "lvis": 1231766, "paco_lvis": 612188, "paco_ego4d": 116356, "pixmo_cap": 620032,
                   "pixmo_point": 1702160, "vqa": 443800}          # get_num_transitions() of each class (dummy statistics)
MAX_POINTS = 20                                                    # pixmo_point_dataset.py:10

COCO_CAPTION_PROMPTS = (
    'Caption the image.',
    'Give a short caption.',
    'Provide a brief description.',
    'What is shown?',
    'Summarize the image in a few words.',
    'Describe it concisely.',
    'One-sentence caption, please.',
    'Give a minimal caption.',
    "What's happening?",
    'A short description.',
    'Describe this briefly.',
    'Caption in one phrase.',
    'What is depicted?',
    'Label the image content.',
    'Provide a simple caption.',
    'In a few words, what is this?',
    'Write a concise caption.',
    'What does the picture show?',
    'Give a very short image description.',
    'Provide a compact caption.',
)     # coco_caption_dataset.py:9-34

PIXMO_CAP_PROMPTS = (
    'Describe this image.',
    'Describe this image',
    'describe the image',
    'Write a long description of this image.',
    'caption the picture',
    'Caption',
    'caption',
    'Construct a long caption for this image',
    'Generate a caption',
    'Create a detailed caption',
    'Write a long caption',
    'Describe this image in detail',
    'Describe this',
    'describe this',
    'Caption this',
    'What can be seen in this image?',
    'What do you see in the image?',
    'Look at this photo carefully and then tell me about it in detail',
    'Write a long description of this image',
    'Tell me about this picture.',
    'Write a paragraph about this image.',
    'Look at this image carefully and then describe it in detail',
    'Generate a long caption about this image.',
    'Describe this image in detail, but without any pointing.',
    'Write a long description of this image, do not produce any points.',
    'Tell me about this picture, use plain text only.',
    'Generate a plain text description of this caption',
    'What is in this image?\nNo pointing\nGive lots of detailWrite a long caption.\nDo not use image coordinates\nOutput a full paragraph',
)     # pixmo_cap_dataset.py:10-43 (28 entries: the reference's list lacks one comma, two prompts are one string)

PIXMO_POINT_PROMPT_PARTS = (
    ('How many ', ' are in the image? Point them out.'),
    ('Point out all the ', ' in this image.'),
    ('Where are the ', ' in the image? Point to each one.'),
    ('Locate all ', ' in the image and point them out.'),
    ('Point to ', ". Please say 'There are none.' if it is not in the image."),
    ('Point to all occurrences of ', '.'),
    ('Point to any ', ' in the image.'),
    ('Point: Where are the ', '?'),
    ('Show me where the ', ' are.'),
    ('If there are any ', ' in the image, show me where they are.'),
    ('Where are the ', '?'),
    ('Generate a list of points showing where the ', ' are.'),
    ('Find the ', '.'),
    ('Locate all ', '.'),
    ('Locate the ', '.'),
    ('Object: ', '. Instruction: Point to the object.'),
    ('find ', '.'),
    ('Point to every ', '.'),
    ('Find any ', '.'),
    ('Point to a ', '.'),
    ('Look for ', ' in the image and show me where they are.'),
    ('Help me find an object in the image by pointing to it. Object: ', '.'),
    ('I am looking for ', ', where can it be found in the image?'),
    ('Can you see any ', ' in the image? Point to them.'),
    ('Point out each ', ' in the image.'),
    ('Show me where the robot should move its end-effector to reach the ', ' in the image.'),
    ('Point to where the robot should position its gripper to grasp the ', '.'),
    ('Locate the point where the robot should align its end-effector with the ', ' in the image.'),
    ('Mark the location the robot should target with its gripper to reach the ', '.'),
    ('Identify the spot the robot should move its arm toward to approach the ', '.'),
    ('Point to the region the robot should aim its end-effector at to interact with the ', '.'),
    ('Show me the point where the robot would position its gripper to approach the ', ' in the image.'),
    ('Indicate where the robot should move its arm to reach the ', '.'),
    ('Point to the location the robot should target to interact with the ', '.'),
    ('Highlight the point the robot should move toward to grasp the ', '.'),
    ('Identify where the robot should position its wrist relative to the ', '.'),
    ('Point out the spot the robot would navigate its arm to in order to reach the ', '.'),
    ('Locate where the robot would need to move its end-effector to get closer to the ', ' in the image.'),
    ('Point to the position the robot should move its gripper toward to access the ', '.'),
    ('Show the point the robot should aim its arm toward when approaching the ', '.'),
    ('Indicate the exact point a robot should target with its gripper when reaching for the ', '.'),
    ('Point to where the robot should aim its wrist to reach the ', '.'),
    ('Mark the precise point where the robot should position its end-effector to approach the ', '.'),
    ('Identify the point where the robot would place its gripper to interact with the ', '.'),
    ('Show the location the robot should move its arm to reach the ', '.'),
    ('Locate the target point the robot should align its manipulator with to access the ', '.'),
    ('Point out the position the robot would need to occupy with its wrist to manipulate the ', '.'),
    ("Point to the region that represents the robot's goal location for reaching the ", '.'),
    ('Find the point in the image that the robot should move its end-effector toward to reach the ', '.'),
    ('Mark the destination point a robot should target with its gripper to successfully approach the ', '.'),
)     # pixmo_point_dataset.py:14-69: (prefix, suffix) around the label

GENERAL_BBOX_PROMPT_PARTS = (
    ('Show me where the robot should move its end-effector to reach the ', ' in the image.'),
    ('Describe the location the robot should align its gripper with to reach the ', ' in the image.'),
    ('Locate the region where the robot should position its wrist to interact with the ', ' in the image.'),
    ('Mark the location the robot should target with its gripper to reach the ', '.'),
    ('Identify the spot the robot should move its arm toward to approach the ', '.'),
    ('Find the region the robot should align its end-effector with to reach the ', ' in the image.'),
    ('Highlight the area the robot should approach with its manipulator to reach the ', ' in the image.'),
    ('Show me where the robot would position its gripper to approach the ', ' in the image.'),
    ('Indicate where the robot should move its arm to reach the ', '.'),
    ('Mark the location the robot should target to interact with the ', '.'),
    ('Highlight the region the robot should move toward to grasp the ', '.'),
    ('Identify where the robot should position its wrist relative to the ', '.'),
    ('Point out the spot the robot would navigate its arm to in order to reach the ', '.'),
    ('Locate where the robot would need to move its end-effector to get closer to the ', ' in the image.'),
    ('Pinpoint the position the robot should move its gripper toward to access the ', '.'),
    ('Show the area the robot should aim its arm toward when approaching the ', '.'),
    ("Outline the region that would guide the robot's end-effector toward the ", '.'),
    ('Indicate the exact region a robot should target with its gripper when reaching for the ', '.'),
    ('Highlight the bounding region the robot should aim its wrist toward to reach the ', '.'),
    ('Mark the precise location where the robot should position its end-effector to approach the ', '.'),
    ('Identify the spatial region where the robot would place its gripper to interact with the ', '.'),
    ('Show the area the robot should move its arm into to reach the ', '.'),
    ('Locate the target region the robot should align its manipulator with to access the ', '.'),
    ('Point out the position the robot would need to occupy with its wrist to manipulate the ', '.'),
    ("Outline the region that represents the robot's goal location for reaching the ", '.'),
    ('Find the area in the image that the robot should move its end-effector toward to reach the ', '.'),
    ('Mark the destination region a robot should select with its gripper to successfully approach the ', '.'),
)     # bbox/prompts.py:13-43

DIRECTION_PROMPT_PARTS = (
    ('From the image center, imagine the robot moving its end-effector toward the ', ' and predict the direction.'),
    ('Relative to the center of the image, imagine the robot aligning its arm toward the ', ' and describe the movement direction.'),
    ("If the robot's base were at the center of the image, which way would the arm extend to reach the ", '.'),
    ('Looking from the center of the frame, imagine the robot orienting its gripper toward the ', ' and state the direction.'),
    ('Which direction from the center would the robot move its end-effector to reach the ', ' in this image.'),
    ('Imagine the robot must reposition its arm to interact with the ', ' and describe its direction.'),
    ('Describe which direction the robot would move its gripper to approach the ', ' in the image.'),
    ("Describe the direction the robot's arm should sweep to align with the ", ' in the image.'),
    ('Point out the direction the robot should move its end-effector to reach the ', '.'),
    ('Show me where the robot should aim its arm to reach the ', '.'),
    ('Describe where the robot would move its wrist to reach the ', ' relative to the center of the image.'),
    ('Show me the direction the robot should move its arm toward the ', ' relative to the center of the image.'),
    ('Imagine the robot needs to extend its arm toward the ', ' and predict the direction.'),
    ('Imagine the robot needs to reposition its manipulator to the ', ' and predict the direction.'),
    ('If the robot needs to grasp the ', ', predict the direction it would move its arm.'),
    ('From the image center, predict the direction the robot should move its end-effector to make contact with the ', '.'),
    ('Assuming the robot starts with its gripper at the image center, describe the direction it should move toward the ', '.'),
    ('If the robot had to plan a straight-line reach from the center to the ', ', which direction would the arm move.'),
    ('Imagine the robot is positioned at the center and must align its gripper with the ', '; indicate the direction.'),
    ('From the center of the image, in which direction should the robot move its wrist to approach the ', '.'),
    ('If the robot were planning a pre-grasp motion from the center, describe the direction toward the ', '.'),
    ('Predict the initial arm movement direction a robot would take from the center to reach the ', '.'),
    ('Considering a robot at the center, which direction would it orient its gripper to approach the ', '.'),
    ('From a manipulation standpoint, which direction should the robot move its arm from the center to reach the ', '.'),
    ('If the robot plans a direct reach from the center to the ', ', what direction would the end-effector move.'),
)     # bbox/prompts.py:103-131

# bbox/prompts.py:49-101: "<verb> the <object><where>" combinations in front of the general prompts
_ROBOT_BBOX_PART1 = ("Pick up the ", "Grasp the ", "Move near to the ", "Navigate to the ")
_ROBOT_BBOX_PART2_IMAGE = (", predict where it is in the image.", ", show where it is in the image.", ", locate it in the image.", ", find it in the image.")
_ROBOT_BBOX_PART2_ROBOT_BASE = (", predict where it is in the robot base frame.", ", relative to the robot base.", ", with respect to the robot base.",
                                ", looking from the external camera.")
_ROBOT_BBOX_PART2_EE = (", predict where it is in the end-effector frame.", ", with respect to the robot gripper.", ", relative to the end-effector.",
                        ", in the wrist camera.", ", looking from the wrist camera.")
ROBOT_BBOX_PROMPT_PARTS = tuple((a, b) for a in _ROBOT_BBOX_PART1 for b in _ROBOT_BBOX_PART2_IMAGE + _ROBOT_BBOX_PART2_ROBOT_BASE + _ROBOT_BBOX_PART2_EE) + GENERAL_BBOX_PROMPT_PARTS
ROBOT_BBOX_PROMPT_PARTS_OXE = tuple((a, b) for a in _ROBOT_BBOX_PART1 for b in _ROBOT_BBOX_PART2_IMAGE + _ROBOT_BBOX_PART2_ROBOT_BASE) + GENERAL_BBOX_PROMPT_PARTS
ROBOT_BBOX_PROMPT_PARTS_EE = tuple((a, b) for a in _ROBOT_BBOX_PART1 for b in _ROBOT_BBOX_PART2_IMAGE + _ROBOT_BBOX_PART2_EE) + GENERAL_BBOX_PROMPT_PARTS
# bbox/prompts.py:137-173
_ROBOT_DIRECTION_PART1 = ("Pick up the ", "Move to the ", "Grab the ", "Navigate to the ")
_ROBOT_DIRECTION_PART2_EE = (", predict the robot's action in the end-effector frame.", ", with respect to the robot gripper.", ", relative to the end-effector.",
                             ", in the wrist camera.", ", looking from the wrist camera.")
_ROBOT_DIRECTION_PART2_ROBOT_BASE = (", predict the robot's action in the robot base frame.", ", relative to the robot base.", ", with respect to the robot base.",
                                     ", in the robot base coordinate frame.", ", in the robot base frame.", ", looking from the external camera.")
ROBOT_DIRECTION_PROMPT_PARTS_OXE = tuple((a, b) for a in _ROBOT_DIRECTION_PART1 for b in _ROBOT_DIRECTION_PART2_ROBOT_BASE) + DIRECTION_PROMPT_PARTS
ROBOT_DIRECTION_PROMPT_PARTS_EE = tuple((a, b) for a in _ROBOT_DIRECTION_PART1 for b in _ROBOT_DIRECTION_PART2_EE) + DIRECTION_PROMPT_PARTS


# ------------------------------------------------------------------------------ geometry -> text
def _loc(v: float, bins: int) -> int:
    return int(np.round(np.float32(v) * np.float32(bins - 1))) if isinstance(v, np.floating) else int(round(v * (bins - 1)))


def bbox_to_text(x_min: float, y_min: float, x_max: float, y_max: float, num_bins: int = 1024) -> str:
    """bbox/coord_utils.py:10-89: normalised corners -> "<locYMIN><locXMIN><locYMAX><locXMAX>" (round half to even like Python's
    round / tf.round)."""
    return "".join(f"<loc{_loc(v, num_bins):04d}>" for v in (y_min, x_min, y_max, x_max))


def points_to_text(points: np.ndarray) -> str:
    """pixmo_point_dataset.py:123-160: points [N, 2] = (x, y) on a 0-100 scale, sorted by x * 10000 + y, as "<locYYYY><locXXXX>" each."""
    p = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    order = np.argsort(p[:, 0] * np.float32(10000.0) + p[:, 1], kind="stable")
    p = p[order]
    idx = np.round(p / np.float32(100.0) * np.float32(1023)).astype(np.int32)
    return "".join(f"<loc{int(y):04d}><loc{int(x):04d}>" for x, y in idx)


def direction_from_bbox(x_min: float, y_min: float, x_max: float, y_max: float, slope: float = 2.0, add_move_prefix: bool = False) -> str:
    """bbox/direction.py:10-78 (and its Python twin :134-181): where the box centre lies relative to the image centre — sectors bounded
    by lines of slope k and 1 / k: forward / back / left / right, else "left and forward" ..."""
    x_rel = (x_min + x_max) / 2.0 - 0.5            # + is right
    y_rel = 0.5 - (y_min + y_max) / 2.0            # + is up = forward
    k, ax, ay = slope, abs(x_rel), abs(y_rel)
    if y_rel > k * ax:
        d = "forward"
    elif y_rel < -k * ax:
        d = "back"
    elif x_rel > ay / k:
        d = "right"
    elif x_rel < -ay / k:
        d = "left"
    else:
        d = ("left" if x_rel < 0.0 else "right") + " and " + ("forward" if y_rel >= 0.0 else "back")
    return "move " + d if add_move_prefix else d


# ------------------------------------------------------------------------------ the injected TensorFlow primitives
def crc_hash_bucket(text: str, n: int) -> int:
    """STAND-IN for tf.strings.to_hash_bucket_fast (FarmHash Fingerprint64 % n): a deterministic bucket of the UTF-8 bytes.  Same use
    (stable ids and seeds), different values; tools/export_vqa_samples.py passes the real one."""
    return zlib.crc32(text.encode("utf-8")) % n


def numpy_choices(seed_pair, n: int) -> int:
    """STAND-IN for tf.random.stateless_uniform([], seed=seed_pair, minval=0, maxval=n, dtype=int32): a draw that is a pure function
    of the seed pair (numpy Philox keyed with it) — same distribution, not TensorFlow's bits."""
    a, b = (int(v) & 0xFFFFFFFF for v in seed_pair)
    return int(np.random.Generator(np.random.Philox(key=(a << 32) | b)).integers(n))


def numpy_uniform(seed_pair) -> float:
    a, b = (int(v) & 0xFFFFFFFF for v in seed_pair)
    return float(np.random.Generator(np.random.Philox(key=(a << 32) | b)).random(dtype=np.float32))


def _s(x) -> str:
    if isinstance(x, (bytes, np.bytes_)):
        return x.decode("utf-8")
    a = np.asarray(x)
    if a.dtype.kind in "SUO" and a.ndim == 0:
        v = a.item()
        return v.decode("utf-8") if isinstance(v, bytes) else str(v)
    return str(x)


def _fmt_float(v) -> str:
    """tf.strings.as_string of a float32 scalar with default arguments prints like "%g"-style shortest ("0.25", "1e-05")."""
    return "%g" % float(np.float32(v))


# ------------------------------------------------------------------------------ per-dataset record -> (id, prompt, caption)
def trajectory_id(name: str, rec: dict, hash_bucket=crc_hash_bucket) -> str:
    """The `create_trajectory_id` of each class: what the train / validation split hashes (vqa_base.py:203-214)."""
    M = 2147483647
    if name == "coco_captions":        # coco_caption_dataset.py:57-61
        return f"coco_{_s(rec['image/filename'])}_{int(rec['image/id'])}"
    if name == "vqa":                  # vqav2_dataset.py:35-39
        return f"vqa_{int(rec['question_id'])}_{int(rec['image/id'])}"
    if name == "pixmo_cap":            # pixmo_cap_dataset.py:65-76
        fn = _s(rec["image_filename"])
        return f"pixmo_cap_{fn}_{hash_bucket(fn + '_' + _s(rec['caption']), M)}"
    if name == "pixmo_point":          # pixmo_point_dataset.py:106-121
        sha = _s(rec["image_sha256"])
        return f"pixmo_point_{sha}_{hash_bucket(sha + '_' + _s(rec['label']) + '_' + str(int(rec['count'])), M)}"
    if name in ("lvis", "paco_lvis", "paco_ego4d"):     # lvis_dataset.py:51-69, paco_dataset.py:44-62
        iid, cat = _s(rec["image_id"]), _s(np.asarray(rec["annotations"]["category_name"]).reshape(-1)[0])
        bbox = np.asarray(rec["annotations"]["bbox"], dtype=np.float32).reshape(-1, 2, 2)[0]
        bbox_str = "_".join(_fmt_float(v) for v in bbox.reshape(-1))
        prefix = "lvis_" if name == "lvis" else "paco_"
        return f"{prefix}{iid}_{hash_bucket(iid + '_' + cat + '_' + bbox_str, M)}"
    raise KeyError(name)


def is_validation(name: str, rec: dict, split_seed: int, val_fraction: float, hash_bucket=crc_hash_bucket) -> bool:
    """vqa_base.py:203-214: bucket(str(seed) + trajectory id) of 1000 below int(val_fraction * 1000)."""
    return hash_bucket(str(int(split_seed)) + trajectory_id(name, rec, hash_bucket), 1000) < int(val_fraction * 1000)


def prompt_and_caption(name: str, rec: dict, seed: int = 0, *, directional: bool = False, direction_prob: float = 0.0, direction_slope: float = 2.0,
                       scale: float = 1.0, max_points: int = MAX_POINTS, hash_bucket=crc_hash_bucket, choose=numpy_choices, uniform=numpy_uniform):
    """`extract_prompt_and_caption` of each class.  `seed`: the dataset object's seed; the per-record part of every RNG seed is the
    record's own id (COCO) or the FarmHash of its id string, as in the reference."""
    M = 2147483647
    if name == "vqa":                  # vqav2_dataset.py:41-51
        return _s(rec["question_text"]), _s(rec["top_answer"])
    if name == "coco_captions":        # coco_caption_dataset.py:63-88: one of the image's captions, one of 20 prompts
        h = int(rec["image/id"]) % M
        caps = [_s(c) for c in np.asarray(rec["captions"]["text"]).reshape(-1)]
        cap = caps[choose((seed, h), len(caps))]
        return COCO_CAPTION_PROMPTS[choose((seed + 1, h), len(COCO_CAPTION_PROMPTS))], cap
    if name == "pixmo_cap":            # pixmo_cap_dataset.py:78-100
        h = hash_bucket(_s(rec["image_filename"]), M)
        return PIXMO_CAP_PROMPTS[choose((seed, h), len(PIXMO_CAP_PROMPTS))], _s(rec["caption"])
    if name == "pixmo_point":          # pixmo_point_dataset.py:162-207
        pts = np.stack([np.asarray(rec["points"]["x"], dtype=np.float32).reshape(-1), np.asarray(rec["points"]["y"], dtype=np.float32).reshape(-1)], 1)
        pts = np.round(pts * np.float32(100.0 / scale) * np.float32(10.0)) / np.float32(10.0)
        pts = pts[:max_points]
        h = hash_bucket(_s(rec["image_sha256"]), M)
        pre, suf = PIXMO_POINT_PROMPT_PARTS[choose((seed, h), len(PIXMO_POINT_PROMPT_PARTS))]
        return pre + _s(rec["label"]) + suf, points_to_text(pts)
    if name in ("lvis", "paco_lvis", "paco_ego4d"):     # lvis_dataset.py:71-118, paco_dataset.py:64-111
        h = hash_bucket(_s(rec["image_id"]), M)
        cat = _s(np.asarray(rec["annotations"]["category_name"]).reshape(-1)[0])
        (x0, y0), (x1, y1) = np.asarray(rec["annotations"]["bbox"], dtype=np.float32).reshape(-1, 2, 2)[0]
        move = direction_from_bbox(float(x0), float(y0), float(x1), float(y1), slope=direction_slope, add_move_prefix=True)
        if directional:
            pre, suf = DIRECTION_PROMPT_PARTS[choose((seed, h), len(DIRECTION_PROMPT_PARTS))]
            return pre + cat + suf, move
        pre, suf = ROBOT_BBOX_PROMPT_PARTS_OXE[choose((seed, h), len(ROBOT_BBOX_PROMPT_PARTS_OXE))]
        use_direction = uniform((seed + 7919, h)) < direction_prob
        return pre + cat + suf, (move if use_direction else bbox_to_text(x0, y0, x1, y1))
    raise KeyError(name)


def sample_from_record(name: str, rec: dict, seed: int = 0, **kw) -> dict | None:
    """One raw record -> the `.npz` fields of `lap_amd.data.VqaDataset`, or None when the reference's frame filter drops it (empty
    question or answer, vqa_base.py:257-265).  The image stays what the record holds (HWC uint8, or encoded bytes)."""
    prompt, caption = prompt_and_caption(name, rec, seed, **kw)
    if not prompt or not caption:
        return None
    img = rec["image"]
    return {"image": np.asarray(img) if not isinstance(img, (bytes, np.bytes_)) else np.frombuffer(img, dtype=np.uint8),
            "image_encoded": isinstance(img, (bytes, np.bytes_)), "prompt": prompt, "caption": caption, "dataset_name": name,
            "vqa_dataset_id": VQA_DATASET_IDS[name]}
