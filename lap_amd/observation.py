"""CoTObservation — the model-input record of the reference (src/lap/models/model_adapter.py:37-80 on top of
openpi.models.model.Observation), as a plain dataclass of torch tensors, plus preprocess_observation
(model_adapter.py:83-181) for the augmentation-off path used by `lap_libero` / the benchmark configs.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import torch


def _t(x, dtype=None, device=None):
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    elif not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if dtype is not None:
        x = x.to(dtype)
    if device is not None:
        x = x.to(device)
    return x


@dataclasses.dataclass
class CoTObservation:
    images: dict            # key -> f32 [B,H,W,3] in [-1,1]
    image_masks: dict       # key -> bool [B]
    state: torch.Tensor | None = None                 # f32 [B, action_dim]
    tokenized_prompt: torch.Tensor | None = None      # i32 [B, L]
    tokenized_prompt_mask: torch.Tensor | None = None  # bool [B, L]
    token_ar_mask: torch.Tensor | None = None
    token_loss_mask: torch.Tensor | None = None       # bool [B, L]
    tokenized_langact_mask: torch.Tensor | None = None  # bool [B, L]
    critical_token_mask: torch.Tensor | None = None
    number_token_mask: torch.Tensor | None = None
    direction_token_mask: torch.Tensor | None = None
    sample_mask: torch.Tensor | None = None           # bool [B]
    tokenized_dataset_name: torch.Tensor | None = None
    is_vqa_sample: torch.Tensor | None = None
    is_prediction_sample: torch.Tensor | None = None
    vqa_dataset_id: torch.Tensor | None = None
    # Engine hint, a HOST int (not a reference field): an upper bound on the number of loss-carrying token positions per sample
    # of this batch (langact & prompt & loss masks, positions 1..L-1).  With it the language head computes logits for that many
    # rows per sample instead of all L-1 (lap.py:221-260 computes them all and multiplies by the mask); set by `from_dict` and the
    # loaders from the host-side masks.  A bound that is too small poisons the loss with NaN (checked on the device, no sync).
    loss_rows_max: int | None = None

    @classmethod
    def from_dict(cls, data: dict, device=None) -> "CoTObservation":
        """openpi Observation.from_dict + CoT extras (model_adapter.py:51-80): accepts `image` / `image_mask`
        keys, uint8 images (converted to [-1, 1] floats) and flat or `extras.cot` CoT fields."""
        if ("tokenized_prompt" in data) != ("tokenized_prompt_mask" in data):
            raise ValueError("tokenized_prompt and tokenized_prompt_mask must be provided together.")
        images = {}
        for k, v in data["image"].items():
            v = _t(v, device=device)
            if v.dtype == torch.uint8:
                v = v.to(torch.float32) / 255.0 * 2.0 - 1.0
            images[k] = v.to(torch.float32)
        cot = data.get("extras", {}).get("cot", {}) if isinstance(data.get("extras"), dict) else {}
        g = lambda k: data.get(k, cot.get(k))
        b = lambda k: _t(g(k), torch.bool, device)
        hint = None
        la, pm, tl = g("tokenized_langact_mask"), data.get("tokenized_prompt_mask"), g("token_loss_mask")
        if la is not None and pm is not None and not (isinstance(la, torch.Tensor) and la.is_cuda):   # host-side masks: count here
            import numpy as np

            m = np.asarray(la, dtype=bool) & np.asarray(pm, dtype=bool)
            if tl is not None:
                m = m & np.asarray(tl, dtype=bool)
            hint = int(m[:, 1:].sum(-1).max()) if m.ndim == 2 and m.shape[1] > 1 else None
        return cls(
            loss_rows_max=hint,
            images=images,
            image_masks={k: _t(v, torch.bool, device) for k, v in data.get("image_mask", {}).items()},
            state=_t(data.get("state"), torch.float32, device),
            tokenized_prompt=_t(data.get("tokenized_prompt"), torch.int32, device),
            tokenized_prompt_mask=b("tokenized_prompt_mask"),
            token_ar_mask=_t(data.get("token_ar_mask"), None, device),
            token_loss_mask=b("token_loss_mask"),
            tokenized_langact_mask=b("tokenized_langact_mask"),
            critical_token_mask=b("critical_token_mask"),
            number_token_mask=b("number_token_mask"),
            direction_token_mask=b("direction_token_mask"),
            sample_mask=b("sample_mask"),
            tokenized_dataset_name=_t(g("tokenized_dataset_name"), None, device),
            is_vqa_sample=b("is_vqa_sample"),
            is_prediction_sample=b("is_prediction_sample"),
            vqa_dataset_id=_t(g("vqa_dataset_id"), None, device),
        )

    def to(self, device) -> "CoTObservation":
        mv = lambda x: x.to(device) if isinstance(x, torch.Tensor) else x
        d = {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}
        d["images"] = {k: mv(v) for k, v in self.images.items()}
        d["image_masks"] = {k: mv(v) for k, v in self.image_masks.items()}
        return CoTObservation(**{k: (v if k in ("images", "image_masks") else mv(v)) for k, v in d.items()})


def resize_with_pad(images: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """openpi.shared.image_tools.resize_with_pad ([UPSTREAM-RECALL], called at model_adapter.py:113-116): aspect-preserving
    bilinear resize (jax.image.resize semantics: half-pixel centres, anti-aliased when shrinking) of [b, h, w, c] images,
    then centred padding to (height, width) with the "black" value of the dtype's range (-1 for [-1, 1] floats, 0 for uint8).
    A request-time preprocessing step on a few hundred KB: torch ops, not a kernel of the hot path."""
    b, h, w, c = images.shape
    if (h, w) == (height, width):
        return images
    ratio = max(w / width, h / height)
    rh, rw = int(h / ratio), int(w / ratio)
    x = images.permute(0, 3, 1, 2).to(torch.float32)
    x = torch.nn.functional.interpolate(x, size=(rh, rw), mode="bilinear", align_corners=False, antialias=True)
    if images.dtype == torch.uint8:
        x = x.round().clamp(0, 255).to(torch.uint8)
        fill = 0
    else:
        x = x.clamp(-1.0, 1.0)
        fill = -1.0
    ph0, pw0 = (height - rh) // 2, (width - rw) // 2
    x = torch.nn.functional.pad(x, (pw0, width - rw - pw0, ph0, height - rh - ph0), value=fill)
    return x.permute(0, 2, 3, 1).contiguous()


def augmentation_params(batch: int, height: int, width: int, generator: torch.Generator, device, skip=None) -> torch.Tensor:
    """Random parameters of the train-time augmentation chain, f32 [batch, 12] in the layout of lap_augment_images:
    RandomCrop(95 %) offset uniform over the valid range, Rotate uniform in [-5, 5] degrees, ColorJitter brightness /
    contrast / saturation uniform in [-0.2, 0.2] (model_adapter.py:126-141)."""
    u = torch.rand((batch, 6), generator=generator, device=device, dtype=torch.float32)
    cw, ch = int(width * 0.95), int(height * 0.95)
    ang = (u[:, 2] * 10.0 - 5.0) * (math.pi / 180.0)
    par = torch.zeros((batch, 12), dtype=torch.float32, device=device)
    par[:, 0] = u[:, 0] * (width - cw)
    par[:, 1] = u[:, 1] * (height - ch)
    par[:, 2], par[:, 3] = float(cw), float(ch)
    par[:, 4], par[:, 5] = torch.cos(ang), torch.sin(ang)
    par[:, 6:9] = u[:, 3:6] * 0.4 - 0.2
    if skip is not None:
        par[:, 9] = skip.to(device=device, dtype=torch.float32)
    return par


def preprocess_observation(observation: CoTObservation, *, train: bool, image_keys, image_resolution,
                           enable_image_augmentation: bool = True, rng: torch.Generator | None = None) -> CoTObservation:
    """model_adapter.py:83-181: selects the image keys, resizes to the model resolution, applies the train-time augmentation
    chain (RandomCrop 95 % / Resize / Rotate +-5 deg / ColorJitter 0.2; augmax restated from memory, see lap_augment_images)
    and fills default image masks."""
    augment = train and enable_image_augmentation
    if augment and rng is None:
        raise ValueError("image augmentation needs a random generator (preprocess_observation(..., rng=...))")
    images, masks = {}, {}
    batch = None
    for key in image_keys:
        if key not in observation.images:
            raise ValueError(f"images dict missing key {key}; got {list(observation.images)}")
        img = observation.images[key]
        if tuple(img.shape[1:3]) != tuple(image_resolution):
            img = resize_with_pad(img, *image_resolution)
        if augment:   # one fused gather + colour kernel per camera (csrc/elementwise.hip); VQA samples pass unchanged
            from lap_amd import hip
            img = img.to(torch.float32).contiguous()
            par = augmentation_params(img.shape[0], img.shape[1], img.shape[2], rng, img.device, skip=observation.is_vqa_sample)
            img = hip.augment_images(img, par)
        images[key] = img
        batch = img.shape[0]
        m = observation.image_masks.get(key)
        masks[key] = m if m is not None else torch.ones(batch, dtype=torch.bool, device=img.device)
    return dataclasses.replace(observation, images=images, image_masks=masks)
