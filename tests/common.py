"""Shared builders for parity tests: the debug-size LAP config in engine and oracle form, seeded inputs."""
import dataclasses

import torch

from lap_amd.config import VQA_DATASET_ID_MAP, get_config
from oracle import lap_oracle as O


def oracle_cfg(model_cfg, **kw) -> O.OracleCfg:
    return O.OracleCfg(paligemma_variant=model_cfg.paligemma_variant, action_expert_variant=model_cfg.action_expert_variant,
                       siglip_variant=model_cfg.siglip_variant, action_dim=model_cfg.action_dim,
                       action_horizon=model_cfg.action_horizon, max_token_len=model_cfg.max_token_len,
                       image_size=model_cfg.image_size, image_keys=model_cfg.image_keys, vocab_size=model_cfg.vocab_size,
                       language_loss_weight=model_cfg.language_loss_weight, action_loss_weight=model_cfg.action_loss_weight,
                       stop_action_to_vlm_grad=model_cfg.stop_action_to_vlm_grad, pi05=model_cfg.pi05,
                       enable_action_training=model_cfg.enable_action_training, enable_langact_training=model_cfg.enable_langact_training,
                       enable_vqa_training=model_cfg.enable_vqa_training, enable_prediction_training=model_cfg.enable_prediction_training,
                       vqa_loss_weight=model_cfg.vqa_loss_weight, prediction_loss_weight=model_cfg.prediction_loss_weight,
                       vqa_loss_weights_by_id=tuple((VQA_DATASET_ID_MAP[k], v) for k, v in (model_cfg.vqa_loss_weights or {}).items()
                                                    if k in VQA_DATASET_ID_MAP), **kw)


def debug_model_cfg(**kw):
    return dataclasses.replace(get_config("debug").model, **kw)


def make_inputs(cfg, B=2, seed=1, ragged=True):
    """Seeded synthetic batch in the reference's input convention incl. padding, an invalid image,
    per-sample langact spans and an idle sample."""
    g = torch.Generator().manual_seed(seed)
    L, H = cfg.max_token_len, cfg.image_size
    images = {k: torch.rand(B, H, H, 3, generator=g) * 2 - 1 for k in cfg.image_keys}
    image_masks = {k: torch.ones(B, dtype=torch.bool) for k in cfg.image_keys}
    pm = torch.ones(B, L, dtype=torch.bool)
    la = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        npad = (3 * b) % 5 if ragged else 0
        nl = 8 if not ragged else 6 + (b % 3)
        pm[b, L - npad:] = False
        la[b, L - npad - nl:L - npad] = True
    if ragged and B > 1:
        image_masks[cfg.image_keys[1]][B - 1] = False
    obs = dict(images=images, image_masks=image_masks,
               tokenized_prompt=torch.randint(0, cfg.vocab_size, (B, L), generator=g),
               tokenized_prompt_mask=pm, tokenized_langact_mask=la, token_loss_mask=torch.ones(B, L, dtype=torch.bool),
               sample_mask=torch.ones(B, dtype=torch.bool), state=torch.rand(B, cfg.action_dim, generator=g) * 2 - 1)
    if ragged and B > 2:
        obs["sample_mask"][1] = False
    actions = torch.randn(B, cfg.action_horizon, cfg.action_dim, generator=g)
    noise = torch.randn(B, cfg.action_horizon, cfg.action_dim, generator=g)
    time = torch.rand(B, generator=g) * 0.999 + 0.001
    return obs, actions, noise, time


def to_observation(obs, device):
    from lap_amd.observation import CoTObservation

    return CoTObservation(images={k: v.to(device) for k, v in obs["images"].items()},
                          image_masks={k: v.to(device) for k, v in obs["image_masks"].items()},
                          state=obs["state"].to(device), tokenized_prompt=obs["tokenized_prompt"].to(torch.int32).to(device),
                          tokenized_prompt_mask=obs["tokenized_prompt_mask"].to(device),
                          tokenized_langact_mask=obs["tokenized_langact_mask"].to(device) if obs.get("tokenized_langact_mask") is not None else None,
                          token_loss_mask=obs["token_loss_mask"].to(device), sample_mask=obs["sample_mask"].to(device),
                          is_vqa_sample=obs["is_vqa_sample"].to(device) if obs.get("is_vqa_sample") is not None else None,
                          is_prediction_sample=obs["is_prediction_sample"].to(device) if obs.get("is_prediction_sample") is not None else None,
                          vqa_dataset_id=obs["vqa_dataset_id"].to(device) if obs.get("vqa_dataset_id") is not None else None)


def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def tiny_sentencepiece_proto(vocab_size=150) -> bytes:
    """A throw-away SentencePiece model with PaliGemma's special ids (pad 0, eos 1, bos 2, unk 3) for the tokenizer tests:
    the real paligemma_tokenizer.model cannot be fetched offline."""
    import io

    import sentencepiece as spm
    corpus = ["Task: pick up the block, predict the robot's action in the robot base frame; State: 12 200 7 0 255 128; Answer: ",
              "move right 3 cm and move up 2 cm and rotate clockwise 10 degrees and open gripper",
              "move left 5 cm move forward 1 cm move back 4 cm close gripper end-effector frame"] * 40
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=buf, vocab_size=vocab_size, model_type="bpe",
                                   user_defined_symbols=["right", "left"], pad_id=0, eos_id=1, bos_id=2, unk_id=3,
                                   character_coverage=1.0, minloglevel=2)
    return buf.getvalue()
