"""Independent cross-check of the CPU oracle (oracle/lap_oracle.py) against HuggingFace
transformers' Gemma decoder and SigLIP vision encoder with shared random weights (SURVEY §8c:
the JAX reference cannot run here, these same-family implementations can), plus the analytic
invariants of the mask builder.  CPU only."""
import dataclasses
import math

import pytest
import torch

from oracle import lap_oracle as O

CFG = O.OracleCfg(paligemma_variant="dummy", action_expert_variant="dummy", siglip_variant="mu/14", action_horizon=10,
                  max_token_len=24, image_size=56, vocab_size=512)


def test_gemma_block_matches_hf():
    tr = pytest.importorskip("transformers")
    from transformers.models.gemma.modeling_gemma import GemmaConfig, GemmaModel

    c = CFG.vlm
    hc = GemmaConfig(vocab_size=CFG.vocab_size, hidden_size=c.width, intermediate_size=c.mlp_dim, num_hidden_layers=c.depth,
                     num_attention_heads=c.num_heads, num_key_value_heads=c.num_kv_heads, head_dim=c.head_dim,
                     hidden_act="gelu_pytorch_tanh", hidden_activation="gelu_pytorch_tanh", rms_norm_eps=1e-6,
                     rope_theta=10000.0, attention_bias=False, attention_dropout=0.0, max_position_embeddings=1024)
    hc._attn_implementation = "eager"
    model = GemmaModel(hc).eval().float()
    P = O.init_params(CFG, seed=3)
    lay = "PaliGemma/llm/layers"
    with torch.no_grad():
        for l, hl in enumerate(model.layers):
            # SURVEY §8(a-bis) weight map: q_einsum.w [N,D,H] <-> q_proj.weight [(N H), D]
            hl.self_attn.q_proj.weight.copy_(P[f"{lay}/attn/q_einsum/w"][l].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.k_proj.weight.copy_(P[f"{lay}/attn/kv_einsum/w"][l][0].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.v_proj.weight.copy_(P[f"{lay}/attn/kv_einsum/w"][l][1].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.o_proj.weight.copy_(P[f"{lay}/attn/attn_vec_einsum/w"][l].reshape(-1, c.width).t())
            hl.mlp.gate_proj.weight.copy_(P[f"{lay}/mlp/gating_einsum"][l][0].t())
            hl.mlp.up_proj.weight.copy_(P[f"{lay}/mlp/gating_einsum"][l][1].t())
            hl.mlp.down_proj.weight.copy_(P[f"{lay}/mlp/linear"][l].t())
            hl.input_layernorm.weight.copy_(P[f"{lay}/pre_attention_norm/scale"][l])
            hl.post_attention_layernorm.weight.copy_(P[f"{lay}/pre_ffw_norm/scale"][l])
        model.norm.weight.copy_(P["PaliGemma/llm/final_norm/scale"])
    B, T = 2, 11
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, c.width, generator=g)
    pos = torch.stack([torch.arange(T), torch.arange(T) + 5])
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))[None].expand(B, -1, -1)
    (ours, _), _ = O.gemma_forward(P, CFG, [x, None], pos, causal, [None, None])
    add = torch.zeros(B, 1, T, T).masked_fill(~causal[:, None], torch.finfo(torch.float32).min)
    # run the decoder stack by hand (version-independent): embeddings are fed as-is
    with torch.no_grad():
        h = x
        pe = model.rotary_emb(h, pos)
        for hl in model.layers:
            o = hl(h, attention_mask=add, position_ids=pos, position_embeddings=pe)
            h = o[0] if isinstance(o, tuple) else o
        h = model.norm(h)
    assert torch.allclose(ours, h, atol=2e-4, rtol=2e-4), (ours - h).abs().max()


def test_siglip_matches_hf():
    pytest.importorskip("transformers")
    from transformers.models.siglip.modeling_siglip import SiglipVisionConfig, SiglipVisionModel

    s = CFG.img
    hc = SiglipVisionConfig(hidden_size=s.width, intermediate_size=s.mlp_dim, num_hidden_layers=s.depth,
                            num_attention_heads=s.num_heads, image_size=CFG.image_size, patch_size=s.patch,
                            layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh", attention_dropout=0.0)
    hc._attn_implementation = "eager"
    model = SiglipVisionModel(hc).eval().float()
    vm = model.vision_model if hasattr(model, "vision_model") else model
    P = O.init_params(CFG, seed=4)
    blk = "PaliGemma/img/Transformer/encoderblock"
    hd = s.width // s.num_heads
    with torch.no_grad():
        # Flax conv kernel [P,P,C,W] <-> torch [W,C,P,P]
        vm.embeddings.patch_embedding.weight.copy_(P["PaliGemma/img/embedding/kernel"].permute(3, 2, 0, 1))
        vm.embeddings.patch_embedding.bias.copy_(P["PaliGemma/img/embedding/bias"])
        vm.embeddings.position_embedding.weight.copy_(P["PaliGemma/img/pos_embedding"][0])
        for l, hl in enumerate(vm.encoder.layers):
            hl.layer_norm1.weight.copy_(P[f"{blk}/LayerNorm_0/scale"][l]); hl.layer_norm1.bias.copy_(P[f"{blk}/LayerNorm_0/bias"][l])
            hl.layer_norm2.weight.copy_(P[f"{blk}/LayerNorm_1/scale"][l]); hl.layer_norm2.bias.copy_(P[f"{blk}/LayerNorm_1/bias"][l])
            for nm, proj in (("query", hl.self_attn.q_proj), ("key", hl.self_attn.k_proj), ("value", hl.self_attn.v_proj)):
                proj.weight.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/kernel"][l].reshape(s.width, -1).t())
                proj.bias.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/bias"][l].reshape(-1))
            hl.self_attn.out_proj.weight.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/out/kernel"][l].reshape(-1, s.width).t())
            hl.self_attn.out_proj.bias.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/out/bias"][l])
            hl.mlp.fc1.weight.copy_(P[f"{blk}/MlpBlock_0/Dense_0/kernel"][l].t()); hl.mlp.fc1.bias.copy_(P[f"{blk}/MlpBlock_0/Dense_0/bias"][l])
            hl.mlp.fc2.weight.copy_(P[f"{blk}/MlpBlock_0/Dense_1/kernel"][l].t()); hl.mlp.fc2.bias.copy_(P[f"{blk}/MlpBlock_0/Dense_1/bias"][l])
        vm.post_layernorm.weight.copy_(P["PaliGemma/img/Transformer/encoder_norm/scale"])
        vm.post_layernorm.bias.copy_(P["PaliGemma/img/Transformer/encoder_norm/bias"])
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, CFG.image_size, CFG.image_size, 3, generator=g) * 2 - 1
    col = {}
    O.siglip_forward(P, CFG, img, col)
    with torch.no_grad():
        ref = model(pixel_values=img.permute(0, 3, 1, 2)).last_hidden_state
    assert torch.allclose(col["img/encoded"], ref, atol=2e-4, rtol=2e-4), (col["img/encoded"] - ref).abs().max()


def test_make_attn_mask_truth_table():
    # I/Q bidirectional, L causal, pad matches nothing, A sees I/Q + A but not L.
    n_iq, n_l, n_pad, S = 5, 4, 2, 3
    Pn = n_iq + n_l + n_pad
    prefix_mask = torch.tensor([[True] * (n_iq + n_l) + [False] * n_pad])
    langact = torch.tensor([[False] * n_iq + [True] * n_l + [False] * n_pad])
    obs = {"tokenized_langact_mask": langact[:, 2:]}  # pretend the first 2 prefix tokens are image tokens
    suffix_mask = torch.ones(1, S, dtype=torch.bool)
    suffix_ar = torch.tensor([[True] + [False] * (S - 1)])
    pma, m, pos = O.build_masks_positions(CFG, obs, prefix_mask, langact, suffix_mask, suffix_ar)
    m = m[0]
    iq, l, pad, a = slice(0, n_iq), slice(n_iq, n_iq + n_l), slice(n_iq + n_l, Pn), slice(Pn, Pn + S)
    assert m[iq, iq].all() and not m[iq, l].any() and not m[iq, pad].any() and not m[iq, a].any()
    assert m[l, iq].all() and torch.equal(m[l, l], torch.tril(torch.ones(n_l, n_l, dtype=torch.bool)))
    assert not m[l, a].any() and not m[pad].any()
    assert m[a, iq].all() and not m[a, l].any() and not m[a, pad].any() and m[a, a].all()
    # positions (lap.py:366-377): action tokens restart right after the prompt
    assert pos[0, :n_iq + n_l].tolist() == list(range(n_iq + n_l))
    assert pos[0, Pn:].tolist() == [n_iq + i for i in range(S)]
    # every valid row's softmax mask is non-empty
    assert m[torch.cat([torch.arange(0, n_iq + n_l), torch.arange(Pn, Pn + S)])].any(-1).all()


def test_flow_matching_identities():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(3, 10, 7, generator=g); e = torch.randn(3, 10, 7, generator=g); t = torch.rand(3, generator=g)
    x_t = t[:, None, None] * e + (1 - t[:, None, None]) * a
    assert torch.allclose(x_t - t[:, None, None] * (e - a), a, atol=1e-6)
    # Euler integration of a constant field from t=1 to 0 in 10 steps recovers x_1 - v
    x = e.clone(); v = torch.randn(3, 10, 7, generator=g)
    tt, dt, n = 1.0, -0.1, 0
    while tt >= -dt / 2:
        x = x + dt * v; tt += dt; n += 1
    assert n == 10 and torch.allclose(x, e - v, atol=1e-5)


def test_oracle_bf16_mode_close_to_f32_and_sampler_uses_cache():
    P = O.init_params(CFG, 0)
    B = 2
    g = torch.Generator().manual_seed(1)
    obs = dict(images={k: torch.rand(B, 56, 56, 3, generator=g) * 2 - 1 for k in CFG.image_keys},
               image_masks={k: torch.ones(B, dtype=torch.bool) for k in CFG.image_keys},
               tokenized_prompt=torch.randint(0, 512, (B, 24), generator=g),
               tokenized_prompt_mask=torch.tensor([[True] * 24, [True] * 20 + [False] * 4]),
               tokenized_langact_mask=torch.tensor([[False] * 16 + [True] * 8, [False] * 12 + [True] * 8 + [False] * 4]),
               token_loss_mask=torch.ones(B, 24, dtype=torch.bool), sample_mask=torch.ones(B, dtype=torch.bool))
    actions = torch.randn(B, 10, 7, generator=g); noise = torch.randn(B, 10, 7, generator=g)
    t = torch.rand(B, generator=g) * 0.999 + 0.001
    l32, _ = O.compute_loss(P, CFG, obs, actions, noise, t)
    l16, _ = O.compute_loss(P, dataclasses.replace(CFG, emulate_bf16=True), obs, actions, noise, t)
    assert abs(l32.item() - l16.item()) / abs(l32.item()) < 2e-2
    # the KV-cached sampler equals a cache-free recomputation of the joint sequence at one step
    so = {k: v for k, v in obs.items() if k != "tokenized_langact_mask"}
    col = {}
    O.sample_actions(P, CFG, so, noise, num_steps=2, collect=col)
    pt, pm, pa = O.embed_prefix(P, CFG, so)
    st, sm, sa, cond = O.embed_suffix(P, CFG, noise, torch.ones(B))
    Pn, S = pm.shape[1], sm.shape[1]
    full = torch.zeros(B, Pn + S, Pn + S, dtype=torch.bool)
    full[:, :Pn, :Pn] = O.make_attn_mask(pm, pa)
    full[:, Pn:, :Pn] = pm[:, None, :]
    full[:, Pn:, Pn:] = True
    pos = torch.cat([torch.cumsum(pm.long(), 1) - 1, pm.long().sum(-1)[:, None] + torch.arange(S)[None]], 1)
    (_, o1), _ = O.gemma_forward(P, CFG, [pt, st], pos, full, [None, cond])
    v = o1[:, -CFG.action_horizon:] @ P["action_out_proj/kernel"] + P["action_out_proj/bias"]
    assert torch.allclose(v, col["v_t/0"], atol=1e-4, rtol=1e-4)


def test_prefix_path_matches_hf_paligemma_end_to_end():
    """The one cross-check of the WHOLE prefix path that does not rest on recall: HuggingFace
    `PaliGemmaForConditionalGeneration` (tiny config, shared random weights) against the oracle's
    `embed_prefix -> gemma_forward(prefix only) -> Embedder.decode` (lap.py:118-170,209-260; gemma.py:148-154,455-531):
    image tokens enter unscaled, text embeddings x sqrt(width), image + prompt tokens attend bidirectionally, language-action
    tokens causally (make_attn_mask over `tokenized_langact_mask` = HF's token_type_ids == 1 suffix), padding is masked out,
    the LM head is the transposed embedding table.  Two image keys per sample, ragged padding."""
    pytest.importorskip("transformers")
    from transformers import PaliGemmaConfig, PaliGemmaForConditionalGeneration
    from transformers.models.gemma.configuration_gemma import GemmaConfig
    from transformers.models.siglip.configuration_siglip import SiglipVisionConfig

    c, s = CFG.vlm, CFG.img
    IMG_TOK = CFG.vocab_size - 1
    tcfg = GemmaConfig(vocab_size=CFG.vocab_size, hidden_size=c.width, intermediate_size=c.mlp_dim, num_hidden_layers=c.depth,
                       num_attention_heads=c.num_heads, num_key_value_heads=c.num_kv_heads, head_dim=c.head_dim,
                       hidden_act="gelu_pytorch_tanh", hidden_activation="gelu_pytorch_tanh", rms_norm_eps=1e-6,
                       rope_theta=10000.0, attention_bias=False, attention_dropout=0.0, max_position_embeddings=1024)
    vcfg = SiglipVisionConfig(hidden_size=s.width, intermediate_size=s.mlp_dim, num_hidden_layers=s.depth,
                              num_attention_heads=s.num_heads, image_size=CFG.image_size, patch_size=s.patch,
                              layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh", attention_dropout=0.0,
                              projection_dim=c.width, vision_use_head=False)
    hc = PaliGemmaConfig(vision_config=vcfg, text_config=tcfg, image_token_index=IMG_TOK, projection_dim=c.width, hidden_size=c.width,
                         vocab_size=CFG.vocab_size)
    hc._attn_implementation = "eager"
    tcfg._attn_implementation = "eager"; vcfg._attn_implementation = "eager"
    model = PaliGemmaForConditionalGeneration(hc).eval().float()
    P = O.init_params(CFG, seed=6)
    vt = model.model.vision_tower
    vm = vt.vision_model if hasattr(vt, "vision_model") else vt
    lm = model.model.language_model
    blk, lay = "PaliGemma/img/Transformer/encoderblock", "PaliGemma/llm/layers"
    with torch.no_grad():
        vm.embeddings.patch_embedding.weight.copy_(P["PaliGemma/img/embedding/kernel"].permute(3, 2, 0, 1))
        vm.embeddings.patch_embedding.bias.copy_(P["PaliGemma/img/embedding/bias"])
        vm.embeddings.position_embedding.weight.copy_(P["PaliGemma/img/pos_embedding"][0])
        for l, hl in enumerate(vm.encoder.layers):
            hl.layer_norm1.weight.copy_(P[f"{blk}/LayerNorm_0/scale"][l]); hl.layer_norm1.bias.copy_(P[f"{blk}/LayerNorm_0/bias"][l])
            hl.layer_norm2.weight.copy_(P[f"{blk}/LayerNorm_1/scale"][l]); hl.layer_norm2.bias.copy_(P[f"{blk}/LayerNorm_1/bias"][l])
            for nm, proj in (("query", hl.self_attn.q_proj), ("key", hl.self_attn.k_proj), ("value", hl.self_attn.v_proj)):
                proj.weight.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/kernel"][l].reshape(s.width, -1).t())
                proj.bias.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/{nm}/bias"][l].reshape(-1))
            hl.self_attn.out_proj.weight.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/out/kernel"][l].reshape(-1, s.width).t())
            hl.self_attn.out_proj.bias.copy_(P[f"{blk}/MultiHeadDotProductAttention_0/out/bias"][l])
            hl.mlp.fc1.weight.copy_(P[f"{blk}/MlpBlock_0/Dense_0/kernel"][l].t()); hl.mlp.fc1.bias.copy_(P[f"{blk}/MlpBlock_0/Dense_0/bias"][l])
            hl.mlp.fc2.weight.copy_(P[f"{blk}/MlpBlock_0/Dense_1/kernel"][l].t()); hl.mlp.fc2.bias.copy_(P[f"{blk}/MlpBlock_0/Dense_1/bias"][l])
        vm.post_layernorm.weight.copy_(P["PaliGemma/img/Transformer/encoder_norm/scale"])
        vm.post_layernorm.bias.copy_(P["PaliGemma/img/Transformer/encoder_norm/bias"])
        model.model.multi_modal_projector.linear.weight.copy_(P["PaliGemma/img/head/kernel"].t())
        model.model.multi_modal_projector.linear.bias.copy_(P["PaliGemma/img/head/bias"])
        lm.embed_tokens.weight.copy_(P["PaliGemma/llm/embedder/input_embedding"])
        for l, hl in enumerate(lm.layers):
            hl.self_attn.q_proj.weight.copy_(P[f"{lay}/attn/q_einsum/w"][l].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.k_proj.weight.copy_(P[f"{lay}/attn/kv_einsum/w"][l][0].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.v_proj.weight.copy_(P[f"{lay}/attn/kv_einsum/w"][l][1].permute(0, 2, 1).reshape(-1, c.width))
            hl.self_attn.o_proj.weight.copy_(P[f"{lay}/attn/attn_vec_einsum/w"][l].reshape(-1, c.width).t())
            hl.mlp.gate_proj.weight.copy_(P[f"{lay}/mlp/gating_einsum"][l][0].t())
            hl.mlp.up_proj.weight.copy_(P[f"{lay}/mlp/gating_einsum"][l][1].t())
            hl.mlp.down_proj.weight.copy_(P[f"{lay}/mlp/linear"][l].t())
            hl.input_layernorm.weight.copy_(P[f"{lay}/pre_attention_norm/scale"][l])
            hl.post_attention_layernorm.weight.copy_(P[f"{lay}/pre_ffw_norm/scale"][l])
        lm.norm.weight.copy_(P["PaliGemma/llm/final_norm/scale"])
        model.lm_head.weight.copy_(P["PaliGemma/llm/embedder/input_embedding"])          # tied head (gemma.py:153-154)
    B, L = 2, CFG.max_token_len
    g = torch.Generator().manual_seed(2)
    keys = CFG.image_keys
    T = (CFG.image_size // s.patch) ** 2
    obs = dict(images={k: torch.rand(B, CFG.image_size, CFG.image_size, 3, generator=g) * 2 - 1 for k in keys},
               image_masks={k: torch.ones(B, dtype=torch.bool) for k in keys},
               tokenized_prompt=torch.randint(0, IMG_TOK, (B, L), generator=g),
               tokenized_prompt_mask=torch.tensor([[True] * L, [True] * (L - 5) + [False] * 5]),
               tokenized_langact_mask=torch.tensor([[False] * (L - 7) + [True] * 7, [False] * (L - 11) + [True] * 6 + [False] * 5]))
    pt, pm, pa = O.embed_prefix(P, CFG, obs)
    pos = torch.cumsum(pm.long(), 1) - 1
    (ours, _), _ = O.gemma_forward(P, CFG, [pt, None], pos, O.make_attn_mask(pm, pa), [None, None])
    logits = ours @ P["PaliGemma/llm/embedder/input_embedding"].t()
    # HF inputs: [image placeholders of every key | prompt]; pixel_values sample-major, key-minor (the scatter order)
    ids = torch.cat([torch.full((B, len(keys) * T), IMG_TOK), obs["tokenized_prompt"]], 1)
    pix = torch.stack([obs["images"][k] for k in keys], 1).reshape(B * len(keys), CFG.image_size, CFG.image_size, 3).permute(0, 3, 1, 2)
    tt = torch.cat([torch.zeros(B, len(keys) * T, dtype=torch.long), obs["tokenized_langact_mask"].long()], 1)
    with torch.no_grad():
        ref = model(input_ids=ids, pixel_values=pix, attention_mask=pm.long(), token_type_ids=tt, position_ids=pos).logits
    assert ref.shape == logits.shape
    # (padding rows attend to nothing in the oracle and to something arbitrary in HF: compare valid positions only)
    d = (logits - ref)[pm]
    assert d.abs().max() < 2e-4 * max(1.0, ref[pm].abs().max().item()), d.abs().max()
