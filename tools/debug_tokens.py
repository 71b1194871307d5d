import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dataclasses
from oracle import lap_oracle as O
from tests.common import debug_model_cfg, make_inputs, oracle_cfg, rel, to_observation
from lap_amd.model import LAP
from lap_amd import hip
cfg = debug_model_cfg(); oc = oracle_cfg(cfg)
P = O.init_params(oc, seed=13)
obs, _, noise, _ = make_inputs(cfg, B=3, ragged=True)
so = dict(obs); so.pop("tokenized_langact_mask")
table = P["PaliGemma/llm/embedder/input_embedding"]
pt, pm, par = O.embed_prefix(P, oc, so)
attn = O.make_attn_mask(pm, par)
B, size = pm.shape
ar = torch.arange(size)
seqlen = (pm.long()*ar).max(-1).values + 1
plen = pm.long().sum(-1)
pos = torch.cumsum(pm.long(), -1) - 1
(pre,_), cache = O.gemma_forward(P, oc, [pt, None], pos, attn, [None,None])
lg0 = pre[torch.arange(B), seqlen-1][:,None] @ table.t()
tok = lg0.argmax(-1)
emb = table[tok.long()]*math.sqrt(oc.vlm.width)
inr = (ar[None] >= (seqlen-plen)[:,None]) & (ar[None] < seqlen[:,None])
mask = torch.cat([inr, torch.ones(B,1,dtype=torch.bool)],1)[:,None,:]
col = {}
(pre1,_), _ = O.gemma_forward(P, oc, [emb, None], plen[:,None], mask, [None,None], kv_cache=cache, collect=col)
print("oracle cache k shape", cache[0][0].shape, "tok", tok.view(-1).tolist(), "plen", plen.tolist(), "seqlen", seqlen.tolist())
model = LAP(cfg, params=P, device="cuda")
o = to_observation(so | {"tokenized_langact_mask": None}, "cuda")
# engine: replicate sample_tokens internals
from lap_amd.observation import preprocess_observation
ob = preprocess_observation(o, train=False, image_keys=cfg.image_keys, image_resolution=cfg.image_resolution)
x0, Pn, _ = model._embed_prefix(ob, False)
qinfo_p, kinfo_p, ppos = model._serve_infos(ob, 1)[:3]
ecache = []
xf0, _, _ = model._llm_fwd(x0, None, None, ppos, qinfo_p, kinfo_p, B, Pn, 0, False, cache_out=ecache)
print("prefill x rel", rel(xf0.view(B, Pn, -1)[pm], None) if False else "")
k_or = cache[0][0][:, :, 0, :]   # [B, size, H]
print("cache k layer0 rel (valid rows)", rel(ecache[0][0].view(B, Pn, -1).float().cpu()[pm], k_or[pm]))
dev = "cuda"
prefix_mask, _ = model._prefix_masks(ob)
ard = torch.arange(Pn, device=dev)
sl = (prefix_mask.to(torch.int64) * ard).max(-1).values + 1
pl = prefix_mask.sum(-1)
inr_d = (ard[None] >= (sl - pl)[:, None]) & (ard[None] < sl[:, None])
kinfo_prefix = (inr_d.to(torch.int32) << 24).contiguous()
qinfo_d = torch.full((B, 1), (1 << 24) | 0xFFFFFF, dtype=torch.int32, device=dev)
gen = [(None, None)] * model.v.depth
token = tok.view(-1).to(torch.int32).to(dev)
posd = pl.to(torch.int32).view(B, 1).contiguous()
# manual first layer
v = model.v; NH, HD, KV, Dv = v.num_heads, v.head_dim, v.num_kv_heads, v.width
x = torch.empty((B, Dv), dtype=torch.bfloat16, device=dev)
rows, lo, hi = model.ps.embed_rows()
hip.embed_gather(rows, token.view(B, 1).contiguous(), x, B, 1, Dv, 1, 0, math.sqrt(Dv), lo, hi)
print("emb rel", rel(x.float().cpu(), emb[:, 0]))
lg = model._vlm_decode_step(token, posd, 0, ecache, gen, qinfo_d, kinfo_prefix, B, Pn)
print("decode logits rel", rel(lg.cpu(), (pre1 @ table.t())[:, 0]))
# layer-0 pieces
p = "llm/0/"
h, _ = hip.rmsnorm_fwd(x, scale=model.F(p + "n_attn"), save_rstd=False)
qkv = hip.linear_fwd(h, model.W(p + "wqkv0"))
q, k, vv = hip.rope_split_fwd(qkv, posd, B, 1, 1, 0, NH, HD, HD ** -0.5)
# oracle layer-0 q,k
lay = "PaliGemma/llm/layers"
y, _ = O.rmsnorm(emb, scale=P[f"{lay}/pre_attention_norm/scale"][0], r=lambda t: t)
qo = torch.einsum("btd,ndh->btnh", y, P[f"{lay}/attn/q_einsum/w"][0])
kvo = torch.einsum("bsd,xkdh->xbskh", y, P[f"{lay}/attn/kv_einsum/w"][0])
qo = O.apply_rope(qo, plen[:, None]) * HD ** -0.5
ko = O.apply_rope(kvo[0], plen[:, None])
print("q rel", rel(q.float().cpu().view(B, NH, HD), qo[:, 0]), "k rel", rel(k.float().cpu(), ko[:, 0, 0]), "v rel", rel(vv.float().cpu(), kvo[1][:, 0, 0]))
ck, cv = ecache[0]
kinfo = torch.cat([kinfo_prefix, torch.full((B, 1), 1 << 24, dtype=torch.int32, device=dev)], 1).contiguous()
oo, _ = hip.attention_fwd([None, q], [ck, k], [cv, vv], [0, 1], [Pn, 1], B, NH, KV, HD, qinfo_d, kinfo, need_lse=False)
# oracle attention for layer 0
kk = torch.cat([cache[0][0], ko], 1); vvv = torch.cat([cache[0][1], kvo[1]], 1)
logits = torch.einsum("btnh,bskh->bnts", qo, kk)
logits = torch.where(mask[:, None], logits, torch.tensor(-1e30))
pr = torch.softmax(logits, -1)
enc = torch.einsum("bnts,bskh->btnh", pr, vvv)
print("attn rel", rel(oo[1].float().cpu().view(B, NH, HD), enc[:, 0]))
for l in range(v.depth):
    pass
xx = x
gen2 = [(None, None)] * v.depth
for l in range(v.depth):
    p = f"llm/{l}/"
    h, _ = hip.rmsnorm_fwd(xx, scale=model.F(p + "n_attn"), save_rstd=False)
    qkv = hip.linear_fwd(h, model.W(p + "wqkv0"))
    q, k, vv = hip.rope_split_fwd(qkv, posd, B, 1, 1, 0, NH, HD, HD ** -0.5)
    ck, cv = ecache[l]
    oo, _ = hip.attention_fwd([None, q], [ck, k], [cv, vv], [0, 1], [Pn, 1], B, NH, KV, HD, qinfo_d, kinfo, need_lse=False)
    xa = hip.linear_fwd(oo[1], model.W(p + "wo0"), residual=xx)
    xa_ref = (oo[1].float() @ model.W(p + "wo0").float().t() + xx.float())
    hf, _ = hip.rmsnorm_fwd(xa, scale=model.F(p + "n_ffw"), save_rstd=False)
    gu = hip.linear_fwd(hf, model.W(p + "wgu0"))
    gu_ref = hf.float() @ model.W(p + "wgu0").float().t()
    act = hip.geglu_fwd(gu)
    xn = hip.linear_fwd(act, model.W(p + "wd0"), residual=xa)
    xn_ref = act.float() @ model.W(p + "wd0").float().t() + xa.float()
    print(l, "x rel vs oracle", rel(xn.float().cpu(), col[f"llm/layer{l:02d}/x0"][:, 0]), "| wo gemm", rel(xa.float(), xa_ref), "gu gemm", rel(gu.float(), gu_ref), "wd gemm", rel(xn.float(), xn_ref),
          "shapes", tuple(model.W(p + "wo0").shape), tuple(model.W(p + "wgu0").shape), tuple(model.W(p + "wd0").shape))
    xx = xn
