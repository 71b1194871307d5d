import sys, os, dataclasses, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import lap_oracle as O
from tests.common import debug_model_cfg, make_inputs, oracle_cfg, rel, to_observation
from lap_amd.model import LAP
from lap_amd import hip
cfg = debug_model_cfg(); oc = oracle_cfg(cfg); P = O.init_params(oc, seed=7)
B = 2
obs, actions, noise, time = make_inputs(cfg, B=B, ragged=False)
model = LAP(cfg, params=P, device="cuda")
col = {}
model.compute_loss(0, to_observation(obs, "cuda"), actions.cuda(), noise=noise.cuda(), time=time.cuda(), collect=col)
x0 = col["x0_in"]; pos = col["pos"]
T = model.n_img_tok; L = cfg.max_token_len; Pn = 2 * T + L; S = cfg.action_horizon
v = model.v; NH, HD = v.num_heads, v.head_dim
lay = "PaliGemma/llm/layers"
xf = x0.float().cpu().view(B, Pn, -1)
# oracle pieces, layer 0, prefix stream only
h_ref, _ = O.rmsnorm(xf, scale=P[f"{lay}/pre_attention_norm/scale"][0])
q_ref = torch.einsum("btd,ndh->btnh", h_ref, P[f"{lay}/attn/q_einsum/w"][0])
kv_ref = torch.einsum("bsd,xkdh->xbskh", h_ref, P[f"{lay}/attn/kv_einsum/w"][0])
ppos = pos[:, :Pn].cpu().long()
qr_ref = O.apply_rope(q_ref, ppos) * HD ** -0.5
kr_ref = O.apply_rope(kv_ref[0], ppos)
h, _ = hip.rmsnorm_fwd(x0, scale=model.F("llm/0/n_attn"))
print("h", rel(h.view(B, Pn, -1), h_ref))
qkv = hip.linear_fwd(h, model.W("llm/0/wqkv0"))
qkv_ref = torch.cat([q_ref.reshape(B, Pn, -1), kv_ref[0].reshape(B, Pn, -1), kv_ref[1].reshape(B, Pn, -1)], -1)
print("qkv", rel(qkv.view(B, Pn, -1), qkv_ref))
q, k, vv = hip.rope_split_fwd(qkv, pos, B, Pn, Pn + S, 0, NH, HD, HD ** -0.5)
print("q rope", rel(q.view(B, Pn, NH, HD), qr_ref), "k rope", rel(k.view(B, Pn, 1, HD), kr_ref), "v", rel(vv.view(B, Pn, 1, HD), kv_ref[1]))
qinfo, kinfo, pos2 = model._train_infos(to_observation(obs, "cuda"), S)
mask = c_mask = None
c = {}
O.compute_loss(P, oc, obs, actions, noise, time, collect=c)
mask = c["mask"]
qi, ki = qinfo.cpu(), kinfo.cpu()
m2 = (((qi >> 24)[:, :, None] & (ki >> 24)[:, None, :]) != 0) & ((ki & 0xFFFFFF)[:, None, :] <= (qi & 0xFFFFFF)[:, :, None])
print("mask equal", torch.equal(m2, mask), (m2 != mask).sum().item())
# attention prefix-only with prefix part of the mask
logits = torch.einsum("btnh,bsh->bnts", qr_ref, kr_ref[:, :, 0])
logits = logits.masked_fill(~mask[:, None, :Pn, :Pn], -1e30)
o_ref = torch.einsum("bnts,bsh->btnh", torch.softmax(logits, -1), kv_ref[1][:, :, 0])
(o, _), _ = hip.attention_fwd([q, None], [k, None], [vv, None], [Pn, 0], [Pn, 0], B, NH, 1, HD, qinfo[:, :Pn].contiguous(), kinfo[:, :Pn].contiguous())
print("attn o", rel(o.view(B, Pn, NH, HD), o_ref))
xa_ref = xf + torch.einsum("btnh,nhd->btd", o_ref, P[f"{lay}/attn/attn_vec_einsum/w"][0])
xa = hip.linear_fwd(o, model.W("llm/0/wo0"), residual=x0)
print("xa", rel(xa.view(B, Pn, -1), xa_ref))
hf_ref, _ = O.rmsnorm(xa_ref, scale=P[f"{lay}/pre_ffw_norm/scale"][0])
hf, _ = hip.rmsnorm_fwd(xa, scale=model.F("llm/0/n_ffw"))
print("hf", rel(hf.view(B, Pn, -1), hf_ref))
wg = P[f"{lay}/mlp/gating_einsum"][0]
act_ref = O.gelu_tanh(hf_ref @ wg[0]) * (hf_ref @ wg[1])
gu = hip.linear_fwd(hf, model.W("llm/0/wgu0")); act = hip.geglu_fwd(gu)
print("act", rel(act.view(B, Pn, -1), act_ref))
xn_ref = xa_ref + act_ref @ P[f"{lay}/mlp/linear"][0]
xn = hip.linear_fwd(act, model.W("llm/0/wd0"), residual=xa)
print("xn", rel(xn.view(B, Pn, -1), xn_ref), "vs collected", rel(col["llm/layer00/x0"], xn))
print("---- sensitivity check")
qe = q.float().cpu().view(B, Pn, NH, HD); ke = k.float().cpu().view(B, Pn, HD); ve = vv.float().cpu().view(B, Pn, HD)
lg = torch.einsum("btnh,bsh->bnts", qe, ke).masked_fill(~mask[:, None, :Pn, :Pn], -1e30)
print("max |logit|", lg[lg > -1e29].abs().max().item(), "ref max", logits[logits > -1e29].abs().max().item())
o_self = torch.einsum("bnts,bsh->btnh", torch.softmax(lg, -1), ve)
print("engine attn vs torch-on-engine-qkv", rel(o.view(B, Pn, NH, HD), o_self), " torch-on-engine-qkv vs oracle", rel(o_self, o_ref))
d = (o.view(B, Pn, NH, HD).float().cpu() - o_self)
n3=lambda t,dims: (t**2).sum(dims).sqrt()
print("err per head", (n3(d,(0,1,3)) / n3(o_self,(0,1,3))).tolist())
print("err per batch", (n3(d,(1,2,3)) / n3(o_self,(1,2,3))).tolist())
et = n3(d,(2,3)) / n3(o_self,(2,3))
print("err per row b0", [round(x, 2) for x in et[0].tolist()])
print("qinfo b0", [hex(x) for x in qinfo[0, :Pn].tolist()][-12:], "kinfo b0", [hex(x) for x in kinfo[0, :Pn].tolist()][-12:])
# no-mask run on same data
(o2, _), _ = hip.attention_fwd([q, None], [k, None], [vv, None], [Pn, 0], [Pn, 0], B, NH, 1, HD)
lg2 = torch.einsum("btnh,bsh->bnts", qe, ke)
o2_ref = torch.einsum("bnts,bsh->btnh", torch.softmax(lg2, -1), ve)
print("nomask err", rel(o2.view(B, Pn, NH, HD), o2_ref))
# random data of same shape with these infos
qr = torch.randn_like(q); kr = torch.randn_like(k); vr = torch.randn_like(vv)
(o3, _), _ = hip.attention_fwd([qr, None], [kr, None], [vr, None], [Pn, 0], [Pn, 0], B, NH, 1, HD, qinfo[:, :Pn].contiguous(), kinfo[:, :Pn].contiguous())
lg3 = torch.einsum("btnh,bsh->bnts", qr.float().cpu().view(B, Pn, NH, HD), kr.float().cpu().view(B, Pn, HD)).masked_fill(~mask[:, None, :Pn, :Pn], -1e30)
o3_ref = torch.einsum("bnts,bsh->btnh", torch.softmax(lg3, -1), vr.float().cpu().view(B, Pn, HD))
print("random data same infos err", rel(o3.view(B, Pn, NH, HD), o3_ref))
