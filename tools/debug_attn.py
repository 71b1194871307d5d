import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lap_amd import hip
from tests.test_kernels_gpu import _attn_ref, _mask_from_info, rel_err
torch.manual_seed(0)
def run(B, T, NH, HD, scale, masked, lang=8):
    q = (torch.randn(B, T, NH * HD, device="cuda") * scale).bfloat16()
    k = (torch.randn(B, T, HD, device="cuda") * scale).bfloat16()
    v = torch.randn(B, T, HD, device="cuda").bfloat16()
    qi = ki = None; mask = None
    if masked:
        cs = torch.zeros(B, T, dtype=torch.int32, device="cuda"); cs[:, T - lang:] = torch.arange(1, lang + 1, device="cuda", dtype=torch.int32)
        qi = ((1 << 24) | cs).contiguous(); ki = ((3 << 24) | cs).contiguous()
        mask = _mask_from_info(qi, ki)
    (o, _), lse = hip.attention_fwd([q, None], [k, None], [v, None], [T, 0], [T, 0], B, NH, 1, HD, qi, ki)
    ref = _attn_ref(q.float().view(B, T, NH, HD), k.float().view(B, T, 1, HD), v.float().view(B, T, 1, HD), mask, NH, 1)
    e = rel_err(o.view(B, T, NH, HD), ref)
    # per-row error
    d = (o.view(B, T, NH, HD).float() - ref).norm(dim=(2, 3)) / ref.norm(dim=(2, 3))
    print(f"B{B} T{T} NH{NH} HD{HD} scale{scale} masked{masked}: err {e:.3e}; worst rows {torch.topk(d.flatten(), 5).indices.tolist()} {torch.topk(d.flatten(),5).values.tolist()}")
for sc in (0.5, 2.0, 6.0):
    for masked in (False, True):
        run(2, 56, 8, 16, sc, masked)
run(2, 64, 8, 16, 2.0, True); run(2, 40, 8, 16, 2.0, True); run(2, 56, 8, 256, 0.5, True); run(1, 56, 1, 16, 4.0, False)
