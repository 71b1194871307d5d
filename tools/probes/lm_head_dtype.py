"""LM head dtype flow (gemma.py:153-154: bf16 pre-logits x F32 table -> f32 logits) vs the engine's bf16 mirror of the table and
bf16 dlogits: deviation of the per-row cross entropy and of d(pre_logits) at vocabulary 257,152, width 2048."""
import torch
torch.manual_seed(0)
dev = "cuda"
V, D, R = 257152, 2048, 512
for std in (0.02, 1.0):
    T = torch.randn(V, D, device=dev) * std
    x = torch.randn(R, D, device=dev)
    x = (x / x.pow(2).mean(-1, keepdim=True).sqrt()).bfloat16().float()          # RMS-normed pre-logits, bf16 values
    tgt = torch.randint(0, V, (R,), device=dev)
    T16 = T.bfloat16().float()
    lo = (T - T16).bfloat16().float()
    out = {}
    for name, tab in (("f32 table", T), ("bf16 table", T16), ("hi+lo", None)):
        lg = x @ T16.t() + x @ lo.t() if tab is None else x @ tab.t()
        lse = torch.logsumexp(lg, -1)
        nll = lse - lg.gather(1, tgt[:, None])[:, 0]
        p = torch.softmax(lg, -1)
        p[torch.arange(R), tgt] -= 1.0
        if name == "bf16 table":
            dpl = p.bfloat16().float() @ T16                                   # engine: bf16 dlogits x bf16 table
        elif name == "hi+lo":
            ph = p.bfloat16().float(); pl_ = (p - ph).bfloat16().float()
            dpl = ph @ T16 + pl_ @ T16 + ph @ lo
        else:
            dpl = p @ T
        out[name] = (nll, dpl)
    ref = out["f32 table"]
    for name in ("bf16 table", "hi+lo"):
        nll, dpl = out[name]
        print(f"std {std}: {name:10s} CE mean {ref[0].mean().item():.4f}; |dCE| max {((nll - ref[0]).abs().max()).item():.2e} mean {((nll - ref[0]).abs().mean()).item():.2e}"
              f" (rel of mean CE {((nll - ref[0]).abs().mean() / ref[0].mean()).item():.1e}); d pre_logits rel L2 {((dpl - ref[1]).norm() / ref[1].norm()).item():.2e}"
              f" (bf16 rounding of the result alone: {((ref[1].bfloat16().float() - ref[1]).norm() / ref[1].norm()).item():.2e})")
