"""Timing ablations of the B = 32 train step (results are WRONG by construction; only ms per step counts): which co-running
stream costs the compute stream how much.  ABL = comma list of:
  noexpert  : the action expert's elementwise + GEMM kernels return uninitialised outputs without launching (rows == B * S)
  noopt     : the optimizer / EMA pass is skipped (LAP_ABL_NOOPT read by nothing: done by patching hip.adamw_ema)
  noattn    : attention forward / backward return uninitialised outputs
  nosigwgrad: SigLIP's weight-gradient GEMMs are skipped;  nosigbgrad: every bias column sum is skipped
usage: ABL=noexpert python tools/probes/abl_step.py [steps]"""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

from bench import synthetic_batch
from lap_amd import hip
from lap_amd.config import get_config
from lap_amd.train import TrainingStepRunner, init_train_state

abl = set(filter(None, os.environ.get("ABL", "").split(",")))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, S = 32, 50
dev = torch.device("cuda", 0)
tc = dataclasses.replace(get_config("lap_bench"), batch_size=B, fsdp_devices=1)
state = init_train_state(tc, device=dev, world_size=1, rank=0, use_fsdp=False)
runner = TrainingStepRunner(tc)
batches = [synthetic_batch(tc.model, B, dev, seed=i) for i in range(2)]
_zc = {}
def E(*sh, dt=torch.bfloat16):
    """A ZERO tensor of that shape, allocated once (the skipped kernels never write: it stays zero; garbage / NaN operands would
    change the chip's clocks and make the ablation look better than it is)."""
    k = (tuple(sh), dt)
    if k not in _zc:
        _zc[k] = torch.zeros(sh, dtype=dt, device=dev)
    return _zc[k]
small = lambda t: t is not None and t.shape[0] == B * S

ELEM = "noexpert" in abl or "noexpert_elem" in abl
GEMM = "noexpert" in abl or "noexpert_gemm" in abl
if ELEM or GEMM:
    o = {n: getattr(hip, n) for n in ("linear_fwd", "linear_dgrad", "linear_wgrad", "rmsnorm_fwd", "rmsnorm_bwd", "geglu_fwd", "geglu_bwd",
                                      "gated_residual_fwd", "gated_residual_bwd", "rope_split_fwd", "rope_split_bwd")}
    def linear_fwd(x, wt, out=None, **kw):
        if small(x) and kw.get("bias") is None:
            return out if out is not None else E(x.shape[0], wt.shape[0])
        return o["linear_fwd"](x, wt, out, **kw)
    def linear_dgrad(dy, wt, out=None, **kw):
        if small(dy):
            return out if out is not None else E(dy.shape[0], wt.shape[1])
        return o["linear_dgrad"](dy, wt, out, **kw)
    def linear_wgrad(dy, x, out, **kw):
        if small(dy):
            return out
        return o["linear_wgrad"](dy, x, out, **kw)
    def rmsnorm_fwd(x, *a, **kw):
        if small(x) and kw.get("mod") is not None:
            return E(*x.shape), (E(x.shape[0], dt=torch.float32) if kw.get("save_rstd", True) else None)
        return o["rmsnorm_fwd"](x, *a, **kw)
    def rmsnorm_bwd(x, dy, rstd, **kw):
        if small(x) and kw.get("mod") is not None:
            return kw["dx"] if kw.get("dx") is not None else E(*x.shape)
        return o["rmsnorm_bwd"](x, dy, rstd, **kw)
    def geglu_fwd(gu, pad=False):
        return E(gu.shape[0], gu.shape[1] // 2) if small(gu) else o["geglu_fwd"](gu, pad=pad)
    def geglu_bwd(gu, dact, pad=False):
        return E(*gu.shape) if small(gu) else o["geglu_bwd"](gu, dact, pad=pad)
    def gated_residual_fwd(x, u, *a, **kw):
        return E(*x.shape) if small(x) else o["gated_residual_fwd"](x, u, *a, **kw)
    def gated_residual_bwd(dy, u, *a, **kw):
        return E(*dy.shape) if small(dy) else o["gated_residual_bwd"](dy, u, *a, **kw)
    def rope_split_fwd(qkv, pos, Bq, T_seg, *a, **kw):
        if T_seg == S:
            NH, HD = a[2], a[3]
            return E(Bq * S, NH * HD), E(Bq * S, HD), E(Bq * S, HD)
        return o["rope_split_fwd"](qkv, pos, Bq, T_seg, *a, **kw)
    def rope_split_bwd(dq, dk, dv, pos, Bq, T_seg, *a, **kw):
        if T_seg == S:
            return E(Bq * S, dq.shape[1] + 2 * dk.shape[1])
        return o["rope_split_bwd"](dq, dk, dv, pos, Bq, T_seg, *a, **kw)
    gemm_names = ("linear_fwd", "linear_dgrad", "linear_wgrad")
    for n, f in list(locals().items()):
        if n in o and ((n in gemm_names and GEMM) or (n not in gemm_names and ELEM)):
            setattr(hip, n, f)
if "nosigwgrad" in abl:      # SigLIP's weight gradients (the third stream's GEMMs) are not computed: what they cost the compute stream
    from lap_amd.model import LAP
    _wg = LAP._wgrad
    def _wgrad(self, dy, x, name, **kw):
        if name.startswith("img/"):
            return
        return _wg(self, dy, x, name, **kw)
    LAP._wgrad = _wgrad
if "nosigbgrad" in abl:      # ... and the bias column sums
    from lap_amd.model import LAP
    LAP._bgrad = lambda self, dy, name: None
if "noopt" in abl:
    hip.adamw_ema = lambda *a, **k: None
if "noattn" in abl:
    def attention_fwd(q, k, v, q_len, k_len, Bq, NH, NKV, HD, qinfo=None, kinfo=None, need_lse=True, **kw):
        outs = [E(Bq * q_len[s], NH * HD) if q[s] is not None else None for s in range(2)]
        return outs, (E(Bq, NH, q_len[0] + q_len[1], dt=torch.float32) if need_lse else None)
    def attention_bwd(q, k, v, o_, d_o, lse, q_len, k_len, *a, **kw):
        f = lambda lst: [E(*t.shape) if t is not None else None for t in lst]
        return f(q), f(k), f(v)
    hip.attention_fwd, hip.attention_bwd = attention_fwd, attention_bwd

for i in range(2):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize()
print(f"ABL={','.join(sorted(abl)) or 'none':24s} {(time.perf_counter() - t0) / steps * 1e3:8.2f} ms per step")
