"""Which gradients differ between the one-stream schedule and the off-path weight-gradient modes (debugging aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
from oracle import lap_oracle as O
from tests.common import make_inputs, oracle_cfg, rel, to_observation
from tests.test_model_parity_gpu import _full_width_cfg, _engine
mp = pytest.MonkeyPatch()
cfg = _full_width_cfg(mp)
P = O.init_params(oracle_cfg(cfg), seed=11)
obs, actions, noise, time = make_inputs(cfg, B=3, ragged=True)
model = _engine(cfg, P)
o = to_observation(obs, "cuda")
def run(dual, mode):
    model.dual_stream, model.wgrad_stream = dual, mode
    for g in model.ps.grad.values():
        g.zero_()
    loss, _ = model.loss_and_grad(0, o, actions.cuda(), noise=noise.cuda(), time=time.cuda())
    torch.cuda.synchronize()
    return {n: model.ps.g(n).detach().clone() for n in model.ps.names()}
g0 = run(False, "")
for mode in sys.argv[1:]:
    for rep in range(3):
        g = run(os.environ.get('PROBE_DUAL', '1') == '1', mode)
        bad = [(n, rel(g[n].float().cpu(), g0[n].float().cpu())) for n in g0 if rel(g[n].float().cpu(), g0[n].float().cpu()) > 1e-5]
        print(mode, rep, [(n, "%.1e" % r) for n, r in bad if not n.startswith("img/0/")], flush=True)
