"""Per-shape census of the train step's GEMM launches: in-situ HIP-event time per (M, N, K, layout, epilogue) group.
Usage: python tools/gemm_census.py [steps]   (LAP-3B bench workload, batch 32)"""
import collections, dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lap_amd import hip
from lap_amd.config import get_config
from lap_amd.train import TrainingStepRunner, init_train_state

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
tc = dataclasses.replace(get_config("lap_bench"), batch_size=32)
state = init_train_state(tc, device=dev, world_size=1, rank=0, use_fsdp=False)
runner = TrainingStepRunner(tc)
batches = [bench.synthetic_batch(tc.model, 32, dev, seed=i) for i in range(2)]
orig = hip.gemm
recs = []
on = False
def gemm(a, b, out, *, M, N, K, **kw):
    if not on:
        return orig(a, b, out, M=M, N=N, K=K, **kw)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(a, b, out, M=M, N=N, K=K, **kw); e.record()
    key = (M, N, K, "k" if kw.get("a_kc", True) else "m", "k" if kw.get("b_kc", True) else "n",
           "f32" if out.dtype == torch.float32 else "bf16", int(bool(kw.get("accum"))), int(bool(kw.get("gelu"))),
           int(kw.get("bias") is not None), int(kw.get("residual") is not None))
    recs.append((key, s, e)); return r
hip.gemm = gemm
for i in range(2):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize(); on = True
for i in range(steps):
    state, info = runner(0, state, batches[i % 2], state.step)
torch.cuda.synchronize(); on = False
g = collections.defaultdict(lambda: [0, 0.0])
for key, s, e in recs:
    g[key][0] += 1; g[key][1] += s.elapsed_time(e)
tot = sum(v[1] for v in g.values()) / steps
fl = sum(2.0 * k[0] * k[1] * k[2] * v[0] for k, v in g.items()) / steps
print(f"GEMM total {tot:.1f} ms/step, {fl/1e12:.1f} TFLOP/step, {fl/tot/1e9:.0f} TFLOP/s; {len(g)} distinct groups")
def iso(k):
    M, N, K, al, bl, od, acc, gelu, bias, res = k
    rnd = lambda *sh: (torch.rand(*sh, device=dev) - 0.5).bfloat16()
    a = rnd(M, K) if al == "k" else rnd(K, M)
    b = rnd(N, K) if bl == "k" else rnd(K, N)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if od == "f32" else torch.bfloat16)
    kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldb=b.stride(0), ldc=N, a_kc=al == "k", b_kc=bl == "k", accum=bool(acc), gelu=bool(gelu))
    if bias: kw["bias"] = torch.zeros(N, device=dev, dtype=torch.float32)
    if res: kw["residual"] = rnd(M, N); kw["ldr"] = N
    for _ in range(3): orig(a, b, out, **kw)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): orig(a, b, out, **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10
del state, runner
torch.cuda.empty_cache()
print(f"{'M':>6} {'N':>6} {'K':>6} A B out  acc gelu bias res |  n/step  avg us   TF/s  ms/step  lost-vs-1.3PF ms   iso us  iso TF/s")
rows = []
for k, v in g.items():
    n = v[0] / steps; ms = v[1] / steps; f = 2.0 * k[0] * k[1] * k[2] * n
    rows.append((ms - f / 1.3e12, k, n, ms, f))
for lost, k, n, ms, f in sorted(rows, reverse=True):
    print(f"{k[0]:>6} {k[1]:>6} {k[2]:>6} {k[3]} {k[4]} {k[5]:>4} {k[6]:>3} {k[7]:>4} {k[8]:>4} {k[9]:>3} | {n:7.0f} {ms/n*1e3:7.1f} {f/ms/1e9:6.0f} {ms:8.2f} {lost:8.2f}  {(t:=iso(k))*1e3:8.1f} {2.0*k[0]*k[1]*k[2]/t/1e9:6.0f}")
