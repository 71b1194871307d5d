"""Where the persistent denoise-layer chain (csrc/serve_chain.hpp) spends its time: blocks 0 and 255 stamp the 100 MHz clock after every
barrier wait and before every arrival; prints per-stage compute and barrier time (us), averaged over the layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lap_amd import hip
DEV = "cuda"
def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)
B, S, Tp, depth, D, NH, HD, H = 1, 50, int(os.environ.get('TP', 560)), 18, 1024, 8, 256, 4096
PACKED = os.environ.get('PACKED', '1') != '0'
TPAR = os.environ.get('TPAR', '0') != '0'
M = S
x = rnd(M, D, seed=1)
mod = rnd(1, (2 * depth + 1) * 3 * D, scale=0.3, seed=2)
W = [(rnd((NH + 2) * HD, D, scale=D ** -0.5, seed=10 + 4 * l), rnd(D, NH * HD, scale=(NH * HD) ** -0.5, seed=11 + 4 * l),
      rnd(2 * H, D, scale=D ** -0.5, seed=12 + 4 * l), rnd(D, H, scale=H ** -0.5, seed=13 + 4 * l)) for l in range(depth)]
cache = [(rnd(Tp, HD, scale=0.25, seed=100 + l), rnd(Tp, HD, seed=200 + l)) for l in range(depth)]
pos = (torch.arange(M, device=DEV, dtype=torch.int32) + Tp).view(1, M).contiguous()
tab = hip.rope_table(pos, 1, M, M, 0, HD)
kinfo = torch.full((B, Tp + S), 3 << 24, dtype=torch.int32, device=DEV); kinfo[:, Tp:] = (4 << 24) | 0x800001
qinfo = torch.full((B, S), (6 << 24) | 0x800001, dtype=torch.int32, device=DEV)
ctr = hip.serve_chain_counters(DEV)
clk = torch.zeros(8192, dtype=torch.int64, device=DEV)
if PACKED:
    W = [(hip.serve_pack_weight(a, hip.PACK_QKV, HD), hip.serve_pack_weight(b, hip.PACK_PLAIN), hip.serve_pack_weight(c, hip.PACK_GATE_UP),
          hip.serve_pack_weight(d, hip.PACK_PLAIN)) for a, b, c, d in W]
sc = hip.serve_chain_scratch(DEV, D, H, NH, HD, tp=TPAR) if PACKED else None
run = lambda dbg: hip.serve_chain(x, mod, 3 * D, W, cache, tab, qinfo, kinfo, B, S, NH, HD, H, Tp, HD ** -0.5, ctr, debug_clock=dbg, packed_scratch=sc, tp=TPAR)
for _ in range(3): run(None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run(None)
e1.record(); torch.cuda.synchronize()
print(f"packed={int(PACKED)} tp={int(TPAR)} Tp={Tp} chain launch: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us for {depth} layers = {e0.elapsed_time(e1) / 20 * 1e3 / depth:.2f} us per layer; failed={hip.serve_chain_failed(ctr)}")
run(clk); torch.cuda.synchronize()
if TPAR:
    names = ["reduceA", "qkv", "attn+comb", "out", "reduceF", "gateup", "down"]
    NS = len(names)
    for blk, off in (("block 0", 0), ("block 255", 4096)):
        t = clk[off:off + 2 * NS * depth].view(depth, NS, 2)[1:].cpu().double() * 0.01
        prev = torch.cat([clk[off:off + 2 * NS * depth].view(depth, NS, 2)[:-1, -1:, 1].cpu().double() * 0.01, t[:, :-1, 1]], 1)     # wait-exit stamp in front of each stage
        comp = (t[:, :, 0] - prev).mean(0); bar = (t[:, :, 1] - t[:, :, 0]).mean(0)
        print(blk, "compute us:", " ".join(f"{n} {v:.2f}" for n, v in zip(names, comp.tolist())), "| barrier after:", " ".join(f"{n} {v:.2f}" for n, v in zip(names, bar.tolist())),
              f"| layer {float(comp.sum() + bar.sum()):.2f}")
    sys.exit(0)
names = ["qkv", "attn+comb", "out", "gateup", "down"]
NS = len(names)
for blk, off in (("block 0", 0), ("block 255", 4096)):
    t = clk[off:off + 2 * NS * depth + 2].cpu().tolist()
    # stamps per layer: [stage-end(before arrive), after-wait] x 6  -> sequence: e0 w0 e1 w1 ... ; first stamp of the kernel = end of qkv stage 0
    comp = [0.0] * NS; bar = [0.0] * NS
    prev_wait = None
    n = 0
    for l in range(depth):
        for s in range(NS):
            e, w = t[(l * NS + s) * 2], t[(l * NS + s) * 2 + 1]
            if prev_wait is not None and l > 0:
                comp[s] += (e - prev_wait) * 0.01
            if l > 0: bar[s] += (w - e) * 0.01
            prev_wait = w
        n += l > 0
    print(blk, "compute us:", " ".join(f"{nm} {c / n:.2f}" for nm, c in zip(names, comp)), "| barrier after:", " ".join(f"{nm} {b / n:.2f}" for nm, b in zip(names, bar)),
          f"| layer {sum(comp) / n + sum(bar) / n:.2f}")

# attention sub-stamps (serve_chain.hpp): [7] stage start (behind the grid barrier), [0] loads landed, [1] behind the block barrier,
# [2] scores + softmax + P.V done, [3] partial stores performed, [4] runs of the (head, tile) all arrived
for blk, off in (("block 0", 2048), ("block 255", 6144)):
    t = clk[off:off + 8 * depth].view(depth, 8)[1:].cpu().double() * 0.01
    seg = lambda a, b: float((t[:, b] - t[:, a]).mean())
    print(blk, f"attention: loads {seg(7, 0):.2f} | block barrier {seg(0, 1):.2f} | S, softmax, PV {seg(1, 2):.2f} | partial stores {seg(2, 3):.2f} | wait for the runs {seg(3, 4):.2f}")
