"""Phases of ONE train step on the main queue of a rocprofv3 rocpd database: SigLIP forward | LLM forward | head + loss | LLM backward | SigLIP
backward | tail, delimited by the first launch of each phase's attention kernel; per phase the wall time, the main queue's busy time and its
ten heaviest kernels, plus what the other queues ran meanwhile.
usage: prof_phases.py <db> [out.txt]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", n).split("(")[0][:48]
starts = [i for i, r in enumerate(rows) if "fm_mix_kernel" in r[0]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0, t1 = step[0][1], rows[b][1]
main_q = max(set(r[3] for r in step), key=lambda q: sum(r[2] - r[1] for r in step if r[3] == q))
def first(sub, after=0):
    for r in step:
        if sub in r[0] and r[1] >= after and r[3] == main_q:
            return r[1]
    return None
def last_end(sub):
    e = None
    for r in step:
        if sub in r[0] and r[3] == main_q:
            e = r[2]
    return e
marks = [("step head", t0), ("SigLIP forward", first("layernorm_fwd")), ("LLM forward", first("attn_dma_q_kernel<256, 0>")),
         ("head + loss + head backward", last_end("attn_dma_q_kernel<256, 0>")), ("LLM backward", first("attn_dma_kv_kernel<256>") and first("attn_dma_q_kernel<256, 1>")),
         ("SigLIP backward", last_end("attn_dma_kv_kernel<256>")), ("tail", last_end("attn_dma_kv_kernel<72>")), ("end", t1)]
out = [f"# step of {(t1 - t0) / 1e6:.2f} ms, main queue {main_q}"]
for (name, s), (_, e) in zip(marks[:-1], marks[1:]):
    if s is None or e is None:
        continue
    mine = [r for r in step if r[3] == main_q and r[1] >= s and r[1] < e]
    busy = sum(r[2] - r[1] for r in mine)
    out.append(f"## {name}: {(e - s) / 1e6:.2f} ms wall, main queue busy {busy / 1e6:.2f} ms in {len(mine)} kernels")
    agg = {}
    for n, ks, ke, q in mine:
        k = short(n); d, c = agg.get(k, (0, 0)); agg[k] = (d + ke - ks, c + 1)
    for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        out.append(f"   {d / 1e6:7.2f} ms  x{c:4d}  avg {d / c / 1e3:7.1f} us  {k}")
    oth = {}
    for n, ks, ke, q in rows:
        if q != main_q and ke > s and ks < e:
            k = f"q{q} " + short(n); oth[k] = oth.get(k, 0) + min(ke, e) - max(ks, s)
    for k, d in sorted(oth.items(), key=lambda kv: -kv[1])[:6]:
        out.append(f"      beside: {d / 1e6:7.2f} ms  {k}")
txt = "\n".join(out)
open(sys.argv[2], "w").write(txt + "\n") if len(sys.argv) > 2 else print(txt)
