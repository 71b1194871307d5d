"""Who runs beside whom: for every launch of the named kernels on the main queue inside the LAST train step of a rocprofv3 rocpd
database, its duration and the kernels of the other queues that overlap it (share of the launch's duration).
usage: prof_overlap.py <db> <step marker substring> <kernel substring> [<kernel substring> ...]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
marker, wanted = sys.argv[2], sys.argv[3:]
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = marks[-2], marks[-1]
t0, t1 = rows[a][1], rows[b][1]
main_q = max(set(r[3] for r in rows[a:b]), key=lambda q: sum(r[2] - r[1] for r in rows[a:b] if r[3] == q))
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", n).split("(")[0][:40]
others = [r for r in rows if r[3] != main_q and r[2] > t0 and r[1] < t1]
print(f"# step of {(t1 - t0) / 1e6:.2f} ms, main queue {main_q}")
for q in sorted(set(r[3] for r in rows[a:b])):
    agg = {}
    for name, s, e, qq in rows[a:b]:
        if qq == q:
            k = short(name); agg[k] = (agg.get(k, (0, 0))[0] + (e - s), agg.get(k, (0, 0))[1] + 1)
    tot = sum(v[0] for v in agg.values())
    print(f"## queue {q}: {tot / 1e6:.2f} ms busy in this step")
    for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:(40 if q == main_q else 12)]:
        print(f"   {d / 1e6:8.2f} ms  x{n:4d}  avg {d / n / 1e3:8.1f} us  {k}")
for w in wanted:
    sel = [r for r in rows[a:b] if r[3] == main_q and w in r[0]]
    tot = sum(r[2] - r[1] for r in sel) / 1e6
    print(f"## {w}: {len(sel)} launches, {tot:.2f} ms")
    for name, s, e, q in sel:
        d = e - s
        ov = {}
        for on, os_, oe, oq in others:
            x = min(e, oe) - max(s, os_)
            if x > 0:
                ov[short(on)] = ov.get(short(on), 0) + x
        txt = ", ".join(f"{k} {v / d:.0%}" for k, v in sorted(ov.items(), key=lambda kv: -kv[1])[:4])
        print(f"  +{(s - t0) / 1e6:8.2f} ms  {d / 1e3:8.1f} us  | {txt}")
