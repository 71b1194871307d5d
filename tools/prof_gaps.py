"""Idle time on the busiest queue of a rocprofv3 kernel trace: gaps between consecutive kernels, grouped by the kernel that FOLLOWS the gap
(who was late) and listed by size.  usage: prof_gaps.py <db> [min_gap_us]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
print("# columns:", [c for c in cols if c in ("queue_id", "stream_id", "queue", "stream", "tid", "start", "end")])
rows = cur.execute(f"select name, start, end, {qcol if qcol else 0} from kernels order by start").fetchall()
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in rs)
    span = rs[-1][2] - rs[0][1]
    print(f"# queue {q}: {len(rs)} kernels, busy {busy/1e6:.1f} ms over a span of {span/1e6:.1f} ms")
main = max(byq.values(), key=len)
# restrict to the steady state: from the (skip+1)-th occurrence of the per-step marker kernel to the last one
marker, skip = (sys.argv[3] if len(sys.argv) > 3 else "fm_mix_kernel"), (int(sys.argv[4]) if len(sys.argv) > 4 else 2)
marks = [r[1] for r in main if marker in r[0]]
if len(marks) > skip + 1:
    t0, t1 = marks[skip], marks[-1]
    main = [r for r in main if t0 <= r[1] < t1]
    nsteps = len(marks) - 1 - skip
    busy = sum(r[2] - r[1] for r in main)
    side = sum(min(r[2], t1) - max(r[1], t0) for q, rs in byq.items() for r in rs if rs is not max(byq.values(), key=len) and r[2] > t0 and r[1] < t1)
    print(f"# window: {nsteps} steps, {(t1 - t0)/1e6/nsteps:.2f} ms/step wall, main-queue busy {busy/1e6/nsteps:.2f} ms/step, other queues {side/1e6/nsteps:.2f} ms/step")
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:60]
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
gaps = collections.defaultdict(lambda: [0, 0.0])
big = []
total = 0.0
end_prev, name_prev = main[0][2], main[0][0]
for name, s, e, _ in main[1:]:
    g = s - end_prev
    if g > 0:
        total += g
        if g > thr:
            gaps[short(name)][0] += 1; gaps[short(name)][1] += g
            big.append((g, short(name_prev), short(name)))
    if e > end_prev:
        end_prev, name_prev = e, name
print(f"# total idle on the main queue {total/1e6:.2f} ms; gaps > {thr/1e3:.0f} us:")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:8.3f} ms  {n:5d} gaps  before {k}")
print("# largest gaps")
for g, a, b in sorted(big, reverse=True)[:25]:
    print(f"{g/1e3:9.1f} us  after {a:60s} before {b}")
# what the other queues were doing during the gaps in front of one kernel: prof_gaps.py <db> <min_us> <marker> <skip> <kernel substring>
if len(sys.argv) > 5:
    target = sys.argv[5]
    print(f"# other queues during the first gaps in front of {target}")
    shown = 0
    end_prev = main[0][2]
    for name, s, e, _ in main[1:]:
        if target in name and s - end_prev > thr and shown < 4:
            shown += 1
            print(f"  gap {end_prev % 10**9 / 1e3:.1f} .. {s % 10**9 / 1e3:.1f} us ({(s - end_prev) / 1e3:.1f} us)")
            for r in main:      # the main queue's own kernels around the gap
                if r[2] > end_prev - 150e3 and r[1] < s + 5e3:
                    print(f"     main    : {r[1] % 10**9 / 1e3:10.1f} .. {r[2] % 10**9 / 1e3:10.1f}  {short(r[0])}")
            for q, rs in byq.items():
                for r in rs:
                    if r[2] > end_prev - 20e3 and r[1] < s + 5e3 and r[3] != main[0][3]:
                        print(f"     queue {q}: {r[1] % 10**9 / 1e3:10.1f} .. {r[2] % 10**9 / 1e3:10.1f}  {short(r[0])}")
        if e > end_prev:
            end_prev = e
