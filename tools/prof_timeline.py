"""Ordered kernel timeline of the LAST occurrence of a repeating unit (one hipGraph replay of the sampler, one train step) in a
rocprofv3 rocpd database: name, grid (workgroups), start offset, duration, gap to the previous kernel's end.
usage: prof_timeline.py <db> <marker kernel substring> [out.txt]   (the unit starts at the marker's last-but-one occurrence)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
marker = sys.argv[2]
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, queue_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
if len(marks) < 2:
    sys.exit(f"marker {marker!r} seen {len(marks)} times")
a, b = marks[-2], marks[-1]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:70]
out = [f"# {b - a} kernels, {(rows[b][1] - rows[a][1]) / 1e3:.1f} us from marker to marker"]
t0, prev_end = rows[a][1], rows[a][1]
for name, s, e, gx, gy, gz, wx, wy, wz, q in rows[a:b]:
    wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
    out.append(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  q{q} wg {wgs:6d} x{wx * wy * wz:4d}  {short(name)}")
    prev_end = max(prev_end, e)
txt = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
else:
    print(txt)
