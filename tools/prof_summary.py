"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel stats table (markdown)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path, out=None, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows[:top]:
        lines.append(f"| `{short(name)}` | {n} | {tot/1e6:.3f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} |")
    lines.append(f"\ntotal kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
