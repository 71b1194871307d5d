"""Per (kernel, grid) stats from a rocprofv3 rocpd database: which GEMM shapes are slow in situ."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
# print(cols)
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
q = f"select name, {gx}, grid_y, workgroup_x, count(*), sum(end-start), avg(end-start) from kernels where name like '%{pat}%' group by name, {gx}, grid_y order by 6 desc limit {int(sys.argv[3]) if len(sys.argv)>3 else 60}"
for name, g, gy, wg, n, tot, avg in cur.execute(q):
    nm = re.sub(r"\(anonymous namespace\)::|void ", "", name)[:70]
    print(f"{nm:70s} grid {g:>8} x{gy:<2} wg {wg:>4} n {n:>5} total {tot/1e6:9.3f} ms avg {avg/1e3:9.2f} us")
