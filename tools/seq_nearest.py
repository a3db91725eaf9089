"""Launch-order dump of ONE shape of the `nearest` workload (the dispatches between the last two k_project_verts launches of a rocprofv3
rocpd kernel trace): short name, duration, gap to the previous kernel's end.  Usage: python tools/seq_nearest.py results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration from kernels order by start").fetchall()
mark = [i for i, r in enumerate(rows) if 'k_project_verts' in r[0]]
a, b = mark[-2], mark[-1]
tot = 0
for i in range(a, b):
    n, st, du = rows[i]
    m = re.search(r'(k_[a-z0-9_]+)', n)
    short = m.group(1) if m else n[:60]
    gap = (st - (rows[i - 1][1] + rows[i - 1][2])) / 1e3 if i > a else 0.0
    tot += du
    print(f"{i - a:3d} {short:60s} {du / 1e3:8.2f} us  gap {gap:6.2f}")
print(f"kernel time {tot / 1e3:.1f} us, wall {(rows[b - 1][1] + rows[b - 1][2] - rows[a][1]) / 1e3:.1f} us over {b - a} dispatches")
