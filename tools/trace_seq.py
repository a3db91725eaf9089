"""Dump one DDNM step (the launches between the last two k_ddnm_update dispatches) of a rocprofv3 rocpd kernel trace in launch
order: short kernel name, grid, duration.  Usage: python tools/trace_seq.py results.db [out.txt]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if 'grid' in c.lower()]
q = "select name, start, duration" + "".join(", " + c for c in gx) + " from kernels order by start"
rows = db.execute(q).fetchall()
upd = [i for i, r in enumerate(rows) if 'k_ddnm_update' in r[0]]
a, b = upd[-2] + 1, upd[-1] + 1
def short(n):
    m = re.search(r'(k_[a-z0-9_]+)', n)
    t = re.findall(r'ILi(\d+)E|Li(\d+)E|Lb(\d)E', n)
    tp = ''.join(x or y or z for x, y, z in t)
    return (m.group(1) if m else n[:30]) + ('<' + tp + '>' if tp else '')
out = [f"# columns: idx kernel grid({','.join(gx)}) duration_us gap_before_us   total {sum(r[2] for r in rows[a:b]) / 1e6:.3f} ms over {b - a} launches, wall {(rows[b-1][1] + rows[b-1][2] - rows[a][1]) / 1e6:.3f} ms"]
for i, r in enumerate(rows[a:b]):
    prev = rows[a + i - 1]
    out.append(f"{i:4d} {short(r[0]):38s} {str(r[3:]):28s} {r[2] / 1e3:9.2f} {(r[1] - prev[1] - prev[2]) / 1e3:7.2f}")
txt = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + "\n")
else:
    print(txt)
