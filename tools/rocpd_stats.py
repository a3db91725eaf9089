"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the classic --stats table (per-kernel calls,
total / average / min / max duration, percentage).  Usage: python tools/rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        lines.append(f"| `{short}` | {n} | {tot/1e6:.3f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} |")
    lines.append(f"\nTotal kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + "\n")
    print(out)


if __name__ == '__main__':
    main()
