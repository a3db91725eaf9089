"""Aggregate a tools/trace_seq.py launch list by (kernel, grid): count, total and average duration.  Usage: python tools/seq_agg.py gpurun_out/seq_n1.txt"""
import collections, sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip() and not l.startswith('#')]
agg = collections.OrderedDict()
for r in rows:
    k = (r[1], ' '.join(r[2:-1]))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r[-1])
tot = sum(t for _, t in agg.values())
byk = collections.Counter()
for (k, g), (n, t) in agg.items(): byk[k.split('<')[0]] += t
print(open(sys.argv[1]).readline().strip())
print('by kernel:', ', '.join(f'{k} {t:.0f}' for k, t in byk.most_common()))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t < (float(sys.argv[2]) if len(sys.argv) > 2 else 0): continue
    print(f"{k[0]:32s} {k[1]:22s} n={n:3d} total={t:8.1f} avg={t/n:7.2f}")
