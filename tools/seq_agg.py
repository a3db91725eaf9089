"""Aggregate a tools/trace_seq.py launch list by (kernel, grid): count, total and average duration.  Usage: python tools/seq_agg.py gpurun_out/seq_n1.txt"""
import collections, sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip() and not l.startswith('#')]
agg = collections.OrderedDict()
for r in rows:
    k = (r[1], ' '.join(r[2:-2]))
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r[-2]); a[2] += float(r[-1])
tot = sum(v[1] for v in agg.values())
byk = collections.Counter()
for (k, g), (n, t, gp) in agg.items(): byk[k.split('<')[0]] += t
print(open(sys.argv[1]).readline().strip())
print('by kernel:', ', '.join(f'{k} {t:.0f}' for k, t in byk.most_common()))
print(f'sum of durations {tot:.1f} us, sum of gaps {sum(v[2] for v in agg.values()):.1f} us')
for k, (n, t, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t < (float(sys.argv[2]) if len(sys.argv) > 2 else 0): continue
    print(f"{k[0]:32s} {k[1]:22s} n={n:3d} total={t:8.1f} avg={t/n:7.2f} gap_before_avg={gp/n:6.2f}")
