# per-kernel minimum times of the hidden-point removal for the production library and for lab builds: tools/hpr_sum.sh [lib.so ...]
for l in "" "$@"; do echo "== ${l:-production}"; bash tools/prof_hpr.sh $l > /dev/null 2>&1; python - <<'PY'
tot=0
for line in open('gpurun_out/kernel_stats_hpr.md'):
    c=line.split('|')
    if len(c)>6 and 'k_hpr' in c[1]:
        name=c[1].strip()[:28]; mn=float(c[5]); print(f"   {name:30s} min {mn:7.2f} us"); tot+=mn
print('   sum of min', round(tot,1))
PY
grep "mismatch" gpurun_out/prof_hpr.log
done
