for l in "" pointdreamer_amd/csrc/build/lab_hpr_r5.so pointdreamer_amd/csrc/build/lab_hpr_r7.so; do echo "== $l"; bash tools/prof_hpr.sh $l > /dev/null 2>&1; python - <<'PY'
import re
tot=0
for line in open('gpurun_out/kernel_stats_hpr.md'):
    c=line.split('|')
    if len(c)>6 and 'k_hpr' in c[1]:
        name=c[1].strip()[:28]; calls=int(c[2]); mn=float(c[5])
        # min over calls = the skip-mask call (the other calls include the all-points case)
        print(f"   {name:30s} min {mn:7.2f} us"); tot+=mn
print('   sum of min', round(tot,1))
PY
grep "mismatch" gpurun_out/prof_hpr.log
done
