cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pp in 1 0 4 1 0; do
  echo "== profile-period $pp"
  python bench.py --profile-period $pp --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('achieved'), r.get('launches'))"
done > gpurun_out/s5n.txt 2>&1
cat gpurun_out/s5n.txt
