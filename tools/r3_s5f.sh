cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in base late1w late5w late1; do
  echo "== $n"; timeout 300 python tools/bench_gnskip.py pointdreamer_amd/csrc/build/lab_$n.so 2>&1 | grep "variant 1" | head -4
done > gpurun_out/s5f.txt 2>&1
cat gpurun_out/s5f.txt
