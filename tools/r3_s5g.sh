cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in 0 16 64; do
  echo "== finc $f"; timeout 600 python tools/time_unet.py --batches 32 --finc $f --out gpurun_out/s5g_$f.json 2>&1 | grep batch
done > gpurun_out/s5g.txt 2>&1
cat gpurun_out/s5g.txt
