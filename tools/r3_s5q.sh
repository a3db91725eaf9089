cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in u4 u8 u4 u8; do
  echo "== $n"; PDHIP_LAB_LIB=pointdreamer_amd/csrc/build/lab_$n.so timeout 600 python tools/time_unet.py --batches 1 8 32 --out gpurun_out/s5q_$n.json 2>&1 | grep batch
done > gpurun_out/s5q.txt 2>&1
cat gpurun_out/s5q.txt
