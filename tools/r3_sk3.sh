for v in "" noglds nomfma nobar nolds noglds_nolds noglds_nobar; do
  lib=""; [ -n "$v" ] && lib="--lib pointdreamer_amd/csrc/build/labsk_$v.so"
  echo "=== variant: ${v:-product}"
  python tools/bench_sk.py --shapes 4 11 --stages 2 3 --tiles 1 2 --splits 1 2 $lib 2>&1 | grep -v amdgpu.ids
done
