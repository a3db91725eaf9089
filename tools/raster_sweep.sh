cd $GRAFT_REPO_ROOT
python tools/raster_time.py pointdreamer_amd/libpdhip.so /tmp/r_prod.npz
for v in ; do
  echo "== $v"; python tools/raster_time.py pointdreamer_amd/csrc/build/lab_raster_$v.so /tmp/r_$v.npz
  python - <<PY
import numpy as np
a=np.load('/tmp/r_prod.npz'); b=np.load('/tmp/r_$v.npz')
print('identical' if all(np.array_equal(a[k],b[k]) for k in a.files) else 'DIFFERENT', len(a.files))
PY
done
