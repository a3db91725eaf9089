# per-kernel times of P1+P2 (tools/raster_time.py): tools/prof_raster.sh [lib.so]
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_raster
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_raster -- python $GRAFT_REPO_ROOT/tools/raster_time.py ${1:+$GRAFT_REPO_ROOT/$1} > $GRAFT_REPO_ROOT/gpurun_out/prof_raster.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_raster/*/*.db gpurun_out/kernel_stats_raster.md > /dev/null 2>&1
rm -rf gpurun_out/prof_raster
grep -v amdgpu.ids gpurun_out/prof_raster.log | tail -3
grep "raster\|project\|minmax\|resize" gpurun_out/kernel_stats_raster.md | cut -c1-50,100-175
