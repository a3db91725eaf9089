cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k unproject 2>&1 | tail -15
python -m pytest tests/test_gpu_geometry.py tests/test_gpu_round4.py -x -q -k "uq or unproject or full_size or shapes or pipeline" 2>&1 | tail -5
python tools/time_stages.py 2>&1 | grep -i "uq\|unproject\|total" 
python tools/time_shapes.py 2>&1 | tail -3
