cd $GRAFT_REPO_ROOT; python tools/chk_opt.py 2>&1 | tail -12
python -m pytest tests/test_gpu_round6.py -q -k "ragged or bit_packed" 2>&1 | tail -3
