cd $GRAFT_REPO_ROOT
python tools/time_stages.py 2>&1 | grep -A8 "8f-1"
