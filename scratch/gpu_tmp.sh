R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_tp.py -x -q -k "eight_xcds" 2>&1 | tail -25
