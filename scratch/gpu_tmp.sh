R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "gqa4" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "not gqa4" 2>&1 | tail -3
