R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "eight_query_heads or small-320-150-8 or tiny-96" 2>&1 | tail -8
