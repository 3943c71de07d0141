R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_tp.py -x -q -k "long_context_equals" 2>&1 | tail -5
