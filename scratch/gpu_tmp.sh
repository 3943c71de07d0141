R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "prefill_then_decode" 2>&1 | tail -12
