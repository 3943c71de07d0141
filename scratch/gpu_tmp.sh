R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_tp.py tests/test_gpu_xengine.py -x -q -k "full_depth or eight_xcds_of_one or gqa4" --durations=6 2>&1 | tail -12
