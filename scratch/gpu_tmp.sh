R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python3 bench.py --config qwen3-32b --tp-virtual 8 --tp-xcd 1 --steps 32 --warmup 8 2>&1 | tail -1 | cut -c1-700
timeout 1500 python -m pytest tests/test_gpu_xengine.py -x -q -k "gqa4" 2>&1 | tail -3
