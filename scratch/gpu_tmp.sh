R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_xengine.py -x -q -k "refusals" 2>&1 | tail -3
timeout 900 python3 bench.py --config qwen3-1.7b --steps 64 --warmup 16 --lean --lean-xcd 8 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); x=d['xcd_replicas']; print(d['value'], x.get('tokens_per_s'), x.get('frac'), x.get('parity'), x.get('error'))"
