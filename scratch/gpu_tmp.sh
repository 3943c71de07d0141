R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python3 bench.py --config qwen3-32b --tp-virtual 8 --tp-xcd 1 --steps 32 --warmup 8 2>&1 | tail -1 | cut -c1-420
STAMPS=1 timeout 900 python3 scratch/xtp_time.py 16 4000 16 2>&1 | grep -v amdgpu.ids | tail -30
