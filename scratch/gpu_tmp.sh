R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "1p7b or refusals" 2>&1 | tail -4
CONFIG=qwen3-1.7b VARIANTS="8x4" NSEQ="16" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
