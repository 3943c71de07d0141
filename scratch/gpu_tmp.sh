R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CONFIG=qwen3-4b VARIANTS="12x6,8x8" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
CONFIG=qwen3-8b VARIANTS="12x6,8x8" NSEQ="8" timeout 900 python3 scratch/xr_time.py 2028 20 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_gpu_xengine.py -x -q -k "gqa4" 2>&1 | tail -3
